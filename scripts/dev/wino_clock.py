"""In-kernel timing of wino_mm_kernel (diagnostic build variants/libfte_wstamp.so, scripts/dev/build_wino_stamp.sh):
    FTE_LIB=variants/libfte_wstamp.so python scripts/dev/wino_clock.py [images]
per resBlock shape, forward and data gradient: the clock the chip holds under the kernel (d s_memtime / d s_memrealtime), shader cycles per
K-step of a tile's loop (ideal: 64 MFMAs x 64 cycles = 4096 per SIMD), cycles and microseconds of the epilogue, tiles per block."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
assert hasattr(lib, 'fte_debug_set_wino_stamp'), 'run with FTE_LIB=variants/libfte_wstamp.so'
lib.fte_debug_set_wino_stamp.argtypes = [ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
_lib.call('fte_set_conv_algo', 1)
stamp = torch.zeros(8 * 100000, dtype=torch.int64, device='cuda')
print('| layer, batch %d | op | ms | clock GHz | cycles per K-step (4096 = pipe) | loop us | epilogue cycles | epilogue us | tile us | tiles per block max |' % B)
print('|---|---|---|---|---|---|---|---|---|---|')
for hw, c in ((28, 128), (14, 256), (7, 512)):
    x = torch.randn(B, hw, hw, c, device='cuda') * 0.5; w = torch.randn(3, 3, c, c, device='cuda') * 0.05
    z = torch.empty_like(x); y = torch.empty_like(x); res = torch.randn_like(x); dzp = torch.empty_like(x); raw = torch.empty_like(x)
    al = torch.full((c,), 0.25, device='cuda'); da = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    need = max(_lib.query('fte_conv3x3_fwd_ws_bytes', B, hw, hw, c, c, 1), _lib.query('fte_conv3x3_dgrad_ws_bytes', B, hw, hw, c, c, 1))
    ws = torch.empty(need // 4 + 1024, device='cuda'); wsb = ws.numel() * 4
    for op, f in (('fwd', lambda: _lib.call('fte_conv3x3_fwd', x, w, None, al, res, z, y, B, hw, hw, c, c, 1, ws, wsb, st)),
                  ('dgrad', lambda: _lib.call('fte_conv3x3_dgrad', x, w, res, z, al, raw, dzp, da, db, B, hw, hw, c, c, 1, ws, wsb, st))):
        lib.fte_debug_set_wino_stamp(None)
        f(); torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 1.5:
            for _ in range(20): f()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        stamp.zero_()
        lib.fte_debug_set_wino_stamp(ctypes.c_void_p(stamp.data_ptr()))
        for _ in range(3): f()
        torch.cuda.synchronize()
        lib.fte_debug_set_wino_stamp(None)
        s = stamp.cpu().numpy().reshape(-1, 8)
        s = s[s[:, 5] > 0]
        clk = s[:, 2] / np.maximum(s[:, 3], 1) * 0.1
        cgh = float(np.median(clk))
        tiles = np.bincount(s[:, 6].astype(np.int64)).max()
        print('| %dx%d c%d | %s | %.3f | %.3f | %.0f | %.1f | %.0f | %.2f | %.1f | %d |' % (
            hw, hw, c, op, ms, cgh, float(np.median(s[:, 0] / s[:, 5])), float(np.median(s[:, 0])) / cgh / 1e3, float(np.median(s[:, 1])),
            float(np.median(s[:, 1])) / cgh / 1e3, float(np.median(s[:, 2])) / cgh / 1e3, tiles), flush=True)
