"""In-kernel clock of the fp32 1x1 convs of ShuffleNet-v2's stages (diagnostic build variants/libfte_stamp.so): where a block of a
launch with a handful of K-steps spends its time -- prologue (entry -> first K-step), K loop, epilogue (incl. its stores).
    FTE_LIB=variants/libfte_stamp.so python scripts/dev/clock_probe_1x1.py [B]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
assert hasattr(lib, 'fte_debug_set_stamp'), 'run with FTE_LIB=variants/libfte_stamp.so (scripts/build_stamp_variant.sh)'
lib.fte_debug_set_stamp.argtypes = [ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
stamp = torch.zeros(8 * 200000, dtype=torch.int64, device='cuda')
print('| layer (batch %d) | op | us per launch | blocks | K-steps | clock GHz | cycles per K-step | block: prologue / loop / epilogue us | launch span of the blocks us |' % B)
print('|---|---|---|---|---|---|---|---|---|')
for hw, cin, cout in ((14, 128, 128), (7, 256, 256), (4, 512, 512), (28, 64, 128)):
    x = torch.randn(B, hw, hw, cin, device='cuda') * 0.5; w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    z = torch.empty(B, hw, hw, cout, device='cuda'); dz = torch.randn_like(z); dx = torch.empty_like(x)
    ops = {'fwd': lambda: _lib.call('fte_conv2d_fwd', x, w, None, None, None, None, z, B, hw, hw, cin, cout, 1, 1, ws, wsb, st),
           'dgrad': lambda: _lib.call('fte_conv2d_dgrad', dz, w, None, None, None, None, dx, None, None, B, hw, hw, cin, cout, 1, 1, ws, wsb, st)}
    for name, f in ops.items():
        lib.fte_debug_set_stamp(None)
        for _ in range(50): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        stamp.zero_()
        lib.fte_debug_set_stamp(ctypes.c_void_p(stamp.data_ptr()))
        f()
        torch.cuda.synchronize()
        lib.fte_debug_set_stamp(None)
        s = stamp.cpu().numpy().reshape(-1, 8)
        s = s[s[:, 2] > 0]
        clk = s[:, 0] / np.maximum(s[:, 1], 1) * 0.1
        c = float(np.median(clk))
        pro, loop, epi = (float(np.median(s[:, 4])) / c / 1e3, float(np.median(s[:, 0])) / c / 1e3, float(np.median(s[:, 6])) / c / 1e3)
        span = (float(s[:, 7].max()) - float(s[:, 3].min())) / 100.0          # s_memrealtime ticks of 10 ns: first loop start -> last block end
        print('| %dx%d %d->%d | %s | %.1f | %d | %d | %.2f | %.0f | %.1f / %.1f / %.1f | %.1f |' % (
            hw, hw, cin, cout, name, us, len(s), int(np.median(s[:, 2])), c, float(np.median(s[:, 0] / s[:, 2])), pro, loop, epi, span))
