"""In-kernel clock of the fp32 MFMA conv kernels (diagnostic build variants/libfte_stamp.so, scripts/build_stamp_variant.sh):
    FTE_LIB=variants/libfte_stamp.so python scripts/clock_probe.py [B]
For each conv shape: >= 2 s of back-to-back launches on random data (the chip settles at the clock it holds under this
load), then one stamped launch: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz per block (median over blocks), shader
cycles per K-step of a block, and what that means for the MFMA pipe:
  busy = resident waves per SIMD x 16 MFMAs x 64 cycles / cycles per K-step  (every wave issues 16 dependent 64-cycle MFMAs per
  K-step; `resident` from the block's register / LDS footprint), and the TFLOP/s ceiling AT THE MEASURED CLOCK."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
assert hasattr(lib, 'fte_debug_set_stamp'), 'run with FTE_LIB=variants/libfte_stamp.so (scripts/build_stamp_variant.sh)'
lib.fte_debug_set_stamp.argtypes = [ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
stamp = torch.zeros(8 * 200000, dtype=torch.int64, device='cuda')
print('| layer (fwd, batch %d) | ms | TFLOP/s | clock GHz (median / p10 / p90) | cycles per K-step | peak at that clock | frac of it | block: prologue / loop / epilogue us | blocks in their K loop per CU |' % B)
print('|---|---|---|---|---|---|---|---|---|')
for hw, cin, cout in ((56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)):
    x = torch.randn(B, hw, hw, cin, device='cuda') * 0.5; w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
    z = torch.empty(B, hw, hw, cout, device='cuda'); y = torch.empty_like(z); res = torch.randn_like(z)
    al = torch.full((cout,), 0.25, device='cuda')
    f = lambda: _lib.call('fte_conv3x3_fwd', x, w, None, al, res, z, y, B, hw, hw, cin, cout, 1, ws, wsb, st)
    lib.fte_debug_set_stamp(None)
    f(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.0:
        for _ in range(20): f()
        torch.cuda.synchronize(); n += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    stamp.zero_()
    lib.fte_debug_set_stamp(ctypes.c_void_p(stamp.data_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    lib.fte_debug_set_stamp(None)
    s = stamp.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 2] > 0]
    clk = s[:, 0] / np.maximum(s[:, 1], 1) * 0.1          # GHz
    cyc = s[:, 0] / s[:, 2]
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    tf = fl / ms / 1e9
    c = float(np.median(clk))
    peak = 157.3 * c / 2.4
    pro, loop, epi = (float(np.median(s[:, 4])) / c / 1e3, float(np.median(s[:, 0])) / c / 1e3, float(np.median(s[:, 6])) / c / 1e3)
    inloop = tf / peak * float(np.median(cyc)) / 1024.0
    print('| %dx%d %d->%d | %.3f | %.1f | %.3f / %.3f / %.3f | %.0f | %.1f | %.3f | %.1f / %.1f / %.1f | %.1f |' % (
        hw, hw, cin, cout, ms, tf, c, np.percentile(clk, 10), np.percentile(clk, 90), float(np.median(cyc)), peak, tf / peak, pro, loop, epi, inloop))
