"""fte_conv2d_fwd_s16 (bf16 storage, 1x1, no BN statistics) on one shape: microseconds per call -- for the FTE_IGEMM16_CFG / _DEEP /
_PERSIST hooks (which tile of the LDS-DMA kernels a small-shard pointwise launch should use).
    python scripts/dev/tile_cfg16.py HW CIN COUT B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import _lib
_lib.load(); _lib.set_mfma_dtype('bf16s')
call, q = _lib.call, _lib.query
st = torch.cuda.current_stream().cuda_stream
hw, cin, cout, n = [int(a) for a in sys.argv[1:5]]
x = torch.randn(n, hw, hw, cin, device='cuda').to(torch.bfloat16).view(torch.int16)
w = (torch.randn(cout, cin, device='cuda') * 0.05).to(torch.bfloat16).view(torch.int16)
z = torch.empty(n, hw, hw, cout, dtype=torch.int16, device='cuda')
nb = q('fte_conv2d_fwd_ws_bytes', n, hw, hw, cin, cout, 1, 1)
ws = torch.empty(nb // 4 + 1024, device='cuda')
_lib.query('fte_prof_enable', 1)
def run():
    call('fte_conv2d_fwd_s16', x, w, None, None, None, None, z, None, None, n, hw, hw, cin, cout, 1, 1, ws, ws.numel() * 4, st)
run(); torch.cuda.synchronize()
sym = _lib.prof_records(True)[-1][5] if _lib.prof_records(True) else '?'
_lib.query('fte_prof_enable', 0)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
print('%dx%d %d->%d B=%d  %6.1f us  %s' % (hw, hw, cin, cout, n, best, sym))
