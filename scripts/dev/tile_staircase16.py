"""fte_conv2d_bn_fwd (bf16 storage, 1x1) across the tile-count boundaries of the LDS-DMA tile kernels: does a launch of 784 tiles
(3.06 rounds of one block per CU) cost a fourth round?   python scripts/dev/tile_staircase16.py HW CIN COUT B1,B2,..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import _lib
_lib.load(); _lib.set_mfma_dtype('bf16s')
call, q = _lib.call, _lib.query
st = torch.cuda.current_stream().cuda_stream
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for n in [int(b) for b in sys.argv[4].split(',')]:
    x = torch.randn(n, hw, hw, cin, device='cuda').to(torch.bfloat16).view(torch.int16)
    w = (torch.randn(cout, cin, device='cuda') * 0.05).to(torch.bfloat16).view(torch.int16)
    z = torch.empty(n, hw, hw, cout, dtype=torch.int16, device='cuda')
    v = [torch.ones(cout, device='cuda') for _ in range(6)]
    nb = q('fte_conv2d_bn_fwd_ws_bytes', n, hw, hw, cin, cout, 1, 1)
    ws = torch.empty(nb // 4 + 1024, device='cuda')
    def run():
        call('fte_conv2d_bn_fwd', x, w, z, v[0], v[1], v[2], v[3], v[4], v[5], None, None, 1e-3, 0.999, None, None, None, n, hw, hw, cin, cout, 1, 1, 1, ws, ws.numel() * 4, st)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
    m = n * hw * hw
    print('%dx%d %d->%d  B=%3d  rows %6d  tiles(128x128) %5d  %6.1f us  %.3f us per 256 tiles' % (hw, hw, cin, cout, n, m, (m + 127) // 128 * (cout // 128), best, best / ((m + 127) // 128 * (cout // 128)) * 256))
