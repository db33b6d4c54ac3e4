"""fte_conv2d_bn_fwd (bf16 storage) on the 1x1 shapes of ResNeXt-50 at 128 images: microseconds per call, kernel + finalize, one stream,
averaged over back-to-back calls (no per-call synchronisation).   FTE_PW16=0 gives the tile kernels for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import _lib
_lib.load(); _lib.set_mfma_dtype('bf16s')
call, q = _lib.call, _lib.query
st = torch.cuda.current_stream().cuda_stream
shapes = [(28, 64, 256), (28, 64, 128), (28, 128, 256), (28, 256, 128), (28, 256, 256), (14, 256, 512), (14, 512, 256), (7, 512, 1024), (7, 1024, 512)]
n = 128
for hw, cin, cout in shapes:
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
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    mb = n * hw * hw * (cin + cout) * 2 / 1e6
    print('%2dx%2d %4d -> %4d  %6.1f us  %5.1f MB  %5.0f GB/s' % (hw, hw, cin, cout, us, mb, mb / us * 1e3))
