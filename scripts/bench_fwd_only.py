import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
B = 512; reps = 10
st = torch.cuda.current_stream().cuda_stream
out = []
ws = torch.empty(256 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
for hw, cin, cout in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]:
    x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
    z = torch.empty(B, hw, hw, cout, device='cuda'); y = torch.empty_like(z); res = torch.randn_like(z)
    al = torch.full((cout,), 0.25, device='cuda')
    f = lambda: _lib.call('fte_conv3x3_fwd', x, w, None, al, res, z, y, B, hw, hw, cin, cout, 1, ws, wsb, st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    out.append('%dx%d/%d %.3fms %.1fTF' % (hw, hw, cin, t, 2.0 * B * hw * hw * 9 * cin * cout / t / 1e9))
print(' | '.join(out))
