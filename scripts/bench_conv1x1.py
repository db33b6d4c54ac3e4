"""1x1 convolutions of the ShuffleNet / ResNet stages (small K): forward / dgrad / wgrad time against the HBM time of their
tensors (exploration)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = torch.cuda.current_stream().cuda_stream
def T(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ws = torch.empty(128 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
for hw, cin, cout in [(14, 128, 128), (7, 256, 256), (4, 512, 512), (28, 64, 256), (28, 256, 64), (14, 512, 128), (7, 1024, 256)]:
    x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(1, 1, cin, cout, device='cuda') * 0.05
    y = torch.empty(B, hw, hw, cout, device='cuda'); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * B * hw * hw * cin * cout
    by = (x.numel() + y.numel()) * 4
    t1 = T(lambda: _lib.call('fte_conv2d_fwd', x, w, None, None, None, None, y, B, hw, hw, cin, cout, 1, 1, ws, wsb, st))
    t2 = T(lambda: _lib.call('fte_conv2d_dgrad', dy, w, None, None, None, None, dx, None, None, B, hw, hw, cin, cout, 1, 1, ws, wsb, st))
    t3 = T(lambda: _lib.call('fte_conv2d_wgrad', x, dy, dw, B, hw, hw, cin, cout, 1, 1, ws, wsb, st))
    print('%2dx%-2d %4d->%-4d | fwd %.3f ms %5.1f TF %4.0f GB/s | dgrad %.3f ms %5.1f TF | wgrad %.3f ms %5.1f TF | hbm floor %.3f ms' % (
        hw, hw, cin, cout, t1, fl / t1 / 1e9, by / t1 / 1e6, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9, by / 6e9))
