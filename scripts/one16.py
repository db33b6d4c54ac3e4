"""Times ONE bf16-source conv forward (fte_conv2d_fwd16) at batch B: for ablation / PMC passes of igemm16.hip.
    python scripts/one16.py HW CIN COUT [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 512
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
z = torch.empty(B, hw, hw, cout, device='cuda'); y = torch.empty_like(z); res = torch.randn_like(z)
al = torch.full((cout,), 0.25, device='cuda')
i16 = dict(dtype=torch.int16, device='cuda')
x16 = torch.empty(x.shape, **i16); w16 = torch.empty(w.shape, **i16); w16t = torch.empty(3, 3, cout, cin, **i16); y16 = torch.empty(z.shape, **i16)
_lib.call('fte_to_bf16', x, x16, x.numel(), st); _lib.call('fte_pack_weights_bf16', w, w16, w16t, 3, cin, cout, st)
f = lambda: _lib.call('fte_conv2d_fwd16', x16, w16t, None, al, res, z, y, y16, B, hw, hw, cin, cout, 3, 1, ws, wsb, st)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps
print('%dx%d %d->%d B=%d: %.3f ms %.1f TF  (FTE_IGEMM16_ABL=%s)' % (hw, hw, cin, cout, B, t, 2.0 * B * hw * hw * 9 * cin * cout / t / 1e9, os.environ.get('FTE_IGEMM16_ABL', '0')))
