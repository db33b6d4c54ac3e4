"""Runs ONE conv shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
which, B, hw, cin, cout, stride = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
st = torch.cuda.current_stream().cuda_stream
ho = (hw + stride - 1) // stride
x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
z = torch.empty(B, ho, ho, cout, device='cuda'); y = torch.empty_like(z); res = torch.randn_like(z)
al = torch.full((cout,), 0.25, device='cuda'); alp = torch.full((cin,), 0.25, device='cuda')
dz = torch.randn_like(z); zp = torch.randn_like(x); raw = torch.empty_like(x); dzp = torch.empty_like(x); add = torch.randn_like(x)
da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda'); dw = torch.empty_like(w)
ws = torch.empty(256 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
for _ in range(reps):
    if which == 'fwd':
        _lib.call('fte_conv3x3_fwd', x, w, None, al, res if stride == 1 else None, z, y, B, hw, hw, cin, cout, stride, ws, wsb, st)
    elif which == 'dgrad':
        _lib.call('fte_conv3x3_dgrad', dz, w, add, zp, alp, raw, dzp, da, db, B, hw, hw, cin, cout, stride, ws, wsb, st)
    else:
        _lib.call('fte_conv3x3_wgrad', x, dz, dw, B, hw, hw, cin, cout, stride, ws, wsb, st)
torch.cuda.synchronize()
