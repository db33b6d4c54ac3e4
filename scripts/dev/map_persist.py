"""Debug: which (row, column) of the product lands where -- 1x1 conv whose result is the column index / the row index mod 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tf_face_toolbox_amd import _lib
_lib.set_mfma_dtype('bf16s')
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
i16 = dict(dtype=torch.int16, device='cuda')
B, hw, cin, cout = 300, 14, 64, 256
M = B * hw * hw
for what in ('col', 'row'):
    x = torch.zeros(M, cin, device='cuda'); w = torch.zeros(1, 1, cin, cout, device='cuda')
    if what == 'col':
        x[:, 0] = 1; w[0, 0, 0, :] = torch.arange(cout, device='cuda').float()
    else:
        x[:, 0] = (torch.arange(M, device='cuda') % 128).float(); w[0, 0, 0, :] = 1
    x16 = x.bfloat16().view(torch.int16).reshape(B, hw, hw, cin)
    w16 = torch.empty(w.shape, **i16); w16t = torch.empty(1, 1, cout, cin, **i16)
    _lib.call('fte_pack_weights_bf16', w, w16, w16t, 1, cin, cout, st)
    z16 = torch.zeros(B, hw, hw, cout, **i16); y16 = torch.zeros_like(z16)
    _lib.call('fte_prof_enable', 1)
    _lib.call('fte_conv2d_fwd_s16', x16, w16t, None, None, None, z16, y16, None, None, B, hw, hw, cin, cout, 1, 1, ws, wsb, st)
    torch.cuda.synchronize()
    print(what, [(r[5], r[3]) for r in _lib.prof_records(shapes=True)])
    z = z16.view(torch.bfloat16).float().reshape(M, cout)
    for r in (0, 1, 5, 33, 127, 128 + 70):
        print('  row', r, z[r, :40].int().tolist())
