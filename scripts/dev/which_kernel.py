import sys, os
sys.path.insert(0, '/root/repo')
import torch
from tf_face_toolbox_amd import _lib
_lib.set_mfma_dtype('bf16s')
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
i16 = dict(dtype=torch.int16, device='cuda')
for hw, c, B in [(56, 64, 48), (56, 64, 64), (28, 128, 126), (28, 128, 200), (28, 128, 252), (14, 256, 262), (14, 256, 400), (14, 256, 504)]:
    w16 = torch.zeros(3, 3, c, c, **i16); w16t = torch.zeros(3, 3, c, c, **i16)
    x16 = torch.zeros(B, hw, hw, c, **i16); z16 = torch.empty_like(x16); y16 = torch.empty_like(x16); raw = torch.empty_like(x16); dzp = torch.empty_like(x16)
    al = torch.ones(c, device='cuda'); da = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    _lib.call('fte_prof_enable', 1)
    _lib.call('fte_conv2d_fwd_s16', x16, w16t, None, al, x16, z16, y16, None, None, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    _lib.call('fte_conv2d_dgrad_s16', x16, w16, x16, z16, al, raw, dzp, da, db, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    torch.cuda.synchronize()
    print(hw, c, B, [r[5] for r in _lib.prof_records(shapes=True)])
