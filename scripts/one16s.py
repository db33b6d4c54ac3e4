"""Times ONE bf16-STORAGE conv forward (fte_conv2d_fwd_s16: bf16 x / shortcut in, bf16 z / y out) at several batch sizes: how much of a
launch is tile-count quantisation (tiles vs resident block slots)?      python scripts/one16s.py HW CIN COUT B1,B2,... [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Bs = [int(v) for v in sys.argv[4].split(',')]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
_lib.set_mfma_dtype('bf16s')
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
i16 = dict(dtype=torch.int16, device='cuda')
w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
w16 = torch.empty(w.shape, **i16); w16t = torch.empty(3, 3, cout, cin, **i16)
_lib.call('fte_pack_weights_bf16', w, w16, w16t, 3, cin, cout, st)
al = torch.full((cout,), 0.25, device='cuda')
for B in Bs:
    x16 = torch.randn(B, hw, hw, cin, device='cuda').bfloat16().view(torch.int16)
    r16 = torch.randn(B, hw, hw, cout, device='cuda').bfloat16().view(torch.int16)
    z16 = torch.empty(B, hw, hw, cout, **i16); y16 = torch.empty_like(z16)
    f = lambda: _lib.call('fte_conv2d_fwd_s16', x16, w16t, None, al, r16, z16, y16, None, None, B, hw, hw, cin, cout, 3, 1, ws, wsb, st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    M = B * hw * hw
    print('%dx%d %d->%d B=%d: rows %d = %.2f row tiles of 128, %.3f ms, %.1f TF, %.2f us per 1000 rows' % (
        hw, hw, cin, cout, B, M, M / 128, t, 2.0 * M * 9 * cin * cout / t / 1e9, t * 1e3 / (M / 1000)))
