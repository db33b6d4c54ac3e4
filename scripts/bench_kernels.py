"""Times the conv kernels of the four SphereNet stages (B images) through the C ABI; prints TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = torch.cuda.current_stream().cuda_stream
def T(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ws = torch.empty(256 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
tot = {'fwd': 0, 'dgrad': 0, 'wgrad': 0}
cases = [(56, 64, 64, 1, 2), (56, 64, 128, 2, 1), (28, 128, 128, 1, 4), (28, 128, 256, 2, 1), (14, 256, 256, 1, 8), (14, 256, 512, 2, 1), (7, 512, 512, 1, 2)]
for hw, cin, cout, stride, count in cases:
    ho = (hw + stride - 1) // stride
    x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
    z = torch.empty(B, ho, ho, cout, device='cuda'); y = torch.empty_like(z); res = torch.randn_like(z)
    al = torch.full((cout,), 0.25, device='cuda'); alp = torch.full((cin,), 0.25, device='cuda')
    dz = torch.randn_like(z); zp = torch.randn_like(x); raw = torch.empty_like(x); dzp = torch.empty_like(x); add = torch.randn_like(x)
    da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda'); dw = torch.empty_like(w)
    fl = 2.0 * B * ho * ho * 9 * cin * cout
    t1 = T(lambda: _lib.call('fte_conv3x3_fwd', x, w, None, al, res if stride == 1 else None, z, y, B, hw, hw, cin, cout, stride, ws, wsb, st))
    t2 = T(lambda: _lib.call('fte_conv3x3_dgrad', dz, w, add, zp, alp, raw, dzp, da, db, B, hw, hw, cin, cout, stride, ws, wsb, st))
    t3 = T(lambda: _lib.call('fte_conv3x3_wgrad', x, dz, dw, B, hw, hw, cin, cout, stride, ws, wsb, st))
    tot['fwd'] += t1 * count; tot['dgrad'] += t2 * count; tot['wgrad'] += t3 * count
    print('%3dx%-3d %3d->%-3d s%d x%d | fwd %.4f ms %5.1f TF | dgrad %.4f ms %5.1f TF | wgrad %.4f ms %5.1f TF' % (
        hw, hw, cin, cout, stride, count, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9))
print('per-step conv totals (ms):', {k: round(v, 2) for k, v in tot.items()}, 'sum %.2f' % sum(tot.values()))
