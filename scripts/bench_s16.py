"""Times the bf16-STORAGE conv forward / data gradient (fte_conv2d_fwd_s16 / fte_conv2d_dgrad_s16) on the four stride-1 SphereNet
stages and prints a checksum of each result, so that kernel variants selected by environment hooks (FTE_IGEMM16_CFG, ...) can be
compared for speed AND equality in one gpurun call.        python scripts/bench_s16.py [B] [reps] [stages, e.g. 14,28]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
_lib.set_mfma_dtype('bf16s')
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
i16 = dict(dtype=torch.int16, device='cuda')


def T(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def csum(t):
    return int(t.view(torch.int16).to(torch.int64).sum().item()) & 0xffffffff


tot = [0.0, 0.0]
for hw, c, count in [(56, 64, 2), (28, 128, 4), (14, 256, 8), (7, 512, 2)]:
    if only and hw not in only: continue
    g = torch.Generator(device='cuda'); g.manual_seed(hw)
    w = torch.randn(3, 3, c, c, device='cuda', generator=g) * 0.05
    w16 = torch.empty(w.shape, **i16); w16t = torch.empty(3, 3, c, c, **i16)
    _lib.call('fte_pack_weights_bf16', w, w16, w16t, 3, c, c, st)
    al = torch.full((c,), 0.25, device='cuda')
    x16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    r16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    z16 = torch.empty(B, hw, hw, c, **i16); y16 = torch.empty_like(z16)
    raw16 = torch.empty_like(z16); dzp16 = torch.empty_like(z16)
    da = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    t1 = T(lambda: _lib.call('fte_conv2d_fwd_s16', x16, w16t, None, al, r16, z16, y16, None, None, B, hw, hw, c, c, 3, 1, ws, wsb, st))
    t2 = T(lambda: _lib.call('fte_conv2d_dgrad_s16', x16, w16, r16, z16, al, raw16, dzp16, da, db, B, hw, hw, c, c, 3, 1, ws, wsb, st))
    fl = 2.0 * B * hw * hw * 9 * c * c
    tot[0] += t1 * count; tot[1] += t2 * count
    print('%3dx%-3d %3d->%-3d x%d | fwd %.4f ms %6.1f TF | dgrad %.4f ms %6.1f TF | sums %08x %08x %08x %08x %.6e' % (
        hw, hw, c, c, count, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, csum(z16), csum(y16), csum(raw16), csum(dzp16), float(da.double().sum())))
print('per-step totals (ms): fwd %.3f dgrad %.3f' % tuple(tot))
