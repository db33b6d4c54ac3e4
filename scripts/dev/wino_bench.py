"""Per-layer A/B of the conv algorithms at SphereNet's four resBlock shapes: direct implicit GEMM vs Winograd F(2x2,3x3), forward / data
gradient / filter gradient, whole call (transforms included) by HIP events on the launch stream + the MFMA kernel alone from the
library's launch records.  usage: python scripts/dev/wino_bench.py [images] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tf_face_toolbox_amd import _lib  # noqa: E402

call, query = _lib.call, _lib.query
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
SHAPES = [(56, 64), (28, 128), (14, 256), (7, 512)]
if len(sys.argv) > 3:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[3].split(',')]
st = torch.cuda.current_stream().cuda_stream


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    whole = e0.elapsed_time(e1) / REPS
    call('fte_prof_enable', 1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    call('fte_prof_enable', 0)
    recs = _lib.prof_records(shapes=True)
    per = {}
    for r in recs:
        per.setdefault(r[5], []).append((r[2], r[1]))
    kern = ', '.join('%s %.3f ms %.1f TF' % (k, sum(a for a, _ in v) / 3, sum(b for _, b in v) / sum(a for a, _ in v) / 1e9) for k, v in per.items())
    return whole, kern


for hw, c in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, hw, hw, c, device='cuda', generator=g)
    dz = torch.randn(N, hw, hw, c, device='cuda', generator=g)
    w = torch.randn(3, 3, c, c, device='cuda', generator=g) * 0.05
    al = torch.full((c,), 0.25, device='cuda')
    res = torch.randn(N, hw, hw, c, device='cuda', generator=g)
    z = torch.empty_like(x); y = torch.empty_like(x); raw = torch.empty_like(x); dzp = torch.empty_like(x)
    da = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda'); dw = torch.empty_like(w)
    direct_flops = 2.0 * N * hw * hw * 9 * c * c
    for algo, name in ((0, 'direct'), (1, 'winograd')):
        call('fte_set_conv_algo', algo)
        need = max(query('fte_conv3x3_fwd_ws_bytes', N, hw, hw, c, c, 1), query('fte_conv3x3_dgrad_ws_bytes', N, hw, hw, c, c, 1),
                   query('fte_conv3x3_wgrad_ws_bytes', N, hw, hw, c, c, 1))
        ws = torch.empty(need // 4 + 1024, device='cuda')
        nb = ws.numel() * 4
        ops = [('fwd', lambda: call('fte_conv3x3_fwd', x, w, None, al, res, z, y, N, hw, hw, c, c, 1, ws, nb, st)),
               ('dgrad', lambda: call('fte_conv3x3_dgrad', dz, w, res, z, al, raw, dzp, da, db, N, hw, hw, c, c, 1, ws, nb, st)),
               ('wgrad', lambda: call('fte_conv3x3_wgrad', x, dz, dw, N, hw, hw, c, c, 1, ws, nb, st))]
        for op, fn in ops:
            whole, kern = timed(fn)
            print('%3dx%-3d c%-4d n%-4d %-8s %-5s whole %.3f ms = %.1f direct-equivalent TF/s | %s' % (hw, hw, c, N, name, op, whole, direct_flops / whole / 1e9, kern), flush=True)
        del ws
    call('fte_set_conv_algo', 2)
