"""Achieved HBM bandwidth of the bandwidth-bound entry points on representative shapes at batch 512.
Algorithmic bytes = every tensor the op must read or write, once; peak 8 TB/s.  Prints a markdown table."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_face_toolbox_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = torch.cuda.current_stream().cuda_stream
PEAK = 8000.0
f32 = dict(dtype=torch.float32, device='cuda')


def T(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []


def rec(name, shape, nbytes, ms):
    gbs = nbytes / ms / 1e6
    rows.append('| `%s` | %s | %.1f | %.3f | %.0f | %.0f %% |' % (name, shape, nbytes / 1e6, ms, gbs, 100 * gbs / PEAK))


ws = torch.empty(64 << 20, **f32); wsb = ws.numel() * 4
for (h, c) in [(56, 64), (28, 256), (14, 512), (7, 1024), (4, 2048), (28, 128)]:
    r = B * h * h
    z = torch.randn(r, c, **f32); y = torch.empty_like(z); res = torch.randn_like(z); dy = torch.randn_like(z); dz = torch.empty_like(z)
    g = torch.ones(c, **f32); b = torch.zeros(c, **f32); v = [torch.empty(c, **f32) for _ in range(6)]
    tb = z.numel() * 4
    ms = T(lambda: _lib.call('fte_bn_train_fwd', z, g, b, res, y, v[0], v[1], v[2], v[3], v[4], v[5], r, c, 1e-3, 0.999, 1, ws, wsb, st))
    rec('fte_bn_train_fwd (+add+ReLU)', '%dx%dx%dx%d' % (B, h, h, c), 4 * tb, ms)          # z twice (stats, apply), res, y
    ms = T(lambda: _lib.call('fte_bn_train_bwd', dy, y, z, g, v[0], v[1], dz, v[2], v[3], r, c, ws, wsb, st))
    rec('fte_bn_train_bwd (ReLU mask)', '%dx%dx%dx%d' % (B, h, h, c), 7 * tb, ms)          # (dy, y, z) twice + dz
    ms = T(lambda: _lib.call('fte_relu_bwd', dy, y, dz, z.numel(), st))
    rec('fte_relu_bwd', '%dx%dx%dx%d' % (B, h, h, c), 3 * tb, ms)
x = torch.randn(B, 56, 56, 64, **f32); y = torch.empty(B, 28, 28, 64, **f32); idx = torch.empty(B, 28, 28, 64, dtype=torch.uint8, device='cuda')
dy = torch.randn_like(y); dx = torch.empty_like(x)
ms = T(lambda: _lib.call('fte_maxpool3x3s2_fwd', x, y, idx, B, 56, 56, 64, st)); rec('fte_maxpool3x3s2_fwd', '%dx56x56x64' % B, x.numel() * 4 + y.numel() * 5, ms)
ms = T(lambda: _lib.call('fte_maxpool3x3s2_bwd', dy, idx, dx, B, 56, 56, 64, st)); rec('fte_maxpool3x3s2_bwd', '%dx56x56x64' % B, x.numel() * 4 + y.numel() * 5, ms)
x = torch.randn(B, 4, 4, 2048, **f32); y = torch.empty(B, 2048, **f32)
ms = T(lambda: _lib.call('fte_gap_fwd', x, y, B, 16, 2048, st)); rec('fte_gap_fwd', '%dx4x4x2048' % B, x.numel() * 4, ms)
for (h, c, s) in [(28, 128, 2), (14, 128, 1), (14, 256, 2), (7, 256, 1), (7, 512, 2), (4, 512, 1)]:
    ho = (h + s - 1) // s
    x = torch.randn(B, h, h, c, **f32); w = torch.randn(3, 3, c, **f32); y = torch.empty(B, ho, ho, c, **f32); dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    nb = (x.numel() + y.numel()) * 4
    ms = T(lambda: _lib.call('fte_dwconv3x3_fwd', x, w, y, B, h, h, c, s, st)); rec('fte_dwconv3x3_fwd s%d' % s, '%dx%dx%dx%d' % (B, h, h, c), nb, ms)
    ms = T(lambda: _lib.call('fte_dwconv3x3_dgrad', dy, w, dx, B, h, h, c, s, st)); rec('fte_dwconv3x3_dgrad s%d' % s, '%dx%dx%dx%d' % (B, h, h, c), nb, ms)
    ms = T(lambda: _lib.call('fte_dwconv3x3_wgrad', x, dy, dw, B, h, h, c, s, ws, wsb, st)); rec('fte_dwconv3x3_wgrad s%d' % s, '%dx%dx%dx%d' % (B, h, h, c), nb, ms)
for (h, c) in [(14, 128), (7, 256), (4, 512)]:
    a = torch.randn(B * h * h, c, **f32); b = torch.randn_like(a); o = torch.empty_like(a)
    tbl = torch.tensor([((k & 1) << 16) | (k >> 1) for k in range(c)], dtype=torch.int32, device='cuda')     # interleave of the two first halves
    ms = T(lambda: _lib.call('fte_channel_gather', a, b, o, tbl, a.shape[0], c, c, c, st)); rec('fte_channel_gather', '%dx%dx%dx%d' % (B, h, h, c), 2 * o.numel() * 4, ms)
n = 29916352
w = torch.randn(n, **f32); acc = torch.zeros(n, **f32); g = torch.randn(n, **f32)
ms = T(lambda: _lib.call('fte_momentum_update', w, acc, g, n, 0.1, 0.9, 5e-4, 1.0, st)); rec('fte_momentum_update', '29.9 M params', 5 * n * 4, ms)
lg = torch.randn(B, 10624, **f32); lab = torch.randint(0, 10575, (B,), dtype=torch.int32, device='cuda'); lr_ = torch.empty(B, **f32); dl = torch.empty_like(lg)
ms = T(lambda: _lib.call('fte_softmax_ce_fwd_bwd', lg, lab, lr_, dl, B, 10575, 10624, 1.0 / B, st)); rec('fte_softmax_ce_fwd_bwd', '%dx10575' % B, 2 * lg.numel() * 4, ms)
x = torch.randn(B, 112, 112, 3, **f32); w = torch.randn(3, 3, 3, 64, **f32) * 0.1; bias = torch.zeros(64, **f32); al = torch.full((64,), 0.25, **f32)
z = torch.empty(B, 56, 56, 64, **f32); y = torch.empty_like(z); dz = torch.randn_like(z); dw = torch.empty_like(w)
ms = T(lambda: _lib.call('fte_conv3x3_first_fwd', x, w, bias, al, z, y, B, 112, 112, 3, 64, 2, st)); rec('fte_conv3x3_first_fwd', '%dx112x112x3 -> 64' % B, (x.numel() + 2 * z.numel()) * 4, ms)
wsf, wsfb = torch.empty(_lib.query('fte_conv3x3_first_wgrad_ws_bytes', B, 112, 112, 3, 64, 2) // 4 + 1024, **f32), 0
wsfb = wsf.numel() * 4
ms = T(lambda: _lib.call('fte_conv3x3_first_wgrad', x, dz, dw, B, 112, 112, 3, 64, 2, wsf, wsfb, st)); rec('fte_conv3x3_first_wgrad', '%dx112x112x3 -> 64' % B, (x.numel() + z.numel()) * 4, ms)
print('| entry point | shape | algorithmic MB | ms | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|')
print('\n'.join(rows))
