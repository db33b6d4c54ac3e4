"""-m gpu: layer kernels of the BN / pooling nets against the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import call, query, dev, host, stream, ws, check_maxabs, check_rell2


def _rng(s):
    return np.random.default_rng(s)


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', [
    (3, 14, 14, 64, 256, 1, 1), (2, 14, 14, 256, 64, 1, 1), (3, 12, 12, 256, 512, 1, 2), (2, 7, 7, 512, 128, 1, 2),
    (2, 9, 9, 64, 64, 3, 2), (40, 14, 14, 1024, 256, 1, 1),
    (360, 14, 14, 128, 128, 1, 1), (131, 7, 7, 256, 256, 1, 1), (5, 3, 3, 64, 64, 1, 1), (77, 4, 4, 512, 512, 1, 1),
])
def test_conv2d_1x1_and_3x3(n, h, w, cin, cout, ks, stride):
    r = _rng(1)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((ks, ks, cin, cout)) * 0.05
    z_ref = ops.conv2d_fwd(x, wt, stride)
    z = torch.empty(z_ref.shape, device='cuda')
    wsb, nb = ws(max(query('fte_conv2d_fwd_ws_bytes', n, h, w, cin, cout, ks, stride),
                     query('fte_conv2d_dgrad_ws_bytes', n, h, w, cin, cout, ks, stride),
                     query('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, ks, stride)))
    call('fte_conv2d_fwd', dev(x), dev(wt), None, None, None, None, z, n, h, w, cin, cout, ks, stride, wsb, nb, stream())
    check_maxabs(host(z), z_ref, what='fwd')
    dz = r.standard_normal(z_ref.shape)
    dx_ref, dw_ref = ops.conv2d_bwd(x, wt, dz, stride)
    dx = torch.full(x.shape, 7.0, device='cuda')                 # poison: zero-gradient parity classes must be WRITTEN
    call('fte_conv2d_dgrad', dev(dz), dev(wt), None, None, None, None, dx, None, None, n, h, w, cin, cout, ks, stride, wsb, nb, stream())
    check_maxabs(host(dx), dx_ref, what='dgrad')
    dw = torch.empty(ks, ks, cin, cout, device='cuda')
    call('fte_conv2d_wgrad', dev(x), dev(dz), dw, n, h, w, cin, cout, ks, stride, wsb, nb, stream())
    check_maxabs(host(dw), dw_ref, what='wgrad')


@pytest.mark.parametrize('n,h,w,cin,cout', [(300, 7, 7, 128, 192), (9, 5, 5, 64, 64)])
def test_conv1x1_dgrad_accumulates_into_an_existing_gradient(n, h, w, cin, cout):
    """`addin` (a tensor with two consumers, nets/resnet.py shortcut + branch): dx = addin + dz * w^T, on the persistent kernel
    (several tiles per block at the first shape)."""
    r = _rng(11)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((1, 1, cin, cout)) * 0.05
    dz = r.standard_normal((n, h, w, cout)); prev = r.standard_normal(x.shape)
    dx_ref, _ = ops.conv2d_bwd(x, wt, dz, 1)
    dx = torch.full(x.shape, 7.0, device='cuda')
    wsb, nb = ws(query('fte_conv2d_dgrad_ws_bytes', n, h, w, cin, cout, 1, 1))
    call('fte_conv2d_dgrad', dev(dz), dev(wt), dev(prev), None, None, None, dx, None, None, n, h, w, cin, cout, 1, 1, wsb, nb, stream())
    check_maxabs(host(dx), dx_ref + prev, what='dgrad + addin')


@pytest.mark.parametrize('rows_shape,c', [((3, 9, 7), 64), ((16, 28, 28), 256), ((2, 4, 4), 2048), ((64, 14, 14), 1024),
                                          ((5, 11, 13), 32), ((2, 3, 3), 192), ((37,), 8)])
def test_batch_norm_train_fwd_bwd(rows_shape, c):
    r = _rng(2)
    shape = rows_shape + (c,)
    z = r.standard_normal(shape) * 2.0 + 3.0                      # mean well away from 0: E[x^2]-E[x]^2 would lose digits
    gamma = 1 + 0.2 * r.standard_normal(c); beta = 0.3 * r.standard_normal(c)
    res = r.standard_normal(shape)
    rows = int(np.prod(rows_shape))
    bn_ref, cache = ops.bn_train_fwd(z, gamma, beta)
    y_ref = np.maximum(bn_ref + res, 0)
    mm = r.standard_normal(c) * 0.1; mv = 1 + 0.1 * r.random(c)
    mm_ref, mv_ref = ops.bn_moving_update(mm, mv, cache['mean'], cache['var'], rows)
    y = torch.empty(shape, device='cuda')
    mean, rstd, scale, shift = [torch.empty(c, device='cuda') for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    wsb, nb = ws(query('fte_bn_ws_bytes', c))
    zd, gd = dev(z), dev(gamma)
    call('fte_bn_train_fwd', zd, gd, dev(beta), dev(res), y, mean, rstd, scale, shift, mmd, mvd, rows, c, 1e-3, 0.999, 1, wsb, nb, stream())
    check_maxabs(host(mean), cache['mean'], 1e-6, 'mean'); check_maxabs(host(rstd), cache['rstd'], 2e-6, 'rstd')
    check_maxabs(host(y), y_ref, what='y = relu(bn + res)')
    check_maxabs(host(mmd), mm_ref, 1e-6, 'moving mean'); check_maxabs(host(mvd), mv_ref, 1e-6, 'moving var')
    # backward through relu(bn(z) + res): g = dy * (y > 0)
    dy = r.standard_normal(shape)
    g_ref = dy * (y_ref > 0)
    dz_ref, dg_ref, db_ref = ops.bn_train_bwd(g_ref, gamma, cache)
    dz = torch.empty(shape, device='cuda'); dg = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    call('fte_bn_train_bwd', dev(dy), y, zd, gd, mean, rstd, dz, dg, db, rows, c, wsb, nb, stream())
    check_maxabs(host(dz), dz_ref, what='dz'); check_rell2(host(dg), dg_ref, what='dgamma'); check_rell2(host(db), db_ref, what='dbeta')
    g = torch.empty(shape, device='cuda')
    call('fte_relu_bwd', dev(dy), y, g, int(np.prod(shape)), stream())
    check_maxabs(host(g), g_ref, 1e-7, 'relu bwd')
    # no mask / no residual / no relu variant, and inference mode
    call('fte_bn_train_fwd', zd, gd, dev(beta), None, y, mean, rstd, scale, shift, None, None, rows, c, 1e-3, 0.999, 0, wsb, nb, stream())
    check_maxabs(host(y), bn_ref, what='bn plain')
    call('fte_bn_train_bwd', dev(dy), None, zd, gd, mean, rstd, dz, dg, db, rows, c, wsb, nb, stream())
    check_maxabs(host(dz), ops.bn_train_bwd(dy, gamma, cache)[0], what='dz plain')
    call('fte_bn_infer_fwd', zd, gd, dev(beta), dev(mm), dev(mv), None, y, scale, shift, rows, c, 1e-3, 0, stream())
    check_maxabs(host(y), ops.bn_infer(z, gamma, beta, mm, mv), what='bn infer')


@pytest.mark.parametrize('n,h,w,c', [(2, 56, 56, 64), (3, 7, 6, 8), (2, 13, 9, 24), (1, 2, 2, 4), (3, 4, 6, 32), (2, 8, 7, 8)])
def test_maxpool_gap_dropout(n, h, w, c):
    r = _rng(3)
    x = r.integers(0, 4, (n, h, w, c)).astype(np.float64) + 0.25 * r.integers(0, 2, (n, h, w, c))   # plenty of ties
    y_ref, cache = ops.maxpool3x3s2_fwd(x)
    y = torch.empty(y_ref.shape, device='cuda'); idx = torch.empty(y_ref.shape, dtype=torch.uint8, device='cuda')
    call('fte_maxpool3x3s2_fwd', dev(x), y, idx, n, h, w, c, stream())
    assert np.array_equal(host(y), y_ref)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), cache['arg'])
    dy = r.standard_normal(y_ref.shape)
    dx = torch.empty(x.shape, device='cuda')
    call('fte_maxpool3x3s2_bwd', dev(dy), idx, dx, n, h, w, c, stream())
    check_maxabs(host(dx), ops.maxpool3x3s2_bwd(dy, cache), 1e-6, 'maxpool bwd')
    g = torch.empty(n, c, device='cuda')
    call('fte_gap_fwd', dev(x), g, n, h * w, c, stream())
    check_maxabs(host(g), ops.gap_fwd(x), 1e-6, 'gap')
    dg = r.standard_normal((n, c)); dxx = torch.empty(x.shape, device='cuda')
    call('fte_gap_bwd', dev(dg), dxx, n, h * w, c, stream())
    check_maxabs(host(dxx), ops.gap_bwd(dg, x.shape), 1e-6, 'gap bwd')
    f = r.standard_normal((n, 2048)); m = torch.empty(n, 2048, device='cuda'); o = torch.empty(n, 2048, device='cuda')
    call('fte_dropout_fwd', dev(f), m, o, n * 2048, 0.5, 1234, stream())
    mh = host(m)
    assert set(np.unique(mh)) <= {0.0, 1.0} and 0.4 < mh.mean() < 0.6
    check_maxabs(host(o), ops.dropout_fwd(f, mh, 0.5), 1e-7, 'dropout')
    m2 = torch.empty_like(m)
    call('fte_dropout_fwd', dev(f), m2, o, n * 2048, 0.5, 1235, stream())
    assert not torch.equal(m, m2)                                   # another seed, another mask
    call('fte_dropout_bwd', dev(f), m, o, n * 2048, 0.5, stream())
    check_maxabs(host(o), f * mh / 0.5, 1e-7, 'dropout bwd')


def test_first_layer_7x7_as_im2col_gemm():
    r = _rng(4)
    n, h, w, cin, cout, ks, stride, kpad = 3, 32, 24, 3, 64, 7, 2, 160
    x = r.uniform(-1, 1, (n, h, w, cin)); wt = r.standard_normal((ks, ks, cin, cout)) * 0.1
    z_ref = ops.conv2d_fwd(x, wt, stride)
    ho, wo = z_ref.shape[1], z_ref.shape[2]
    m = n * ho * wo
    cols = torch.empty(m, kpad, device='cuda')
    call('fte_im2col_first', dev(x), cols, n, h, w, cin, ks, stride, kpad, stream())
    wp = np.zeros((kpad, cout)); wp[:ks * ks * cin] = wt.reshape(-1, cout)
    z = torch.empty(m, cout, device='cuda')
    wsb, nb = ws(max(query('fte_gemm_ws_bytes', m, cout, kpad), 1 << 20))
    call('fte_gemm_nn', cols, dev(wp), None, z, m, cout, kpad, wsb, nb, stream())
    check_maxabs(host(z).reshape(z_ref.shape), z_ref, what='stem fwd')
    dz = r.standard_normal(z_ref.shape)
    _, dw_ref = ops.conv2d_bwd(x, wt, dz, stride, need_dx=False)
    dw = torch.empty(kpad, cout, device='cuda')
    call('fte_gemm_tn', cols, dev(dz.reshape(m, cout)), dw, m, cout, kpad, wsb, nb, stream())
    check_maxabs(host(dw)[:ks * ks * cin].reshape(wt.shape), dw_ref, what='stem wgrad')
    assert float(dw[ks * ks * cin:].abs().max()) == 0.0


@pytest.mark.parametrize('n,h,w,c,groups,stride', [(3, 14, 14, 128, 32, 1), (2, 13, 9, 256, 32, 2), (2, 8, 8, 512, 32, 1),
                                                   (2, 7, 7, 1024, 32, 2), (4, 28, 28, 128, 32, 2)])
def test_grouped_conv3x3(n, h, w, c, groups, stride):
    r = _rng(5)
    gw = c // groups
    x = r.standard_normal((n, h, w, c)); wt = r.standard_normal((groups, 3, 3, gw, gw)) * 0.2
    # oracle: the reference's split / conv / concat (nets/resnext.py:43-49)
    ys = [ops.conv2d_fwd(x[..., g * gw:(g + 1) * gw], wt[g], stride) for g in range(groups)]
    y_ref = np.concatenate(ys, axis=-1)
    y = torch.empty(y_ref.shape, device='cuda')
    call('fte_gconv3x3_fwd', dev(x), dev(wt), y, n, h, w, c, groups, stride, stream())
    check_maxabs(host(y), y_ref, what='gconv fwd')
    dz = r.standard_normal(y_ref.shape)
    dx_ref = np.zeros_like(x); dw_ref = np.zeros_like(wt)
    for g in range(groups):
        dxg, dwg = ops.conv2d_bwd(x[..., g * gw:(g + 1) * gw], wt[g], dz[..., g * gw:(g + 1) * gw], stride)
        dx_ref[..., g * gw:(g + 1) * gw] = dxg; dw_ref[g] = dwg
    dx = torch.empty(x.shape, device='cuda')
    call('fte_gconv3x3_dgrad', dev(dz), dev(wt), dx, n, h, w, c, groups, stride, stream())
    check_maxabs(host(dx), dx_ref, what='gconv dgrad')
    dw = torch.empty(wt.shape, device='cuda')
    wsb, nb = ws(query('fte_gconv3x3_wgrad_ws_bytes', n, h, w, c, groups, stride))
    call('fte_gconv3x3_wgrad', dev(x), dev(dz), dw, n, h, w, c, groups, stride, wsb, nb, stream())
    check_maxabs(host(dw), dw_ref, what='gconv wgrad')


def test_se_gate_pieces():
    r = _rng(6)
    n, hw, c = 5, 49, 256
    x = r.standard_normal((n, hw, c)); gate = 1 / (1 + np.exp(-r.standard_normal((n, c))))
    y = torch.empty(n, hw, c, device='cuda')
    call('fte_channel_scale_fwd', dev(x), dev(gate), y, n, hw, c, stream())
    check_maxabs(host(y), x * gate[:, None, :], 1e-6, 'scale fwd')
    dy = r.standard_normal((n, hw, c)); dx = torch.empty(n, hw, c, device='cuda'); dg = torch.empty(n, c, device='cuda')
    call('fte_channel_scale_bwd', dev(dy), dev(x), dev(gate), dx, dg, n, hw, c, 0, stream())
    check_maxabs(host(dx), dy * gate[:, None, :], 1e-6, 'scale dx'); check_maxabs(host(dg), (dy * x).sum(1), 1e-5, 'scale dgate')
    call('fte_channel_scale_bwd', dev(dy), dev(x), dev(gate), dx, dg, n, hw, c, 1, stream())
    check_maxabs(host(dg), (dy * x).sum(1) * gate * (1 - gate), 1e-5, 'scale d(pre-sigmoid)')
    v = r.standard_normal(1000) * 3
    for kind, f, df in ((0, lambda a: np.maximum(a, 0), lambda o: (o > 0).astype(float)), (1, lambda a: 1 / (1 + np.exp(-a)), lambda o: o * (1 - o))):
        o = torch.empty(1000, device='cuda'); call('fte_act_fwd', dev(v), o, 1000, kind, stream())
        check_maxabs(host(o), f(v), 1e-6, 'act fwd')
        d = torch.empty(1000, device='cuda'); call('fte_act_bwd', dev(v), o, d, 1000, kind, stream())
        check_maxabs(host(d), v * df(f(v)), 1e-5, 'act bwd')


@pytest.mark.parametrize('flags', [0, 3])
@pytest.mark.parametrize('n,hw,c', [(3, 49, 64), (2, 10, 96), (5, 196, 128), (4, 16, 2048), (2, 784, 256), (1, 67, 32)])
def test_se_residual_block_fused_kernels(n, hw, c, flags):
    """fte_se_squeeze / _apply_fwd / _bwd_gate / _bn_bwd_coef / _bn_bwd_apply (csrc/layers.hip "SE residual block"; nets/resnet.py:63-92 +
    the gate of nets/shufflenet_v2.py:79-85) against float64 on the same inputs: fp32 tensors (flags 0) and bf16 storage (flags 3: the
    inputs ARE bf16 values, stored results within half a bf16 step), spatial sizes that are not multiples of the 16 row lanes or of
    the 64-row trip, 4x4 images, one image."""
    r = _rng(41)
    s16 = flags == 3
    rd = (lambda a: ops.bf16_round(np.asarray(a, np.float64))) if s16 else (lambda a: np.asarray(a, np.float32).astype(np.float64))

    def tdev(a):          # host values -> device tensor in the storage the flags say
        t = dev(a)
        return t.bfloat16().view(torch.int16) if s16 else t

    def thost(t):
        return host(t.view(torch.bfloat16).float()) if s16 else host(t)
    z = rd(r.standard_normal((n, hw, c)) * 1.5 + 0.3); sc_ = rd(r.standard_normal((n, hw, c)))
    gamma = 1 + 0.2 * r.standard_normal(c); beta = 0.1 * r.standard_normal(c)
    gamma, beta = np.asarray(gamma, np.float32).astype(np.float64), np.asarray(beta, np.float32).astype(np.float64)
    mean = z.reshape(-1, c).mean(0); rstd = 1 / np.sqrt(z.reshape(-1, c).var(0) + 1e-3)
    mean, rstd = np.asarray(mean, np.float32).astype(np.float64), np.asarray(rstd, np.float32).astype(np.float64)
    scale = np.asarray(gamma * rstd, np.float32).astype(np.float64); shift = np.asarray(beta - mean * scale, np.float32).astype(np.float64)
    gate = np.asarray(1 / (1 + np.exp(-r.standard_normal((n, c)))), np.float32).astype(np.float64)
    # forward
    sq = torch.empty(n, c, device='cuda'); xm = torch.empty(n, c, device='cuda')
    call('fte_se_squeeze', tdev(z), dev(scale), dev(shift), dev(mean), dev(rstd), sq, xm, n, hw, c, flags & 1, stream())
    zm = z.mean(1)
    check_maxabs(host(sq), zm * scale + shift, 2e-5, 'sq')
    assert np.abs(host(xm) - (zm - mean) * rstd).max() <= 2e-6 * (1 + (np.abs(mean) * rstd).max())
    out = torch.empty(n, hw, c, dtype=torch.int16 if s16 else torch.float32, device='cuda')
    call('fte_se_apply_fwd', tdev(z), dev(scale), dev(shift), dev(gate), tdev(sc_), out, n, hw, c, flags, stream())
    y = z * scale + shift
    out_ref = np.maximum(y * gate[:, None, :] + sc_, 0)
    lim = np.abs(out_ref) * (2.0 ** -8 if s16 else 0) + 2e-5 * np.abs(out_ref).max()
    assert (np.abs(thost(out) - out_ref) <= lim).all(), 'se_apply_fwd'
    # backward, on the STORED out
    outv = thost(out)
    dy = rd(r.standard_normal((n, hw, c)))
    g = torch.empty_like(out); s1 = torch.empty(n, c, device='cuda'); s2 = torch.empty_like(s1); dgate = torch.empty_like(s1)
    call('fte_se_bwd_gate', tdev(dy), out, tdev(z), dev(gamma), dev(beta), dev(mean), dev(rstd), dev(gate), g, s1, s2, dgate, n, hw, c, flags, stream())
    g_ref = dy * (outv > 0)
    gv = thost(g)
    assert np.array_equal(gv, rd(g_ref)), 'g = dy * (out > 0), rounded once where stored'
    xhat = (z - mean) * rstd
    check_maxabs(host(s1), gv.sum(1), 2e-6 * max(1.0, np.abs(gv).sum(1).max() / max(np.abs(gv.sum(1)).max(), 1e-30)), 's1')
    check_maxabs(host(s2), (gv * xhat).sum(1), 2e-6 * max(1.0, np.abs(gv * xhat).sum(1).max() / max(np.abs((gv * xhat).sum(1)).max(), 1e-30)), 's2')
    dg_ref = (gamma * (gv * xhat).sum(1) + beta * gv.sum(1)) * gate * (1 - gate)
    check_maxabs(host(dgate), dg_ref, 2e-5 * max(1.0, (np.abs(gamma) * np.abs(gv * xhat).sum(1) + np.abs(beta) * np.abs(gv).sum(1)).max() / max(np.abs(dg_ref).max(), 1e-30)), 'dgate')
    dsq = np.asarray(0.1 * r.standard_normal((n, c)), np.float32).astype(np.float64)
    dgam = torch.empty(c, device='cuda'); dbet = torch.empty(c, device='cuda'); coef = torch.empty(3 * c, device='cuda')
    call('fte_se_bn_bwd_coef', s1, s2, dev(gate), dev(dsq), xm, dev(gamma), dev(mean), dev(rstd), dgam, dbet, coef, n, hw, c, stream())
    dyb = gv * gate[:, None, :] + dsq[:, None, :] / hw                       # gradient w.r.t. the BN output
    db_ref, dg2_ref = dyb.reshape(-1, c).sum(0), (dyb * xhat).reshape(-1, c).sum(0)
    tb, tg = np.abs(dyb).reshape(-1, c).sum(0).max(), np.abs(dyb * xhat).reshape(-1, c).sum(0).max()
    assert np.abs(host(dbet) - db_ref).max() <= 4e-6 * tb and np.abs(host(dgam) - dg2_ref).max() <= 4e-6 * tg + 2e-6 * np.abs(dsq).sum(0).max() * (1 + (np.abs(mean) * rstd).max())
    dz = torch.empty_like(out)
    call('fte_se_bn_bwd_apply', g, tdev(z), coef, dev(gate), dev(dsq), dz, n, hw, c, flags, stream())
    cnt = n * hw
    dz_ref = gamma * rstd * (dyb - db_ref / cnt - xhat * (dg2_ref / cnt))         # FusedBatchNormGrad
    lim = np.abs(dz_ref) * (2.0 ** -8 if s16 else 0) + 5e-5 * np.abs(dz_ref).max()
    assert (np.abs(thost(dz) - dz_ref) <= lim).all(), 'se_bn_bwd_apply vs the batch-norm gradient of g * gate + dsq / hw'


@pytest.mark.parametrize('rows,c', [(3 * 14 * 14, 256), (128 * 7 * 7, 1024), (50, 64), (37, 8)])
def test_residual_bn_backward_writes_the_masked_gradient_as_a_by_product(rows, c):
    """fte_bn_train_bwd_res == fte_relu_bwd followed by fte_bn_train_bwd without a mask, bit for bit: g = dy * (y > 0) for the
    shortcut, dz / dgamma / dbeta for the branch."""
    g = torch.Generator(device='cuda').manual_seed(rows + c)
    rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
    z, res, dy = rnd(rows, c), rnd(rows, c), rnd(rows, c)
    gamma, beta = rnd(c) * 0.5 + 1.0, rnd(c) * 0.3
    buf, nb = ws(query('fte_bn_ws_bytes', c))
    mean, rstd, scale, shift = [torch.empty(c, device='cuda') for _ in range(4)]
    y = torch.empty_like(z)
    call('fte_bn_train_fwd', z, gamma, beta, res, y, mean, rstd, scale, shift, None, None, rows, c, 1e-5, 0.9, 1, buf, nb, stream())
    g0 = torch.empty_like(z); dz0 = torch.empty_like(z); dg0 = torch.empty(c, device='cuda'); db0 = torch.empty(c, device='cuda')
    call('fte_relu_bwd', dy, y, g0, dy.numel(), stream())
    call('fte_bn_train_bwd', g0, None, z, gamma, mean, rstd, dz0, dg0, db0, rows, c, buf, nb, stream())
    g1 = torch.full_like(z, 7.0); dz1 = torch.empty_like(z); dg1 = torch.empty(c, device='cuda'); db1 = torch.empty(c, device='cuda')
    call('fte_bn_train_bwd_res', dy, y, z, gamma, mean, rstd, g1, dz1, dg1, db1, rows, c, buf, nb, stream())
    assert torch.equal(g0, g1) and torch.equal(dz0, dz1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


def test_batch_norm_with_the_in_launch_finalize_option():
    """FTE_BN_TAIL=1 (off by default: measured no faster, csrc/layers.hip ticket_slot): the BN statistics / backward sums are merged
    inside the producing launch by its last-arriving blocks (arrival tickets, write-through partials) instead of by a second
    launch.  Same oracle, same tolerances -- the batch-norm cases of this file re-run in a child process with the option on (the hook
    is read once per process)."""
    import os, subprocess, sys
    env = dict(os.environ, FTE_BN_TAIL='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'batch_norm and not in_launch',
                        '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
