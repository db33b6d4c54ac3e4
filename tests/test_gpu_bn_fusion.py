"""-m gpu: the "BN fusion" entry points (include/fte.h) against the float64 oracle -- the conv forward that leaves the batch statistics
of its output, the data gradient that applies the BN layer's ReLU mask and leaves its two backward sums, and the grouped-3x3 twins.
Reference graph: nets/resnet.py:47-61 (conv -> batch_norm -> relu), nets/resnext.py:34-67, nets/shufflenet_v2.py:120-135."""
import numpy as np
import pytest
import torch

from oracle import ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import call, query, dev, host, stream, ws, check_maxabs, check_rell2
    from tf_face_toolbox_amd import _lib

EPS, DECAY = 1e-3, 0.999


def _rng(s):
    return np.random.default_rng(s)


def _bf(a):
    return ops.bf16_round(np.asarray(a, np.float64))


def _dev16(a):
    """float64 array of bf16-exact values -> int16 device tensor holding the bf16 bits"""
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32).to(torch.bfloat16).view(torch.int16).cuda()


def _host16(t):
    return t.view(torch.bfloat16).float().cpu().numpy().astype(np.float64)


def _pack16(wt):
    """HWIO float64 (bf16-exact) -> (w16 HWIO pack, w16t [tap][cout][cin] pack) as the engine's fte_pack_weights_bf16 makes them"""
    k, _, cin, cout = wt.shape
    w = dev(wt)
    w16 = torch.empty(k * k * cin * cout, dtype=torch.int16, device='cuda')
    w16t = torch.empty_like(w16)
    call('fte_pack_weights_bf16', w, w16, w16t, k, cin, cout, stream())
    return w16, w16t


def _stats_ref(z, gamma, beta, mm, mv):
    c = z.shape[-1]
    rows = z.size // c
    _, cache = ops.bn_train_fwd(z, gamma, beta)
    scale = gamma * cache['rstd']
    shift = beta - cache['mean'] * scale
    mm_ref, mv_ref = ops.bn_moving_update(mm, mv, cache['mean'], cache['var'], rows)
    return cache, scale, shift, mm_ref, mv_ref


# shapes: 1x1 / 3x3, stride 1 / 2, ragged row counts, one- and many-tile launches, the 64 / 128 / 256-row tiles of the planner
FWD_SHAPES = [
    (3, 14, 14, 64, 256, 1, 1), (2, 14, 14, 256, 64, 1, 1), (3, 12, 12, 256, 512, 1, 2), (2, 9, 9, 64, 64, 3, 2),
    (5, 3, 3, 64, 64, 1, 1), (77, 4, 4, 512, 512, 1, 1), (131, 7, 7, 256, 256, 1, 1), (64, 28, 28, 128, 256, 1, 1),
    (16, 56, 56, 64, 64, 1, 1), (7, 13, 11, 128, 192, 3, 1),
]


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', FWD_SHAPES)
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_conv_bn_fwd_fp32_tensors(n, h, w, cin, cout, ks, stride, mode):
    """fp32 tensors (the metric's precision, and the 'bf16 operands' mode): z and the statistics of z in one call"""
    r = _rng(21)
    x = r.standard_normal((n, h, w, cin)) + 0.7               # non-zero mean input -> channel means well away from 0
    wt = r.standard_normal((ks, ks, cin, cout)) * 0.05
    if mode == 'bf16':
        x, wt = _bf(x), _bf(wt)
    gamma = 1 + 0.2 * r.standard_normal(cout); beta = 0.3 * r.standard_normal(cout)
    mm = r.standard_normal(cout) * 0.1; mv = 1 + 0.1 * r.random(cout)
    z_ref = ops.conv2d_fwd(x, wt, stride)
    cache, sc_ref, sh_ref, mm_ref, mv_ref = _stats_ref(z_ref, gamma, beta, mm, mv)
    z = torch.full(z_ref.shape, 7.0, device='cuda')
    mean, rstd, scale, shift = [torch.empty(cout, device='cuda') for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    wsb, nb = ws(query('fte_conv2d_bn_fwd_ws_bytes', n, h, w, cin, cout, ks, stride))
    _lib.set_mfma_dtype(mode)
    try:
        call('fte_conv2d_bn_fwd', dev(x), dev(wt), z, dev(gamma), dev(beta), mean, rstd, scale, shift, mmd, mvd, EPS, DECAY,
             None, None, None, n, h, w, cin, cout, ks, stride, 0, wsb, nb, stream())
    finally:
        _lib.set_mfma_dtype('f32')
    check_maxabs(host(z), z_ref, what='z')
    check_maxabs(host(mean), cache['mean'], 2e-6, 'mean'); check_maxabs(host(rstd), cache['rstd'], 4e-6, 'rstd')
    check_maxabs(host(scale), sc_ref, 4e-6, 'scale'); check_maxabs(host(shift), sh_ref, 1e-5, 'shift')
    check_maxabs(host(mmd), mm_ref, 2e-6, 'moving mean'); check_maxabs(host(mvd), mv_ref, 2e-6, 'moving variance')
    # the statistics are those of the STORED tensor: recomputed in float64 from the device's own z they agree tighter still
    c2, _, _, _, _ = _stats_ref(host(z), gamma, beta, mm, mv)
    check_maxabs(host(mean), c2['mean'], 1e-6, 'mean of the stored z'); check_maxabs(host(rstd), c2['rstd'], 2e-6, 'rstd of the stored z')


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', FWD_SHAPES)
def test_conv_bn_fwd_bf16_storage(n, h, w, cin, cout, ks, stride):
    """bf16 storage: z is bf16 in HBM and the statistics are those of the ROUNDED values"""
    r = _rng(22)
    x = _bf(r.standard_normal((n, h, w, cin)) + 0.7); wt = _bf(r.standard_normal((ks, ks, cin, cout)) * 0.05)
    gamma = 1 + 0.2 * r.standard_normal(cout); beta = 0.3 * r.standard_normal(cout)
    mm = r.standard_normal(cout) * 0.1; mv = 1 + 0.1 * r.random(cout)
    z_ref = ops.conv2d_fwd(x, wt, stride)
    _, w16t = _pack16(wt)
    z16 = torch.full(z_ref.shape, 0x4100, dtype=torch.int16, device='cuda')
    mean, rstd, scale, shift = [torch.empty(cout, device='cuda') for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    wsb, nb = ws(query('fte_conv2d_bn_fwd_ws_bytes', n, h, w, cin, cout, ks, stride))
    _lib.set_mfma_dtype('bf16s')
    try:
        call('fte_conv2d_bn_fwd', _dev16(x), w16t, z16, dev(gamma), dev(beta), mean, rstd, scale, shift, mmd, mvd, EPS, DECAY,
             None, None, None, n, h, w, cin, cout, ks, stride, 1, wsb, nb, stream())
    finally:
        _lib.set_mfma_dtype('f32')
    zs = _host16(z16)
    # one bf16 rounding of an fp32 accumulation of exact products: within half an ulp (2^-9 relative) plus the fp32 noise
    assert np.abs(zs - z_ref).max() <= 2.0 ** -8 * np.abs(z_ref).max()
    check_rell2(zs, _bf(z_ref), 2e-3, 'z16 vs the rounded oracle')
    cache, sc_ref, sh_ref, mm_ref, mv_ref = _stats_ref(zs, gamma, beta, mm, mv)        # of the stored values
    check_maxabs(host(mean), cache['mean'], 2e-6, 'mean'); check_maxabs(host(rstd), cache['rstd'], 4e-6, 'rstd')
    check_maxabs(host(scale), sc_ref, 4e-6, 'scale'); check_maxabs(host(shift), sh_ref, 1e-5, 'shift')
    check_maxabs(host(mmd), mm_ref, 2e-6, 'moving mean'); check_maxabs(host(mvd), mv_ref, 2e-6, 'moving variance')


DGRAD_SHAPES = [
    (3, 14, 14, 64, 256, 1, 1), (2, 14, 14, 256, 64, 1, 1), (3, 12, 12, 256, 512, 1, 2), (2, 10, 10, 64, 64, 3, 2),
    (5, 3, 3, 64, 64, 1, 1), (77, 4, 4, 512, 512, 1, 1), (64, 28, 28, 128, 256, 1, 1), (32, 56, 56, 64, 64, 1, 1),
    (7, 13, 11, 128, 192, 3, 1), (3, 9, 9, 128, 128, 3, 2),
]


def _mask_ref(kind, zbn, ybn, scale32, shift32):
    if kind == 'zmask':          # the sign of fma(z, scale, shift) in fp32 = the sign of the exact value (float64 holds the product exactly)
        return (zbn.astype(np.float32).astype(np.float64) * scale32.astype(np.float64) + shift32.astype(np.float64)) > 0
    if kind == 'ymask':
        return ybn > 0
    return np.ones(zbn.shape, bool)


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', DGRAD_SHAPES)
@pytest.mark.parametrize('kind', ['zmask', 'ymask', 'plain'])
def test_conv_dgrad_bn_fp32_tensors(n, h, w, cin, cout, ks, stride, kind):
    r = _rng(23)
    ho, wo = ops.same_pads(h, ks, stride)[0], ops.same_pads(w, ks, stride)[0]
    wt = r.standard_normal((ks, ks, cin, cout)) * 0.05
    dz = r.standard_normal((n, ho, wo, cout))
    addin = r.standard_normal((n, h, w, cin)) if kind != 'plain' else None
    zbn = (r.standard_normal((n, h, w, cin)) * 1.5 + 0.4).astype(np.float32).astype(np.float64)
    gamma = 1 + 0.2 * r.standard_normal(cin); beta = 0.3 * r.standard_normal(cin)
    bn_ref, cache = ops.bn_train_fwd(zbn, gamma, beta)
    mean32 = cache['mean'].astype(np.float32); rstd32 = cache['rstd'].astype(np.float32)
    scale32 = (gamma * cache['rstd']).astype(np.float32); shift32 = (beta - cache['mean'] * gamma * cache['rstd']).astype(np.float32)
    res = r.standard_normal(zbn.shape)
    ybn = np.maximum(bn_ref + res, 0).astype(np.float32).astype(np.float64) if kind == 'ymask' else None
    dy_ref = ops.conv2d_bwd(np.zeros((n, h, w, cin)), wt, dz, stride, need_dw=False)[0]
    if addin is not None:
        dy_ref = dy_ref + addin
    g_ref = dy_ref * _mask_ref(kind, zbn, ybn, scale32, shift32)
    cache32 = dict(cache, mean=mean32.astype(np.float64), rstd=rstd32.astype(np.float64),
                   xhat=(zbn - mean32.astype(np.float64)) * rstd32.astype(np.float64))
    dz_ref, dg_ref, db_ref = ops.bn_train_bwd(g_ref, gamma, cache32)
    g = torch.full(zbn.shape, 7.0, device='cuda')
    dgam, dbet = torch.empty(cin, device='cuda'), torch.empty(cin, device='cuda')
    coef = torch.empty(3 * cin, device='cuda')
    wsb, nb = ws(query('fte_conv2d_dgrad_bn_ws_bytes', n, h, w, cin, cout, ks, stride))
    zd = dev(zbn)
    call('fte_conv2d_dgrad_bn', dev(dz), dev(wt), dev(addin) if addin is not None else None, zd, dev(ybn) if ybn is not None else None,
         dev(gamma), dev(mean32), dev(rstd32), dev(scale32) if kind == 'zmask' else None, dev(shift32) if kind == 'zmask' else None,
         g, dgam, dbet, coef, n, h, w, cin, cout, ks, stride, 0, wsb, nb, stream())
    check_maxabs(host(g), g_ref, what='masked gradient')
    check_rell2(host(dgam), dg_ref, what='dgamma'); check_rell2(host(dbet), db_ref, what='dbeta')
    dzp = torch.empty(zbn.shape, device='cuda')
    call('fte_bn_bwd_apply', g, zd, coef, dzp, n * h * w, cin, 0, stream())
    check_maxabs(host(dzp), dz_ref, what='dz of the BN layer')


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', DGRAD_SHAPES)
@pytest.mark.parametrize('kind', ['zmask', 'ymask'])
def test_conv_dgrad_bn_bf16_storage(n, h, w, cin, cout, ks, stride, kind):
    r = _rng(24)
    ho, wo = ops.same_pads(h, ks, stride)[0], ops.same_pads(w, ks, stride)[0]
    wt = _bf(r.standard_normal((ks, ks, cin, cout)) * 0.05)
    dz = _bf(r.standard_normal((n, ho, wo, cout)))
    addin = _bf(r.standard_normal((n, h, w, cin)))
    zbn = _bf(r.standard_normal((n, h, w, cin)) * 1.5 + 0.4)
    gamma = 1 + 0.2 * r.standard_normal(cin); beta = 0.3 * r.standard_normal(cin)
    bn_ref, cache = ops.bn_train_fwd(zbn, gamma, beta)
    mean32 = cache['mean'].astype(np.float32); rstd32 = cache['rstd'].astype(np.float32)
    scale32 = (gamma * cache['rstd']).astype(np.float32); shift32 = (beta - cache['mean'] * gamma * cache['rstd']).astype(np.float32)
    ybn = _bf(np.maximum(bn_ref + _bf(r.standard_normal(zbn.shape)), 0)) if kind == 'ymask' else None
    dy_ref = ops.conv2d_bwd(np.zeros((n, h, w, cin)), wt, dz, stride, need_dw=False)[0] + addin
    mask = _mask_ref(kind, zbn, ybn, scale32, shift32)
    w16, _ = _pack16(wt)
    g16 = torch.full(zbn.shape, 0x4100, dtype=torch.int16, device='cuda')
    dgam, dbet = torch.empty(cin, device='cuda'), torch.empty(cin, device='cuda')
    coef = torch.empty(3 * cin, device='cuda')
    wsb, nb = ws(query('fte_conv2d_dgrad_bn_ws_bytes', n, h, w, cin, cout, ks, stride))
    z16 = _dev16(zbn)
    _lib.set_mfma_dtype('bf16s')
    try:
        call('fte_conv2d_dgrad_bn', _dev16(dz), w16, _dev16(addin), z16, _dev16(ybn) if ybn is not None else None,
             dev(gamma), dev(mean32), dev(rstd32), dev(scale32) if kind == 'zmask' else None, dev(shift32) if kind == 'zmask' else None,
             g16, dgam, dbet, coef, n, h, w, cin, cout, ks, stride, 1, wsb, nb, stream())
    finally:
        _lib.set_mfma_dtype('f32')
    gs = _host16(g16)
    assert np.array_equal(gs != 0, (gs != 0) & mask), 'a masked element was written non-zero'
    assert np.abs(gs - dy_ref * mask).max() <= 2.0 ** -8 * np.abs(dy_ref).max()
    check_rell2(gs, _bf(dy_ref) * mask, 2e-3, 'g16 vs the rounded oracle')
    # the sums are those of the STORED (rounded) masked gradient
    xhat = (zbn - mean32.astype(np.float64)) * rstd32.astype(np.float64)
    check_rell2(host(dbet), gs.reshape(-1, cin).sum(0), what='dbeta'); check_rell2(host(dgam), (gs * xhat).reshape(-1, cin).sum(0), what='dgamma')
    cache32 = dict(cache, mean=mean32.astype(np.float64), rstd=rstd32.astype(np.float64), xhat=xhat)
    dz_ref = ops.bn_train_bwd(gs, gamma, cache32)[0]
    dzp = torch.empty(zbn.shape, dtype=torch.int16, device='cuda')
    call('fte_bn_bwd_apply', g16, z16, coef, dzp, n * h * w, cin, 3, stream())
    got = _host16(dzp)
    assert np.abs(got - dz_ref).max() <= 2.0 ** -8 * np.abs(dz_ref).max() + 1e-6
    check_rell2(got, _bf(dz_ref), 3e-3, 'dz16 vs the rounded oracle')


GCONV_SHAPES = [(3, 14, 14, 128, 32, 1), (2, 9, 7, 256, 32, 1), (2, 28, 28, 128, 32, 2), (5, 14, 14, 512, 32, 2), (16, 28, 28, 256, 32, 1),
                (1, 5, 5, 1024, 32, 1)]


@pytest.mark.parametrize('n,h,w,c,groups,stride', GCONV_SHAPES)
def test_gconv_bn_fwd_and_dgrad_bf16_storage(n, h, w, c, groups, stride):
    """The grouped 3x3 of ResNeXt (nets/resnext.py:41-51) with the statistics / the BN mask and sums in its epilogue: outputs BIT-equal
    to the plain kernel's (masked), statistics and sums against float64 sums of the stored values."""
    r = _rng(25)
    gw = c // groups
    ho, wo = ops.same_pads(h, 3, stride)[0], ops.same_pads(w, 3, stride)[0]
    x = _bf(r.standard_normal((n, h, w, c)) + 0.5)
    wg = r.standard_normal((groups, 3, 3, gw, gw)) * 0.1
    wf = torch.empty((c // 32) * 9 * 1024, dtype=torch.int16, device='cuda'); wd_ = torch.empty_like(wf)
    call('fte_gconv3x3_pack_bf16', dev(wg), wf, wd_, c, groups, stream())
    gamma = 1 + 0.2 * r.standard_normal(c); beta = 0.3 * r.standard_normal(c)
    mm = r.standard_normal(c) * 0.1; mv = 1 + 0.1 * r.random(c)
    x16 = _dev16(x)
    z_plain = torch.empty((n, ho, wo, c), dtype=torch.int16, device='cuda')
    call('fte_gconv3x3_bf16_s16', x16, wf, z_plain, n, h, w, c, stride, 0, stream())
    z16 = torch.full((n, ho, wo, c), 0x4100, dtype=torch.int16, device='cuda')
    mean, rstd, scale, shift = [torch.empty(c, device='cuda') for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    wsb, nb = ws(query('fte_gconv3x3_bn_ws_bytes', n, h, w, c, stride))
    call('fte_gconv3x3_bn_fwd_bf16_s16', x16, wf, z16, dev(gamma), dev(beta), mean, rstd, scale, shift, mmd, mvd, EPS, DECAY,
         None, None, None, n, h, w, c, stride, wsb, nb, stream())
    assert torch.equal(z16, z_plain)
    zs = _host16(z16)
    cache, sc_ref, sh_ref, mm_ref, mv_ref = _stats_ref(zs, gamma, beta, mm, mv)
    check_maxabs(host(mean), cache['mean'], 2e-6, 'mean'); check_maxabs(host(rstd), cache['rstd'], 4e-6, 'rstd')
    check_maxabs(host(scale), sc_ref, 4e-6, 'scale'); check_maxabs(host(shift), sh_ref, 1e-5, 'shift')
    check_maxabs(host(mmd), mm_ref, 2e-6, 'moving mean'); check_maxabs(host(mvd), mv_ref, 2e-6, 'moving variance')
    # data gradient landing on BN + ReLU of the layer's INPUT tensor
    zbn = _bf(r.standard_normal((n, h, w, c)) * 1.5 + 0.4)
    g1 = 1 + 0.2 * r.standard_normal(c); b1 = 0.3 * r.standard_normal(c)
    _, c1 = ops.bn_train_fwd(zbn, g1, b1)
    mean32 = c1['mean'].astype(np.float32); rstd32 = c1['rstd'].astype(np.float32)
    scale32 = (g1 * c1['rstd']).astype(np.float32); shift32 = (b1 - c1['mean'] * g1 * c1['rstd']).astype(np.float32)
    dz16 = _dev16(_bf(r.standard_normal((n, ho, wo, c))))
    dx_plain = torch.empty((n, h, w, c), dtype=torch.int16, device='cuda')
    call('fte_gconv3x3_bf16_s16', dz16, wd_, dx_plain, n, h, w, c, stride, 1, stream())
    for kind in ('zmask', 'plain'):
        g16 = torch.full((n, h, w, c), 0x4100, dtype=torch.int16, device='cuda')
        dgam, dbet, coef = torch.empty(c, device='cuda'), torch.empty(c, device='cuda'), torch.empty(3 * c, device='cuda')
        zb16 = _dev16(zbn)
        call('fte_gconv3x3_dgrad_bn_bf16_s16', dz16, wd_, zb16, dev(g1), dev(mean32), dev(rstd32),
             dev(scale32) if kind == 'zmask' else None, dev(shift32) if kind == 'zmask' else None, g16, dgam, dbet, coef,
             n, h, w, c, stride, wsb, nb, stream())
        mask = _mask_ref(kind, zbn, None, scale32, shift32)
        gs = _host16(g16)
        assert np.array_equal(gs, _host16(dx_plain) * mask), kind
        xhat = (zbn - mean32.astype(np.float64)) * rstd32.astype(np.float64)
        check_rell2(host(dbet), gs.reshape(-1, c).sum(0), what='dbeta'); check_rell2(host(dgam), (gs * xhat).reshape(-1, c).sum(0), what='dgamma')
        cache32 = dict(c1, mean=mean32.astype(np.float64), rstd=rstd32.astype(np.float64), xhat=xhat)
        dz_ref = ops.bn_train_bwd(gs, g1, cache32)[0]
        dzp = torch.empty((n, h, w, c), dtype=torch.int16, device='cuda')
        call('fte_bn_bwd_apply', g16, zb16, coef, dzp, n * h * w, c, 3, stream())
        assert np.abs(_host16(dzp) - dz_ref).max() <= 2.0 ** -8 * np.abs(dz_ref).max() + 1e-6


def test_bn_apply_matches_the_train_forward():
    """fte_bn_apply with the coefficients of fte_bn_train_fwd reproduces its output bit for bit (fp32 and bf16 storage)"""
    r = _rng(26)
    shape, c = (6, 9, 7), 96
    z = r.standard_normal(shape + (c,)) * 2 + 1; res = r.standard_normal(shape + (c,))
    gamma = 1 + 0.2 * r.standard_normal(c); beta = 0.3 * r.standard_normal(c)
    rows = int(np.prod(shape))
    mean, rstd, scale, shift = [torch.empty(c, device='cuda') for _ in range(4)]
    wsb, nb = ws(query('fte_bn_ws_bytes', c))
    y1 = torch.empty(shape + (c,), device='cuda'); y2 = torch.empty_like(y1)
    zd, rd = dev(z), dev(res)
    call('fte_bn_train_fwd', zd, dev(gamma), dev(beta), rd, y1, mean, rstd, scale, shift, None, None, rows, c, EPS, DECAY, 1, wsb, nb, stream())
    call('fte_bn_apply', zd, scale, shift, rd, y2, rows, c, 1, 0, stream())
    assert torch.equal(y1, y2)
    z16, r16 = _dev16(_bf(z)), _dev16(_bf(res))
    y3 = torch.empty(shape + (c,), dtype=torch.int16, device='cuda'); y4 = torch.empty_like(y3)
    call('fte_bn_train_fwd_s16', z16, dev(gamma), dev(beta), r16, y3, mean, rstd, scale, shift, None, None, rows, c, EPS, DECAY, 1, 3, wsb, nb, stream())
    call('fte_bn_apply', z16, scale, shift, r16, y4, rows, c, 1, 3, stream())
    assert torch.equal(y3, y4)


FOLD_SHAPES = [(3, 14, 14, 64, 256), (2, 9, 7, 128, 128), (64, 28, 28, 128, 256), (131, 7, 7, 256, 512), (5, 3, 3, 256, 64), (33, 28, 28, 64, 64)]


@pytest.mark.parametrize('n,h,w,cin,cout', FOLD_SHAPES)
def test_conv_bn_fwd_folds_the_normalise_pass_of_the_bn_in_front(n, h, w, cin, cout):
    """conv -> BN -> ReLU -> conv (nets/resnet.py:47-61): the second conv takes the first BN's PRE-normalisation tensor and its scale /
    shift; z, the statistics and the side-stored y must equal, bit for bit, fte_bn_apply followed by the plain fused conv"""
    r = _rng(27)
    assert query('fte_conv2d_bn_fwd_folds', n, h, w, cin, cout, 1, 1, 1) == 1
    zp = _bf(r.standard_normal((n, h, w, cin)) * 1.5 + 0.3)
    isc = (1 + 0.2 * r.standard_normal(cin)).astype(np.float32); ish = (0.3 * r.standard_normal(cin)).astype(np.float32)
    wt = _bf(r.standard_normal((1, 1, cin, cout)) * 0.05)
    gamma = 1 + 0.2 * r.standard_normal(cout); beta = 0.3 * r.standard_normal(cout)
    _, w16t = _pack16(wt)
    zp16 = _dev16(zp)
    rows = n * h * w
    y_ref = torch.empty((n, h, w, cin), dtype=torch.int16, device='cuda')
    call('fte_bn_apply', zp16, dev(isc), dev(ish), None, y_ref, rows, cin, 1, 3, stream())
    wsb, nb = ws(query('fte_conv2d_bn_fwd_ws_bytes', n, h, w, cin, cout, 1, 1))
    outs = []
    _lib.set_mfma_dtype('bf16s')
    try:
        for fold in (False, True):
            z16 = torch.full((n, h, w, cout), 0x4100, dtype=torch.int16, device='cuda')
            st = [torch.empty(cout, device='cuda') for _ in range(4)]
            mm, mv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
            ys = torch.full((n, h, w, cin), 0x4100, dtype=torch.int16, device='cuda')
            if fold:
                call('fte_conv2d_bn_fwd', zp16, w16t, z16, dev(gamma), dev(beta), st[0], st[1], st[2], st[3], mm, mv, EPS, DECAY,
                     dev(isc), dev(ish), ys, n, h, w, cin, cout, 1, 1, 1, wsb, nb, stream())
            else:
                call('fte_conv2d_bn_fwd', y_ref, w16t, z16, dev(gamma), dev(beta), st[0], st[1], st[2], st[3], mm, mv, EPS, DECAY,
                     None, None, None, n, h, w, cin, cout, 1, 1, 1, wsb, nb, stream())
            outs.append((z16, st, mm, mv, ys))
    finally:
        _lib.set_mfma_dtype('f32')
    assert torch.equal(outs[1][4], y_ref), 'side-stored y differs from fte_bn_apply'
    assert torch.equal(outs[0][0], outs[1][0]), 'z differs'
    for a, b in zip(outs[0][1] + [outs[0][2], outs[0][3]], outs[1][1] + [outs[1][2], outs[1][3]]):
        assert torch.equal(a, b)
    # and against the oracle: y = relu(isc * zp + ish) rounded, z = y * w
    y64 = _bf(np.maximum(zp * isc.astype(np.float64) + ish.astype(np.float64), 0))
    check_rell2(_host16(outs[1][4]), y64, 2e-3, 'y vs the oracle')
    check_rell2(_host16(outs[1][0]), _bf(ops.conv2d_fwd(_host16(outs[1][4]), wt, 1)), 2e-3, 'z vs the oracle')


@pytest.mark.parametrize('n,h,w,c,groups', [(3, 14, 14, 128, 32), (2, 9, 7, 256, 32), (16, 28, 28, 128, 32), (1, 5, 5, 1024, 32)])
def test_gconv_bn_fwd_folds_the_normalise_pass_of_the_bn_in_front(n, h, w, c, groups):
    r = _rng(28)
    gw = c // groups
    zp = _bf(r.standard_normal((n, h, w, c)) * 1.5 + 0.3)
    isc = (1 + 0.2 * r.standard_normal(c)).astype(np.float32); ish = (0.3 * r.standard_normal(c)).astype(np.float32)
    wg = r.standard_normal((groups, 3, 3, gw, gw)) * 0.1
    wf = torch.empty((c // 32) * 9 * 1024, dtype=torch.int16, device='cuda'); wd_ = torch.empty_like(wf)
    call('fte_gconv3x3_pack_bf16', dev(wg), wf, wd_, c, groups, stream())
    gamma = 1 + 0.2 * r.standard_normal(c); beta = 0.3 * r.standard_normal(c)
    zp16 = _dev16(zp)
    rows = n * h * w
    y_ref = torch.empty((n, h, w, c), dtype=torch.int16, device='cuda')
    call('fte_bn_apply', zp16, dev(isc), dev(ish), None, y_ref, rows, c, 1, 3, stream())
    wsb, nb = ws(query('fte_gconv3x3_bn_ws_bytes', n, h, w, c, 1))
    outs = []
    for fold in (False, True):
        z16 = torch.full((n, h, w, c), 0x4100, dtype=torch.int16, device='cuda')
        st = [torch.empty(c, device='cuda') for _ in range(4)]
        ys = torch.full((n, h, w, c), 0x4100, dtype=torch.int16, device='cuda')
        call('fte_gconv3x3_bn_fwd_bf16_s16', zp16 if fold else y_ref, wf, z16, dev(gamma), dev(beta), st[0], st[1], st[2], st[3], None, None,
             EPS, DECAY, dev(isc) if fold else None, dev(ish) if fold else None, ys if fold else None, n, h, w, c, 1, wsb, nb, stream())
        outs.append((z16, st, ys))
    assert torch.equal(outs[1][2], y_ref), 'side-stored y differs from fte_bn_apply'
    assert torch.equal(outs[0][0], outs[1][0]), 'z differs'
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
