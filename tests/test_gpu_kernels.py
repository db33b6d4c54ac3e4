"""-m gpu: every C-ABI entry point against the float64 oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import call, query, dev, host, stream, ws, check_maxabs, check_rell2


def _rng(seed):
    return np.random.default_rng(seed)


CONV_CASES = [
    # n, h, w, cin, cout, stride
    (2, 14, 14, 64, 64, 1),      # 256x64 / small tile paths
    (3, 9, 7, 64, 128, 1),       # odd, non-square, ragged M
    (2, 12, 12, 64, 128, 2),     # stage-entry conv, asymmetric SAME (0,1)
    (2, 7, 7, 128, 256, 2),      # odd size at stride 2: pads (1,1)
    (1, 28, 28, 128, 128, 1),
    (2, 6, 6, 256, 512, 2),
    (5, 7, 7, 512, 512, 1),
    (16, 28, 28, 128, 128, 1),   # enough rows for the 128x128 tile
    (84, 28, 28, 128, 128, 1),   # 515 big tiles: one whole round + a 3-tile split-K tail
    (40, 14, 14, 256, 256, 1),   # 124 big tiles (< one round): every tile split-K x4 + fix-up
    (66, 28, 28, 64, 128, 2),    # stride-2 dgrad classes with split-K tails
]


@pytest.mark.parametrize('n,h,w,cin,cout,stride', CONV_CASES)
def test_conv3x3_fwd(n, h, w, cin, cout, stride):
    r = _rng(1)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((3, 3, cin, cout)) * 0.05
    b = r.standard_normal(cout); al = 0.25 + 0.1 * r.standard_normal(cout)
    z_ref = ops.conv2d_fwd(x, wt, stride, b)
    res = r.standard_normal(z_ref.shape)
    y_ref = ops.prelu_fwd(z_ref, al) + res
    z = torch.empty(z_ref.shape, device='cuda'); y = torch.empty_like(z)
    wsb, nb = ws(query('fte_conv3x3_fwd_ws_bytes', n, h, w, cin, cout, stride))
    call('fte_conv3x3_fwd', dev(x), dev(wt), dev(b), dev(al), dev(res), z, y, n, h, w, cin, cout, stride, wsb, nb, stream())
    check_maxabs(host(z), z_ref, what='z'); check_maxabs(host(y), y_ref, what='y')
    # no bias / no activation / no residual / no z
    y2 = torch.empty_like(z)
    call('fte_conv3x3_fwd', dev(x), dev(wt), None, None, None, None, y2, n, h, w, cin, cout, stride, None, 0, stream())   # no workspace: small-tile path
    check_maxabs(host(y2), ops.conv2d_fwd(x, wt, stride), what='plain')


@pytest.mark.parametrize('n,h,w,cin,cout,stride', CONV_CASES)
def test_conv3x3_dgrad_with_prelu_backward(n, h, w, cin, cout, stride):
    r = _rng(2)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((3, 3, cin, cout)) * 0.05
    zshape = ops.conv2d_fwd(x, wt, stride).shape
    dz = r.standard_normal(zshape)
    dx_ref, _ = ops.conv2d_bwd(x, wt, dz, stride)
    addin = r.standard_normal(x.shape); zprev = r.standard_normal(x.shape); alp = 0.25 + 0.1 * r.standard_normal(cin)
    zprev[0, 0, 0, :4] = 0.0                                  # exercise the z == 0 sub-gradient
    g_ref = dx_ref + addin
    dzprev_ref, dalpha_ref = ops.prelu_bwd(zprev, alp, g_ref)
    dbias_ref = dzprev_ref.sum(axis=(0, 1, 2))
    raw = torch.empty(x.shape, device='cuda'); dzp = torch.empty_like(raw)
    da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_dgrad_ws_bytes', n, h, w, cin, cout, stride))
    call('fte_conv3x3_dgrad', dev(dz), dev(wt), dev(addin), dev(zprev), dev(alp), raw, dzp, da, db,
         n, h, w, cin, cout, stride, wsb, nb, stream())
    check_maxabs(host(raw), g_ref, what='raw'); check_maxabs(host(dzp), dzprev_ref, what='dzprev')
    check_rell2(host(da), dalpha_ref, what='dalpha'); check_rell2(host(db), dbias_ref, what='dbias')
    # plain dgrad (no fusion)
    dzp2 = torch.empty_like(raw)
    call('fte_conv3x3_dgrad', dev(dz), dev(wt), None, None, None, None, dzp2, None, None,
         n, h, w, cin, cout, stride, wsb, nb, stream())
    check_maxabs(host(dzp2), dx_ref, what='plain dgrad')


@pytest.mark.parametrize('n,h,w,cin,cout,stride', CONV_CASES)
def test_conv3x3_wgrad(n, h, w, cin, cout, stride):
    r = _rng(3)
    x = r.standard_normal((n, h, w, cin)); wt = np.zeros((3, 3, cin, cout))
    dz = r.standard_normal(ops.conv2d_fwd(x, wt, stride).shape)
    _, dw_ref = ops.conv2d_bwd(x, wt, dz, stride, need_dx=False)
    dw = torch.empty(3, 3, cin, cout, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_wgrad_ws_bytes', n, h, w, cin, cout, stride))
    call('fte_conv3x3_wgrad', dev(x), dev(dz), dw, n, h, w, cin, cout, stride, wsb, nb, stream())
    check_maxabs(host(dw), dw_ref, what='dw')


@pytest.mark.parametrize('n,h,w,cin,cout,stride', [(3, 16, 16, 3, 64, 2), (2, 13, 9, 1, 64, 2), (2, 112, 112, 3, 64, 2),
                                                   (3, 16, 16, 3, 32, 2), (2, 13, 9, 1, 32, 1), (2, 112, 112, 3, 32, 2)])
def test_first_conv_fwd_and_wgrad(n, h, w, cin, cout, stride):
    """cout 64: SphereNet's first conv (bias + PReLU fused); cout 32: ShuffleNet-v2's 24-filter stem stored 32 wide (plain)."""
    r = _rng(4)
    x = r.uniform(-1, 1, (n, h, w, cin)); wt = r.standard_normal((3, 3, cin, cout)) * 0.2
    b = r.standard_normal(cout); al = 0.25 + 0.1 * r.standard_normal(cout)
    z_ref = ops.conv2d_fwd(x, wt, stride, b); y_ref = ops.prelu_fwd(z_ref, al)
    z = torch.empty(z_ref.shape, device='cuda'); y = torch.empty_like(z)
    call('fte_conv3x3_first_fwd', dev(x), dev(wt), dev(b), dev(al), z, y, n, h, w, cin, cout, stride, stream())
    check_maxabs(host(z), z_ref, what='z'); check_maxabs(host(y), y_ref, what='y')
    call('fte_conv3x3_first_fwd', dev(x), dev(wt), None, None, None, y, n, h, w, cin, cout, stride, stream())
    check_maxabs(host(y), ops.conv2d_fwd(x, wt, stride), what='plain')
    dz = r.standard_normal(z_ref.shape)
    _, dw_ref = ops.conv2d_bwd(x, wt, dz, stride, need_dx=False)
    dw = torch.empty(3, 3, cin, cout, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_first_wgrad_ws_bytes', n, h, w, cin, cout, stride))
    call('fte_conv3x3_first_wgrad', dev(x), dev(dz), dw, n, h, w, cin, cout, stride, wsb, nb, stream())
    check_maxabs(host(dw), dw_ref, what='dw')


@pytest.mark.parametrize('m,n,k', [(128, 128, 256), (128, 1024, 2048), (7, 64, 96), (2048, 4096, 64)])
def test_dense_with_activation(m, n, k):
    """fte_gemm_nn_act: y = act(x @ w + bias) with the activation in the pass that writes y (the slab reduction of the split plan,
    or a pass of its own behind an unsplit launch): the squeeze-excitation gate's two dense layers."""
    r = _rng(9)
    x = r.standard_normal((m, k)); w = r.standard_normal((k, n)) * 0.1; b = r.standard_normal(n)
    wsb, nb = ws(query('fte_gemm_ws_bytes', m, n, k))
    lin = x @ w + b
    for act, ref in ((0, lin), (1, np.maximum(lin, 0.0)), (2, 1.0 / (1.0 + np.exp(-lin)))):
        y = torch.full((m, n), 7.0, device='cuda')
        call('fte_gemm_nn_act', dev(x), dev(w), dev(b), y, m, n, k, act, wsb, nb, stream())
        check_maxabs(host(y), ref, what='act %d' % act)
    assert query('fte_gemm_nn_act', dev(x).data_ptr(), dev(w).data_ptr(), 0, y.data_ptr(), m, n, k, 3, wsb.data_ptr(), nb, 0) != 0


@pytest.mark.parametrize('m,n,k', [(4, 512, 2048), (64, 512, 25088), (3, 128, 512), (70, 10624, 512), (512, 512, 512)])
def test_dense_nn_nt_tn(m, n, k):
    r = _rng(5)
    x = r.standard_normal((m, k)); w = r.standard_normal((k, n)) * 0.05; b = r.standard_normal(n)
    wsb, nb = ws(max(query('fte_gemm_ws_bytes', m, n, k), query('fte_gemm_ws_bytes', m, k, n)))
    y = torch.empty(m, n, device='cuda')
    call('fte_gemm_nn', dev(x), dev(w), dev(b), y, m, n, k, wsb, nb, stream())
    check_maxabs(host(y), x @ w + b, what='nn')
    dy = r.standard_normal((m, n))
    if k % 64 == 0:
        dx = torch.empty(m, k, device='cuda')
        call('fte_gemm_nt', dev(dy), dev(w), None, None, 0, None, dx, None, m, n, k, wsb, nb, stream())
        check_maxabs(host(dx), dy @ w.T, what='nt')
        # fused PReLU backward with per-(column % amod) alpha
        amod = 64
        zp = r.standard_normal((m, k)); al = 0.25 + 0.1 * r.standard_normal(amod)
        g_ref = dy @ w.T
        alf = np.tile(al, k // amod)
        dzp_ref, dal_full = ops.prelu_bwd(zp, alf, g_ref)
        raw = torch.empty(m, k, device='cuda'); da = torch.empty(amod, device='cuda')
        call('fte_gemm_nt', dev(dy), dev(w), dev(zp), dev(al), amod, raw, dx, da, m, n, k, wsb, nb, stream())
        check_maxabs(host(raw), g_ref, what='nt raw'); check_maxabs(host(dx), dzp_ref, what='nt masked')
        check_rell2(host(da), dal_full.reshape(-1, amod).sum(0), what='nt dalpha')
    dw = torch.empty(k, n, device='cuda')
    call('fte_gemm_tn', dev(x), dev(dy), dw, m, n, k, wsb, nb, stream())
    check_maxabs(host(dw), x.T @ dy, what='tn')


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('m,n,k', [(128, 128, 256), (128, 256, 128), (37, 64, 128), (128, 1024, 2048), (5, 32, 384), (256, 2048, 1024), (128, 128, 2048), (128, 64, 1024), (64, 128, 512), (128, 96, 1536)])
def test_dense_small_one_launch_layers(m, n, k, mode):
    """fte_dense_small: the squeeze-excitation gate's dense layers (nets/shufflenet_v2.py:79-85) in one launch -- forward with bias and
    ReLU / sigmoid, the gradient w.r.t. the input through w^T with the ReLU mask; ragged m (37, 5 rows: part of a 32-row tile); fp32
    operands, and bf16 operands against the oracle evaluated on the bf16-rounded operands (fp32 accumulation both ways)."""
    from tf_face_toolbox_amd import _lib
    r = _rng(31)
    a = r.standard_normal((m, k)); w = r.standard_normal((k, n)) * 0.05; b = r.standard_normal(n)
    wt = r.standard_normal((n, k)) * 0.05; msk = r.standard_normal((m, n))
    rd = ops.bf16_round if mode == 'bf16' else (lambda v: v)
    a32, w32, wt32 = (np.asarray(v, np.float32).astype(np.float64) for v in (a, w, wt))
    _lib.set_mfma_dtype(mode)
    try:
        for act in (0, 1, 2):
            out = torch.empty(m, n, device='cuda')
            call('fte_dense_small', dev(a), dev(w), dev(b), None, out, m, n, k, 0, act, stream())
            ref = rd(a32) @ rd(w32) + np.asarray(b, np.float32)
            ref = np.maximum(ref, 0) if act == 1 else (1 / (1 + np.exp(-ref)) if act == 2 else ref)
            check_maxabs(host(out), ref, 2e-5, 'dense_small nn act %d' % act)
        out = torch.empty(m, n, device='cuda')
        call('fte_dense_small', dev(a), dev(wt), None, dev(msk), out, m, n, k, 1, 0, stream())
        ref = (rd(a32) @ rd(wt32).T) * (np.asarray(msk, np.float32) > 0)
        check_maxabs(host(out), ref, 2e-5, 'dense_small nt with the ReLU mask')
        out2 = torch.empty(m, n, device='cuda')
        call('fte_dense_small', dev(a), dev(wt), None, dev(msk), out2, m, n, k, 1, 0, stream())
        assert torch.equal(out, out2)                                        # fixed summation order
    finally:
        _lib.set_mfma_dtype('f32')
    from tf_face_toolbox_amd._lib import FteError
    with pytest.raises(FteError):
        call('fte_dense_small', dev(a), dev(w), None, None, out, m, n, k + 64, 0, 0, stream())      # k % 128 != 0


@pytest.mark.parametrize('n,c', [(5, 10), (64, 10575), (3, 129)])
def test_softmax_ce(n, c):
    r = _rng(6)
    ld = (c + 127) // 128 * 128
    z = np.zeros((n, ld)); z[:, :c] = r.standard_normal((n, c)) * 4; z[:, c:] = 77.0   # pad garbage must be ignored
    y = r.integers(0, c, n)
    loss_ref, d_ref = ops.softmax_ce(z[:, :c], y, 0.37 / n)
    lr_ = torch.empty(n, device='cuda'); d = torch.empty(n, ld, device='cuda')
    call('fte_softmax_ce_fwd_bwd', dev(z), dev(y, torch.int32), lr_, d, n, c, ld, 0.37 / n, stream())
    assert abs(host(lr_).mean() - loss_ref) <= 1e-5 * max(1, abs(loss_ref))
    check_maxabs(host(d)[:, :c], d_ref, what='dlogits')
    assert (host(d)[:, c:] == 0).all()


@pytest.mark.parametrize('lam', [1000.0, 5.0])
def test_asoftmax_head(lam):
    r = _rng(7)
    n, dd, c = 24, 512, 300
    ld = 384
    x = r.standard_normal((n, dd)); w = r.standard_normal((dd, c)) * 0.05; y = r.integers(0, c, n)
    # force every k-branch of psi: align some rows with (or against) their target column
    for i, t in enumerate([0.95, 0.5, -0.3, -0.9]):
        v = w[:, y[i]] / np.linalg.norm(w[:, y[i]])
        o = r.standard_normal(dd); o -= o.dot(v) * v; o /= np.linalg.norm(o)
        x[i] = 3.0 * (t * v + np.sqrt(1 - t * t) * o)
    loss_ref, f_ref, dx_ref, dw_ref = ops.asoftmax_fwd_bwd(x, w, y, lam, 1.0 / n)
    wp = np.zeros((dd, ld)); wp[:, :c] = w
    s = x @ wp
    xn = torch.empty(n, device='cuda'); wn = torch.empty(ld, device='cuda')
    call('fte_row_norms', dev(x), xn, n, dd, dd, stream())
    call('fte_col_norms', dev(wp), wn, dd, c, ld, stream())
    check_maxabs(host(xn), np.linalg.norm(x, axis=1), 1e-6, 'xn'); check_maxabs(host(wn)[:c], np.linalg.norm(w, axis=0), 1e-6, 'wn')
    f = torch.empty(n, ld, device='cuda'); G = torch.empty(n, ld, device='cuda')
    lrows = torch.empty(n, device='cuda'); rcf = torch.empty(n, device='cuda'); ccf = torch.empty(ld, device='cuda')
    sd = dev(s)
    call('fte_asoftmax_fwd_bwd', sd, xn, wn, dev(y, torch.int32), lam, f, lrows, G, rcf, n, c, ld, 1.0 / n, stream())
    call('fte_asoftmax_colcoef', G, sd, wn, ccf, n, c, ld, stream())
    check_maxabs(host(f)[:, :c], f_ref, 2e-5, 'margin logits')
    assert abs(host(lrows).mean() - loss_ref) <= 1e-5 * max(1, abs(loss_ref))
    Gh, rch, cch = host(G), host(rcf), host(ccf)
    dx = Gh @ wp.T + rch[:, None] * x
    dw = (x.T @ Gh + cch[None, :] * wp)[:, :c]
    check_rell2(dx, dx_ref, 2e-5, 'dx'); check_rell2(dw, dw_ref, 2e-5, 'dw')
    # the fix-up kernel itself
    a = r.standard_normal((n, ld)); bb = r.standard_normal((n, ld)); rc = r.standard_normal(n); cc = r.standard_normal(ld)
    at = dev(a)
    call('fte_add_scaled_rows_cols', at, dev(bb), dev(rc), dev(cc), n, ld, ld, stream())
    check_maxabs(host(at), a + rc[:, None] * bb + cc[None, :] * bb, 1e-6, 'add_scaled')


def test_center_loss_and_triplet():
    r = _rng(8)
    n, d, c = 32, 512, 11
    f = r.standard_normal((n, d)); y = r.integers(0, c, n); cen = r.standard_normal((c, d)) * 0.1
    loss_ref, df_ref, newc_ref = ops.center_loss(f, y, cen, 0.99)
    cd = dev(cen); lrows = torch.empty(n, device='cuda'); df = torch.empty(n, d, device='cuda')
    wsb, nb = ws(n * d * 4)
    call('fte_center_loss_fwd_bwd_update', dev(f), dev(y, torch.int32), cd, lrows, df, n, d, c, 0.99, 1.0 / (n * d), wsb, nb, stream())
    assert abs(host(lrows).sum() / (n * d) - loss_ref) <= 1e-5 * loss_ref
    check_maxabs(host(df), df_ref, 1e-5, 'dfeat'); check_maxabs(host(cd), newc_ref, 1e-5, 'centers')
    for margin in (None, 0.3, -1.0):
        yk = np.repeat(np.arange(8), 4); yk[-1] = 99                    # P x K with one singleton identity
        l_ref, g_ref = ops.batch_hard_triplet(f, yk, margin)
        lr_ = torch.empty(n, device='cuda'); g = torch.empty(n, d, device='cuda')
        wsb, nb = ws(3 * n * n * 4)
        call('fte_batch_hard_triplet_fwd_bwd', dev(f), dev(yk, torch.int32), 0.0 if margin is None else margin, int(margin is None), 1.0,
             lr_, g, n, d, wsb, nb, stream())
        check_maxabs(host(lr_), l_ref, 1e-5, 'triplet loss'); check_rell2(host(g), g_ref, 1e-5, 'triplet grad')


def test_optimizers_and_reductions():
    r = _rng(9)
    n = 100003 * 4
    w = r.standard_normal(n); acc = r.standard_normal(n) * 0.1; g = r.standard_normal(n)
    w_ref, acc_ref = ops.momentum_step(w, acc, 0.5 * g + 5e-4 * w, 0.1)
    wd_, ad = dev(w), dev(acc)
    call('fte_momentum_update', wd_, ad, dev(g), n, 0.1, 0.9, 5e-4, 0.5, stream())
    check_maxabs(host(wd_), w_ref, 1e-6, 'momentum w'); check_maxabs(host(ad), acc_ref, 1e-6, 'momentum acc')
    m = r.standard_normal(n) * 0.1; v = np.abs(r.standard_normal(n)) * 0.01
    w2, m2, v2 = ops.adam_step(w, m, v, g, 0.01, 3)
    wd_, md, vd = dev(w), dev(m), dev(v)
    call('fte_adam_update', wd_, md, vd, dev(g), n, 0.01, 0.5, 0.999, 1e-8, 0.0, 1.0, 3, stream())
    # (1 - beta2) is formed in fp32 like TF's ApplyAdam does: 1 - 0.999f is off by 1.3e-5 relative
    check_maxabs(host(wd_), w2, 2e-5, 'adam w'); check_maxabs(host(md), m2, 1e-6, 'adam m'); check_maxabs(host(vd), v2, 2e-5, 'adam v')
    out = torch.empty(1, device='cuda'); wsb, nb = ws(4096)
    call('fte_sumsq', dev(w), n, 0.25, out, wsb, nb, stream())
    assert abs(float(out) - 0.25 * (w * w).sum()) <= 1e-5 * 0.25 * (w * w).sum()
    call('fte_sum', dev(w[:1001 * 4]), 1001 * 4, 2.0, out, wsb, nb, stream())
    assert abs(float(out) - 2.0 * w[:1001 * 4].sum()) <= 1e-3
    a = r.standard_normal((37, 96)); b = r.standard_normal(24)
    o = torch.empty(24, device='cuda')
    call('fte_reduce_rows', dev(a), o, dev(b), 24, 37, 96, 4, 0.5, stream())
    check_maxabs(host(o), 0.5 * a.reshape(37, 4, 24).sum((0, 1)) + b, 1e-5, 'reduce_rows fold')


@pytest.mark.parametrize('rows,cols', [(2, 4096), (7, 36868), (33, 147456), (129, 4100), (300, 36864), (42, 589824)])
def test_reduce_rows_of_split_k_slabs(rows, cols):
    """column sums of a few long rows: the 16-byte slab reduction (reduce_slabs_kernel<64> for long rows, <16> when the columns alone
    would not fill the chip; ragged last block; row counts that are no multiple of the eight loads in flight) against float64."""
    r = _rng(rows)
    a = r.standard_normal((rows, cols)).astype(np.float32)
    o = torch.empty(cols, device='cuda')
    call('fte_reduce_rows', dev(a), o, None, 1, rows, cols, 1, 1.0, stream())
    ref = a.astype(np.float64).sum(0)
    check_maxabs(host(o), ref, 4e-6 * np.sqrt(rows) * 4, 'reduce_rows slabs %dx%d' % (rows, cols))


def test_bad_arguments_are_rejected_not_run():
    from tf_face_toolbox_amd._lib import FteError
    y = torch.empty(4, device='cuda')
    with pytest.raises(FteError):
        call('fte_conv3x3_fwd', y, y, None, None, None, None, y, 1, 8, 8, 48, 64, 1, None, 0, stream())     # cin % 32
    with pytest.raises(FteError):
        call('fte_conv3x3_wgrad', y, y, y, 64, 28, 28, 64, 64, 1, None, 0, stream())               # split-K needs a workspace


def test_unaligned_pointers_are_rejected_not_executed():
    """16 bytes per lane everywhere: a pointer or pitch that is not 16-byte aligned must come back as an error code."""
    from tf_face_toolbox_amd._lib import FteError
    n, h, w, cin, cout = 2, 8, 8, 64, 64
    x = torch.randn(n * h * w * cin + 4, device='cuda'); wt = torch.randn(9 * cin * cout, device='cuda')
    y = torch.empty(n * h * w * cout + 4, device='cuda')
    buf, nb = ws(query('fte_conv3x3_fwd_ws_bytes', n, h, w, cin, cout, 1))
    call('fte_conv3x3_fwd', x, wt, None, None, None, None, y, n, h, w, cin, cout, 1, buf, nb, stream())       # aligned: fine
    with pytest.raises(FteError):
        call('fte_conv3x3_fwd', x[1:], wt, None, None, None, None, y, n, h, w, cin, cout, 1, buf, nb, stream())
    with pytest.raises(FteError):
        call('fte_conv3x3_fwd', x, wt, None, None, None, None, y[1:], n, h, w, cin, cout, 1, buf, nb, stream())
    torch.cuda.synchronize()


def test_two_gib_tensors_are_rejected():
    """fte.h conventions: every tensor is smaller than 2 GiB (32-bit buffer offsets, 0x80000000 = the out-of-range marker); a larger
    shape is FTE_EINVAL before anything is launched, in the direct and the Winograd plan alike (csrc/api.hip set_bytes / wino_shape_ok).
    9000 x 56 x 56 x 64 floats = 7.2 GB: the pointers are never dereferenced."""
    y = torch.empty(64, device='cuda')
    for algo in (0, 1, 2):
        call('fte_set_conv_algo', algo)
        try:
            assert query('fte_conv3x3_fwd', y.data_ptr(), y.data_ptr(), 0, 0, 0, 0, y.data_ptr(), 9000, 56, 56, 64, 64, 1, 0, 0, 0) == -1
            assert query('fte_conv3x3_wgrad', y.data_ptr(), y.data_ptr(), y.data_ptr(), 9000, 56, 56, 64, 64, 1, y.data_ptr(), 256, 0) == -1
        finally:
            call('fte_set_conv_algo', 2)
    # the eval path's helpers (fte_flip_width, fte_axpby) against numpy
    r = _rng(31)
    x = r.standard_normal((3, 5, 7, 3)).astype(np.float32); x4 = r.standard_normal((2, 4, 6, 8)).astype(np.float32)
    for a in (x, x4):
        o = torch.empty(a.shape, device='cuda')
        call('fte_flip_width', dev(a), o, a.shape[0], a.shape[1], a.shape[2], a.shape[3], stream())
        np.testing.assert_array_equal(o.cpu().numpy(), a[:, :, ::-1, :])
    u, v = r.standard_normal(1000).astype(np.float32), r.standard_normal(1000).astype(np.float32)
    o = torch.empty(1000, device='cuda')
    call('fte_axpby', 0.5, dev(u), 0.5, dev(v), o, 1000, stream())
    np.testing.assert_array_equal(o.cpu().numpy(), np.float32(0.5) * u + np.float32(0.5) * v)
