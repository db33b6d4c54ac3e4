"""-m gpu: the Winograd F(2x2,3x3) / F(3x3,2x2) path of the stride-1 3x3 layers (csrc/wino.hip) through the C ABI against the float64
oracle's DIRECT convolution on the same seeded inputs -- the algorithm the reference switches on for every run (train.py:260,
TF_ENABLE_WINOGRAD_NONFUSED=1) for the layers of nets/sphere.py:38-45,61-70.  Tolerances are the direct path's (tests/util_gpu.py):
forward / data gradient max-abs <= 2e-5 max|ref|, dalpha / dbias rel-L2 <= 2e-5; the transforms hold 0, +-1, +-1/2 only."""
import numpy as np
import pytest
import torch

from oracle import ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import call, query, dev, host, stream, ws, check_maxabs, check_rell2
    from tf_face_toolbox_amd import _lib

DIRECT, WINOGRAD, AUTO = 0, 1, 2

CASES = [
    # n, h, w, cin, cout
    (2, 14, 14, 256, 256),       # stage 3: 7x7 tiles per image, ragged last row block
    (3, 9, 7, 64, 128),          # odd, non-square: half-outside tiles on both edges, cin != cout
    (5, 7, 7, 512, 512),         # stage 4: 7x7 padded to 4x4 tiles
    (1, 28, 28, 128, 128),       # stage 2
    (2, 56, 56, 64, 64),         # stage 1: one column block
    (40, 14, 14, 256, 256),      # 31 row blocks: every XCD class, several tiles per filter-gradient share
    (64, 7, 7, 512, 512),        # the 8-GPU shard of stage 4: 128 tiles = 256 half tiles, no whole round
    (84, 14, 14, 256, 256),      # 260 tiles: a whole round + a second round of 4 tiles (resident blocks with 2 and with 1 tile)
]


@pytest.fixture
def winograd():
    prev = query('fte_get_conv_algo')
    call('fte_set_conv_algo', WINOGRAD)
    yield
    call('fte_set_conv_algo', prev)


def _symbols(fn):
    """kernel symbols of the MFMA launches `fn` makes"""
    call('fte_prof_enable', 1)
    fn()
    torch.cuda.synchronize()
    call('fte_prof_enable', 0)
    return [r[5] for r in _lib.prof_records(shapes=True)]


@pytest.mark.parametrize('n,h,w,cin,cout', CASES)
def test_wino_fwd(winograd, n, h, w, cin, cout):
    r = np.random.default_rng(21)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((3, 3, cin, cout)) * 0.05
    b = r.standard_normal(cout); al = 0.25 + 0.1 * r.standard_normal(cout)
    z_ref = ops.conv2d_fwd(x, wt, 1, b)
    res = r.standard_normal(z_ref.shape)
    y_ref = ops.prelu_fwd(z_ref, al) + res
    z = torch.full(z_ref.shape, 7.0, device='cuda'); y = torch.full(z_ref.shape, 7.0, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_fwd_ws_bytes', n, h, w, cin, cout, 1))
    args = (dev(x), dev(wt), dev(b), dev(al), dev(res), z, y, n, h, w, cin, cout, 1, wsb, nb, stream())
    syms = _symbols(lambda: call('fte_conv3x3_fwd', *args))
    assert syms and all(s_.startswith('wino_mm_kernel<0,') for s_ in syms), syms
    check_maxabs(host(z), z_ref, what='z'); check_maxabs(host(y), y_ref, what='y')
    y2 = torch.full(z_ref.shape, 7.0, device='cuda')
    call('fte_conv3x3_fwd', dev(x), dev(wt), None, None, None, None, y2, n, h, w, cin, cout, 1, wsb, nb, stream())
    check_maxabs(host(y2), ops.conv2d_fwd(x, wt, 1), what='plain')
    # bit-identical run to run (fixed summation orders, no atomics)
    y3 = torch.empty_like(y2)
    call('fte_conv3x3_fwd', dev(x), dev(wt), None, None, None, None, y3, n, h, w, cin, cout, 1, wsb, nb, stream())
    assert torch.equal(y2, y3)


@pytest.mark.parametrize('n,h,w,cin,cout', CASES)
def test_wino_dgrad_with_prelu_backward(winograd, n, h, w, cin, cout):
    r = np.random.default_rng(22)
    x = r.standard_normal((n, h, w, cin)); wt = r.standard_normal((3, 3, cin, cout)) * 0.05
    dz = r.standard_normal((n, h, w, cout))
    dx_ref, _ = ops.conv2d_bwd(x, wt, dz, 1)
    addin = r.standard_normal(x.shape); zprev = r.standard_normal(x.shape); alp = 0.25 + 0.1 * r.standard_normal(cin)
    zprev[0, 0, 0, :4] = 0.0
    g_ref = dx_ref + addin
    dzprev_ref, dalpha_ref = ops.prelu_bwd(zprev, alp, g_ref)
    dbias_ref = dzprev_ref.sum(axis=(0, 1, 2))
    raw = torch.full(x.shape, 7.0, device='cuda'); dzp = torch.full(x.shape, 7.0, device='cuda')
    da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_dgrad_ws_bytes', n, h, w, cin, cout, 1))
    args = (dev(dz), dev(wt), dev(addin), dev(zprev), dev(alp), raw, dzp, da, db, n, h, w, cin, cout, 1, wsb, nb, stream())
    syms = _symbols(lambda: call('fte_conv3x3_dgrad', *args))
    assert syms and all(s_.startswith('wino_mm_kernel<1,') for s_ in syms), syms
    check_maxabs(host(raw), g_ref, what='raw'); check_maxabs(host(dzp), dzprev_ref, what='dzprev')
    check_rell2(host(da), dalpha_ref, what='dalpha'); check_rell2(host(db), dbias_ref, what='dbias')
    dzp2 = torch.full(x.shape, 7.0, device='cuda')
    call('fte_conv3x3_dgrad', dev(dz), dev(wt), None, None, None, None, dzp2, None, None, n, h, w, cin, cout, 1, wsb, nb, stream())
    check_maxabs(host(dzp2), dx_ref, what='plain dgrad')


@pytest.mark.parametrize('n,h,w,cin,cout', CASES)
def test_wino_wgrad(winograd, n, h, w, cin, cout):
    r = np.random.default_rng(23)
    x = r.standard_normal((n, h, w, cin)); wt = np.zeros((3, 3, cin, cout))
    dz = r.standard_normal((n, h, w, cout))
    _, dw_ref = ops.conv2d_bwd(x, wt, dz, 1, need_dx=False)
    dw = torch.full((3, 3, cin, cout), 7.0, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_wgrad_ws_bytes', n, h, w, cin, cout, 1))
    args = (dev(x), dev(dz), dw, n, h, w, cin, cout, 1, wsb, nb, stream())
    syms = _symbols(lambda: call('fte_conv3x3_wgrad', *args))
    assert syms == ['wino_wgrad_kernel'], syms
    check_maxabs(host(dw), dw_ref, what='dw')
    dw2 = torch.empty_like(dw)
    call('fte_conv3x3_wgrad', dev(x), dev(dz), dw2, n, h, w, cin, cout, 1, wsb, nb, stream())
    assert torch.equal(dw, dw2)


def test_switch_off_runs_the_direct_kernels():
    """FTE_CONV_DIRECT: no Winograd launch; the workspace query shrinks back; too small a workspace under FTE_CONV_WINOGRAD falls
    back to the direct algorithm instead of failing; stride 2 and the bf16 operand mode never take the path."""
    n, h, w, c = 4, 14, 14, 256
    r = np.random.default_rng(24)
    x = dev(r.standard_normal((n, h, w, c))); wt = dev(r.standard_normal((3, 3, c, c)) * 0.05)
    y = torch.empty(n, h, w, c, device='cuda')
    prev = query('fte_get_conv_algo')
    try:
        call('fte_set_conv_algo', DIRECT)
        small = query('fte_conv3x3_fwd_ws_bytes', n, h, w, c, c, 1)
        wsb, nb = ws(small)
        syms = _symbols(lambda: call('fte_conv3x3_fwd', x, wt, None, None, None, None, y, n, h, w, c, c, 1, wsb, nb, stream()))
        assert syms and all(s.startswith('igemm') for s in syms), syms
        y_direct = y.clone()
        call('fte_set_conv_algo', WINOGRAD)
        big = query('fte_conv3x3_fwd_ws_bytes', n, h, w, c, c, 1)
        assert big > small and big >= 16 * 4 * ((n * 49 + 63) // 64 * 64) * c
        syms = _symbols(lambda: call('fte_conv3x3_fwd', x, wt, None, None, None, None, y, n, h, w, c, c, 1, wsb, nb, stream()))
        assert all(s.startswith('igemm') for s in syms), syms          # the small workspace: direct
        assert torch.equal(y, y_direct)
        wsb2, nb2 = ws(big)
        syms = _symbols(lambda: call('fte_conv3x3_fwd', x, wt, None, None, None, None, y, n, h, w, c, c, 1, wsb2, nb2, stream()))
        assert syms and all(s_.startswith('wino_mm_kernel<0,') for s_ in syms), syms
        check_maxabs(host(y), host(y_direct), tol=4e-5, what='winograd vs direct')
        y2 = torch.empty(n, 7, 7, c, device='cuda')
        syms = _symbols(lambda: call('fte_conv3x3_fwd', x, wt, None, None, None, None, y2, n, h, w, c, c, 2, wsb2, nb2, stream()))
        assert all(s.startswith('igemm') for s in syms), syms
        call('fte_set_mfma_dtype', 1)
        try:
            syms = _symbols(lambda: call('fte_conv3x3_fwd', x, wt, None, None, None, None, y, n, h, w, c, c, 1, wsb2, nb2, stream()))
            assert all(s.startswith('igemm') for s in syms), syms
        finally:
            call('fte_set_mfma_dtype', 0)
        assert query('fte_set_conv_algo', 3) != 0
    finally:
        call('fte_set_conv_algo', prev)


def test_kept_v_pack_feeds_the_filter_gradient(winograd):
    """fte_conv3x3_fwd_keep leaves V = B^T d B of x; fte_conv3x3_wgrad_kept reads it: bit-identical to the self-contained calls; a kept
    pack on a layer the switch does not select is an error, never a silent fallback."""
    n, h, w, cin, cout = 6, 14, 14, 128, 256
    r = np.random.default_rng(25)
    x = dev(r.standard_normal((n, h, w, cin))); wt = dev(r.standard_normal((3, 3, cin, cout)) * 0.05)
    dz = dev(r.standard_normal((n, h, w, cout)))
    assert query('fte_conv3x3_algo', n, h, w, cin, cout, 1, 0) == WINOGRAD and query('fte_conv3x3_algo', n, h, w, cin, cout, 1, 2) == WINOGRAD
    vb = query('fte_wino_pack_bytes', n, h, w, cin)
    assert vb == 16 * 4 * ((n * 49 + 63) // 64 * 64) * cin
    vpack = torch.empty(vb // 4, device='cuda')
    wsb, nb = ws(max(query('fte_conv3x3_fwd_ws_bytes', n, h, w, cin, cout, 1), query('fte_conv3x3_wgrad_ws_bytes', n, h, w, cin, cout, 1)))
    y1 = torch.empty(n, h, w, cout, device='cuda'); y2 = torch.empty_like(y1)
    call('fte_conv3x3_fwd', x, wt, None, None, None, None, y1, n, h, w, cin, cout, 1, wsb, nb, stream())
    call('fte_conv3x3_fwd_keep', x, wt, None, None, None, None, y2, n, h, w, cin, cout, 1, vpack, wsb, nb, stream())
    assert torch.equal(y1, y2)
    dw1 = torch.empty(3, 3, cin, cout, device='cuda'); dw2 = torch.empty_like(dw1)
    call('fte_conv3x3_wgrad', x, dz, dw1, n, h, w, cin, cout, 1, wsb, nb, stream())
    call('fte_conv3x3_wgrad_kept', x, dz, dw2, n, h, w, cin, cout, 1, vpack, wsb, nb, stream())
    assert torch.equal(dw1, dw2)
    call('fte_set_conv_algo', DIRECT)
    assert query('fte_conv3x3_algo', n, h, w, cin, cout, 1, 0) == DIRECT
    assert query('fte_conv3x3_fwd_keep', x.data_ptr(), wt.data_ptr(), 0, 0, 0, 0, y2.data_ptr(), n, h, w, cin, cout, 1, vpack.data_ptr(),
                 wsb.data_ptr(), nb, 0) == -2
    assert query('fte_conv3x3_wgrad_kept', x.data_ptr(), dz.data_ptr(), dw2.data_ptr(), n, h, w, cin, cout, 1, vpack.data_ptr(),
                 wsb.data_ptr(), nb, 0) == -2


@pytest.mark.parametrize('n,h,w,c', [(64, 7, 7, 512), (2, 14, 14, 256), (4, 8, 8, 128), (2, 56, 56, 64), (40, 14, 14, 256)])
def test_products_are_stable_over_many_launches(winograd, n, h, w, c):
    """300 launches of the forward and the data-gradient product on the same operands while another stream keeps the memory system busy
    (as the filter gradients do in the backward walk): every result equal to the first, bit for bit (half-tile kernel: all cases but
    the 40-image one).  A late zero-fill of an out-of-range LDS-DMA slot landing in the epilogue's exchange buffer showed as a wrong
    output of the half-tile kernel once in a few training steps."""
    r = np.random.default_rng(27)
    x = dev(r.standard_normal((n, h, w, c))); wt = dev(r.standard_normal((3, 3, c, c)) * 0.05)
    b = dev(r.standard_normal(c)); al = dev(r.uniform(0.1, 0.4, c)); res = dev(r.standard_normal((n, h, w, c)))
    wsb, nb = ws(max(query('fte_conv3x3_fwd_ws_bytes', n, h, w, c, c, 1), query('fte_conv3x3_dgrad_ws_bytes', n, h, w, c, c, 1)))
    z0 = torch.empty(n, h, w, c, device='cuda'); y0 = torch.empty_like(z0); z = torch.empty_like(z0); y = torch.empty_like(z0)
    d0 = torch.empty_like(z0); d = torch.empty_like(z0); da0 = torch.empty(c, device='cuda'); da = torch.empty_like(da0)
    call('fte_conv3x3_fwd', x, wt, b, al, res, z0, y0, n, h, w, c, c, 1, wsb, nb, stream())
    call('fte_conv3x3_dgrad', x, wt, None, res, al, None, d0, da0, None, n, h, w, c, c, 1, wsb, nb, stream())
    torch.cuda.synchronize()
    noise_a = torch.empty(64 << 20, device='cuda'); noise_b = torch.empty_like(noise_a)
    side = torch.cuda.Stream()
    bad = 0
    for it in range(300):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a)
        call('fte_conv3x3_fwd', x, wt, b, al, res, z, y, n, h, w, c, c, 1, wsb, nb, stream())
        call('fte_conv3x3_dgrad', x, wt, None, res, al, None, d, da, None, n, h, w, c, c, 1, wsb, nb, stream())
        bad += int(not (torch.equal(z, z0) and torch.equal(y, y0) and torch.equal(d, d0) and torch.equal(da, da0)))
    torch.cuda.synchronize()
    assert bad == 0, '%d of 300 launches differ from the first' % bad
