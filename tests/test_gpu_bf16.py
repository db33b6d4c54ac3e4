"""-m gpu: the bf16-operand MFMA mode (fte_set_mfma_dtype(FTE_MFMA_BF16)).  The kernels round operand tiles to bf16
(RNE) and accumulate in fp32, so against the float64 oracle evaluated on the SAME bf16-rounded operands they must
agree to fp32 accumulation error; against the unrounded oracle the stated mixed-precision tolerance is rel-L2 <= 1e-2
(SURVEY 8c)."""
import numpy as np
import pytest
import torch

from oracle import ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host, stream, ws, check_maxabs, check_rell2, kink_of
    from tf_face_toolbox_amd import _lib


def rb(a):
    """fp32 -> bf16 (round to nearest even) -> float64, as the kernel's v_cvt_pk_bf16_f32 does"""
    return torch.tensor(np.asarray(a, np.float32)).bfloat16().double().numpy()


@pytest.fixture
def bf16_mode():
    _lib.set_mfma_dtype('bf16')
    assert _lib.get_mfma_dtype() == 'bf16'
    yield
    _lib.set_mfma_dtype('f32')


def test_dtype_switch_validates():
    with pytest.raises(_lib.FteError):
        _lib.call('fte_set_mfma_dtype', 7)
    with pytest.raises(ValueError):
        _lib.set_mfma_dtype('fp8')
    assert _lib.get_mfma_dtype() == 'f32'


@pytest.mark.parametrize('n,h,w,cin,cout,k,stride', [(4, 14, 14, 64, 64, 3, 1), (3, 15, 9, 64, 128, 3, 2), (2, 28, 28, 128, 64, 3, 1),
                                                     (5, 8, 8, 256, 128, 1, 1), (2, 12, 12, 32, 64, 3, 1), (64, 14, 14, 128, 128, 3, 2),
                                                     (2, 7, 7, 512, 512, 3, 1), (3, 16, 16, 96, 192, 1, 2)])
def test_conv_fwd_dgrad_wgrad_bf16_operands(bf16_mode, n, h, w, cin, cout, k, stride):
    rng = np.random.default_rng(n * 1000 + cin + cout + k + stride)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32); wt = (rng.standard_normal((k, k, cin, cout)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.1
    alpha = rng.uniform(0.1, 0.4, cout).astype(np.float32)
    z_ref = ops.conv2d_fwd(rb(x), rb(wt), stride, bias.astype(np.float64))
    y_ref = ops.prelu_fwd(z_ref, alpha.astype(np.float64))
    dz = rng.standard_normal(z_ref.shape).astype(np.float32)
    dx_ref, _ = ops.conv2d_bwd(x.astype(np.float64), rb(wt), rb(dz), stride)            # dgrad: operands dz, w
    _, dw_ref = ops.conv2d_bwd(rb(x), wt.astype(np.float64), rb(dz), stride, need_dx=False)   # wgrad: operands x, dz
    xd, wd_, dzd = dev(x), dev(wt), dev(dz)
    zd = torch.empty(z_ref.shape, device='cuda'); yd = torch.empty_like(zd)
    q = _lib.query
    buf, nb = ws(max(q('fte_conv2d_fwd_ws_bytes', n, h, w, cin, cout, k, stride), q('fte_conv2d_dgrad_ws_bytes', n, h, w, cin, cout, k, stride),
                     q('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, k, stride)))
    _lib.call('fte_conv2d_fwd', xd, wd_, dev(bias), dev(alpha), None, zd, yd, n, h, w, cin, cout, k, stride, buf, nb, stream())
    check_maxabs(host(zd), z_ref, 2e-5, 'z (bf16 operands)')
    check_maxabs(host(yd), y_ref, 2e-5, 'y')
    z_exact = ops.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), stride, bias.astype(np.float64))
    check_rell2(host(zd), z_exact, 1e-2, 'z vs the unrounded oracle')               # the stated mixed-precision tolerance
    assert np.sqrt(((host(zd) - z_exact) ** 2).sum()) > 1e-4 * np.sqrt((z_exact ** 2).sum())   # ... and it IS the bf16 path
    if cin % 64 == 0:
        dxd = torch.empty(x.shape, device='cuda')
        _lib.call('fte_conv2d_dgrad', dzd, wd_, None, None, None, None, dxd, None, None, n, h, w, cin, cout, k, stride, buf, nb, stream())
        check_maxabs(host(dxd), dx_ref, 2e-5, 'dgrad')
    dwd = torch.empty(wt.shape, device='cuda')
    _lib.call('fte_conv2d_wgrad', xd, dzd, dwd, n, h, w, cin, cout, k, stride, buf, nb, stream())
    check_rell2(host(dwd), dw_ref, 2e-5, 'wgrad')


@pytest.mark.parametrize('m,n,k', [(64, 128, 512), (37, 10624, 512), (512, 512, 25088), (8, 128, 2048)])
def test_dense_gemms_bf16_operands(bf16_mode, m, n, k):
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((m, k)).astype(np.float32); b = (rng.standard_normal((k, n)) * 0.05).astype(np.float32)
    g = rng.standard_normal((m, n)).astype(np.float32)
    ad, bd, gd = dev(a), dev(b), dev(g)
    buf, nb = ws(_lib.query('fte_gemm_ws_bytes', m, n, k))
    c = torch.empty(m, n, device='cuda')
    _lib.call('fte_gemm_nn', ad, bd, None, c, m, n, k, buf, nb, stream())
    check_rell2(host(c), rb(a) @ rb(b), 2e-5, 'nn')
    da = torch.empty(m, k, device='cuda')
    _lib.call('fte_gemm_nt', gd, bd, None, None, 0, None, da, None, m, n, k, buf, nb, stream())
    check_rell2(host(da), rb(g) @ rb(b).T, 2e-5, 'nt')
    db = torch.empty(k, n, device='cuda')
    _lib.call('fte_gemm_tn', ad, gd, db, m, n, k, buf, nb, stream())
    check_rell2(host(db), rb(a).T @ rb(g), 2e-5, 'tn')


def test_spherenet_step_bf16_vs_oracle_and_training(bf16_mode):
    """Whole SphereNet-ASoftmax step in the bf16-operand mode against the UNROUNDED float64 oracle at the stated
    mixed-precision tolerance (rel-L2 1e-2 on embeddings / logits, 2e-2 on gradients), then a few optimizer steps."""
    from oracle import spherenet as osn
    from tf_face_toolbox_amd import net_select, Singular
    n, h, w, ch, ncls = 8, 64, 64, 3, 40
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    net = net_select('SphereNet', 'NHWC', 5e-4)
    net.build(h, w, ch, ncls, 'cuda')
    p = {k: host(net.get_variable(k)) for k in net.variables}
    out = net.forward(dev(x), num_classes=ncls, is_training=True)
    losses, names, _ = net.loss_function('T', dev(y, torch.int32), **out)
    net.backward()
    torch.cuda.synchronize()
    l_ref, g_ref, cache = osn.loss_and_grads(p, x, y, data_format='NHWC', weight_decay=5e-4, kink=kink_of(net), kink_mode='bf16')
    check_rell2(host(out['logits']), cache['logits'], 1e-2, 'logits (bf16 operands)')
    assert abs(float(losses[0]) - l_ref[0]) <= 1e-2 * l_ref[0]
    worst = 0.0
    for k in net.variables:
        got = host(net.get_variable(k, net.grads)) + (5e-4 * p[k] if k.endswith('/weights') else 0)
        worst = max(worst, check_rell2(got, g_ref[k], 3e-2, 'grad ' + k))
    assert worst > 1e-5                                     # it is the bf16 path, not fp32
    step, ls, names, _ = Singular(net_select('SphereNet-ASoftmax', 'NCHW', 5e-4), 0.01, 'Momentum')(
        {'images': dev(x), 'labels': dev(y, torch.int32), 'num_classes': ncls, 'num_examples': n})
    hist = []
    for _ in range(30):
        step()
        hist.append(float(ls[0]))
    assert np.isfinite(hist).all() and hist[-1] < hist[0]


def _bf16_bits(t):
    """torch float32 -> uint16 bit pattern of its bf16 rounding (RNE), as a torch int16 tensor on the same device"""
    return t.bfloat16().view(torch.int16)


@pytest.mark.parametrize('n,h,w,cin,cout,k,stride', [(4, 14, 14, 64, 64, 3, 1), (3, 15, 9, 64, 128, 3, 2), (2, 28, 28, 128, 64, 3, 1),
                                                     (5, 8, 8, 256, 128, 1, 1), (64, 14, 14, 128, 128, 3, 2), (2, 7, 7, 512, 512, 3, 1),
                                                     (512, 14, 14, 128, 128, 3, 1), (32, 56, 56, 64, 64, 3, 1), (500, 14, 14, 256, 256, 3, 1),
                                                     (130, 28, 28, 128, 128, 1, 1),
                                                     # 128x128-tile launches of 3x3 / stride 1 layers: W = 28 with a ragged last tile, W = 7 (also the shapes of
                                                     # igemm16.hip's optional WINDOW kernel, test_window_kernel_option below)
                                                     (126, 28, 28, 128, 128, 3, 1), (512, 7, 7, 512, 512, 3, 1)])
def test_bf16_source_entry_points_equal_the_operand_mode(bf16_mode, n, h, w, cin, cout, k, stride):
    """fte_conv2d_{fwd,dgrad,wgrad}16 read bf16 COPIES of the operands; the FTE_MFMA_BF16 mode rounds the fp32 operands
    inside the kernel.  Same rounded operands, fp32 accumulation: the results agree to fp32 summation order (the LDS-DMA
    kernel of igemm16.hip walks K in 64-deep steps, the register-staged one in 32-deep steps; where both run the same
    kernel the bits are identical, which 2e-6 still pins).  The bf16 result copies (y16, dzprev16) must be the RNE
    rounding of the fp32 results, exactly."""
    g = torch.Generator(device='cuda').manual_seed(n + cin + cout + k)
    x = torch.randn(n, h, w, cin, device='cuda', generator=g); wt = torch.randn(k, k, cin, cout, device='cuda', generator=g) * 0.1
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1; alpha = torch.rand(cout, device='cuda', generator=g) * 0.3 + 0.1
    alp = torch.rand(cin, device='cuda', generator=g) * 0.3 + 0.1
    res = torch.randn(n, ho, wo, cout, device='cuda', generator=g)
    dz = torch.randn(n, ho, wo, cout, device='cuda', generator=g); zp = torch.randn(n, h, w, cin, device='cuda', generator=g)
    add = torch.randn(n, h, w, cin, device='cuda', generator=g)
    q = _lib.query
    buf, nb = ws(max(q('fte_conv2d_fwd_ws_bytes', n, h, w, cin, cout, k, stride), q('fte_conv2d_dgrad_ws_bytes', n, h, w, cin, cout, k, stride),
                     q('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, k, stride)))
    st = stream()
    # operand mode (fp32 sources, rounded in the kernel)
    z0 = torch.empty_like(res); y0 = torch.empty_like(res)
    _lib.call('fte_conv2d_fwd', x, wt, bias, alpha, res, z0, y0, n, h, w, cin, cout, k, stride, buf, nb, st)
    raw0 = torch.empty_like(x); dx0 = torch.empty_like(x); da0 = torch.empty(cin, device='cuda'); db0 = torch.empty(cin, device='cuda')
    _lib.call('fte_conv2d_dgrad', dz, wt, add, zp, alp, raw0, dx0, da0, db0, n, h, w, cin, cout, k, stride, buf, nb, st)
    dw0 = torch.empty_like(wt)
    _lib.call('fte_conv2d_wgrad', x, dz, dw0, n, h, w, cin, cout, k, stride, buf, nb, st)
    # bf16 copies
    x16 = torch.empty(x.shape, dtype=torch.int16, device='cuda'); dz16 = torch.empty(dz.shape, dtype=torch.int16, device='cuda')
    w16 = torch.empty(wt.shape, dtype=torch.int16, device='cuda'); w16t = torch.empty(k, k, cout, cin, dtype=torch.int16, device='cuda')
    _lib.call('fte_to_bf16', x, x16, x.numel(), st); _lib.call('fte_to_bf16', dz, dz16, dz.numel(), st)
    _lib.call('fte_pack_weights_bf16', wt, w16, w16t, k, cin, cout, st)
    assert torch.equal(x16, _bf16_bits(x)) and torch.equal(w16, _bf16_bits(wt))
    assert torch.equal(w16t, _bf16_bits(wt).permute(0, 1, 3, 2).contiguous())
    z1 = torch.empty_like(res); y1 = torch.empty_like(res); y16 = torch.empty(res.shape, dtype=torch.int16, device='cuda')
    _lib.call('fte_conv2d_fwd16', x16, w16t, bias, alpha, res, z1, y1, y16, n, h, w, cin, cout, k, stride, buf, nb, st)
    _close(z1, z0, 'z'); _close(y1, y0, 'y')
    assert torch.equal(y16, _bf16_bits(y1))
    raw1 = torch.empty_like(x); dx1 = torch.empty_like(x); dx16 = torch.empty(x.shape, dtype=torch.int16, device='cuda')
    da1 = torch.empty(cin, device='cuda'); db1 = torch.empty(cin, device='cuda')
    _lib.call('fte_conv2d_dgrad16', dz16, w16, add, zp, alp, raw1, dx1, dx16, da1, db1, n, h, w, cin, cout, k, stride, buf, nb, st)
    _close(raw1, raw0, 'raw'); _close(dx1, dx0, 'dx'); _close(da1, da0, 'dalpha', 2e-5); _close(db1, db0, 'dbias', 2e-5)
    assert torch.equal(dx16, _bf16_bits(dx1))
    dw1 = torch.empty_like(wt)
    _lib.call('fte_conv2d_wgrad16', x16, dz16, dw1, n, h, w, cin, cout, k, stride, buf, nb, st)
    _close(dw1, dw0, 'dw')        # same bf16 products, fp32 sums; the resident kernel (wgrad16.hip) takes them in another fixed order


@pytest.mark.parametrize('mode', ['8', '1'])
def test_window_kernel_option(mode):
    """igemm16.hip's window kernel (FTE_IGEMM16_WIN: A rows fetched once per 64-channel chunk, the nine taps as row-shifted views with
    per-lane edge masks; 8 = eight waves per block, 1 = four) is an option read once per process: the entry-point test above, which
    holds its shapes (14x14 / 28x28 ragged / 7x7, forward and data gradient), runs again in a child process with the option set."""
    import os, subprocess, sys
    env = dict(os.environ, FTE_IGEMM16_WIN=mode)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_bf16.py'), '-m', 'gpu', '-q', '-x',
                        '-k', 'source_entry_points'], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def _close(a, b, what, tol=2e-6):
    check_rell2(host(a), host(b), tol, what + ': bf16-copy entry point vs operand mode')


def test_spherenet_step_with_bf16_copies_equals_operand_mode(bf16_mode):
    """SphereNet routes its convolutions through the bf16-copy entry points in the bf16 mode (y16 / dz16 written by the
    producing epilogues, weights packed per step).  Same arithmetic -> every gradient equal to the operand mode up to fp32 summation order."""
    from tf_face_toolbox_amd import net_select
    n, h, w, ncls = 6, 64, 64, 30
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).cuda(); y = torch.randint(0, ncls, (n,), generator=g, dtype=torch.int32).cuda()
    outs = []
    for copies in (False, True):
        net = net_select('SphereNet-ASoftmax', 'NCHW', 5e-4)
        net.seed = 9
        net.bf16_copies = copies
        net.build(h, w, 3, ncls, 'cuda')
        o = net.forward(x, y, num_classes=ncls, is_training=True)
        losses, _, _ = net.loss_function('T', y, **o)
        net.backward()
        torch.cuda.synchronize()
        assert (net.y16 is not None) == copies
        outs.append(([float(v) for v in losses], net.grads.clone(), net.emb.clone()))
        e_eval = net.forward(x, is_training=False).clone()          # flip-averaged eval path through the same kernels
        outs[-1] += (e_eval,)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    _close(outs[1][1], outs[0][1], 'gradient arena', 1e-4); _close(outs[1][2], outs[0][2], 'embedding', 2e-5); _close(outs[1][3], outs[0][3], 'eval embedding', 2e-5)


@pytest.mark.parametrize('n,h,w,c,groups,stride', [(3, 14, 14, 128, 32, 1), (2, 9, 7, 256, 32, 1), (2, 8, 8, 512, 32, 1), (1, 5, 5, 1024, 32, 1),
                                                   (2, 6, 6, 64, 8, 1), (2, 14, 14, 256, 32, 2), (3, 7, 7, 1024, 32, 2), (2, 9, 6, 512, 32, 2),
                                                   (1, 8, 11, 128, 32, 2)])
def test_grouped_3x3_on_the_bf16_mfma(n, h, w, c, groups, stride):
    """fte_gconv3x3_pack_bf16 + fte_gconv3x3_bf16 (block-diagonal 32-channel slices on v_mfma_f32_32x32x16_bf16; stride 1 and the
    TF-SAME stride 2 of even and odd sizes) against the grouped convolution with bf16-rounded operands (oracle/ops.py
    operand_rounding): forward, data gradient and filter gradient (fragments through ds_read_b64_tr_b16), 2e-5."""
    r = np.random.default_rng(c + groups + stride)
    gw = c // groups
    ho, wo = -(-h // stride), -(-w // stride)
    x = r.standard_normal((n, h, w, c)); wt = r.standard_normal((groups, 3, 3, gw, gw)) * 0.2
    dz = r.standard_normal((n, ho, wo, c))
    with ops.operand_rounding('bf16'):
        y_ref = np.concatenate([ops.conv2d_fwd(x[..., g * gw:(g + 1) * gw], wt[g], stride) for g in range(groups)], axis=-1)
        bwd = [ops.conv2d_bwd(x[..., g * gw:(g + 1) * gw], wt[g], dz[..., g * gw:(g + 1) * gw], stride) for g in range(groups)]
    dx_ref = np.concatenate([b[0] for b in bwd], axis=-1)
    dw_ref = np.stack([b[1] for b in bwd])
    words = (c // 32) * 9 * 1024
    wf = torch.empty(words, dtype=torch.int16, device='cuda'); wd = torch.empty(words, dtype=torch.int16, device='cuda')
    _lib.call('fte_gconv3x3_pack_bf16', dev(wt.reshape(groups, 9, gw, gw)), wf, wd, c, groups, stream())
    y = torch.full((n, ho, wo, c), 7.0, device='cuda'); dx = torch.full((n, h, w, c), 7.0, device='cuda')
    _lib.call('fte_gconv3x3_bf16', dev(x), wf, y, n, h, w, c, stride, 0, stream())
    _lib.call('fte_gconv3x3_bf16', dev(dz), wd, dx, n, h, w, c, stride, 1, stream())
    check_maxabs(host(y), y_ref, 2e-5, 'grouped fwd, bf16 operands')
    check_maxabs(host(dx), dx_ref, 2e-5, 'grouped dgrad, bf16 operands')
    buf, nb = ws(_lib.query('fte_gconv3x3_wgrad_bf16_ws_bytes', n, h, w, c, groups, stride))
    dw = torch.full((groups, 3, 3, gw, gw), 7.0, device='cuda')
    _lib.call('fte_gconv3x3_wgrad_bf16', dev(x), dev(dz), dw, n, h, w, c, groups, stride, buf, nb, stream())
    check_maxabs(host(dw), dw_ref, 2e-5, 'grouped wgrad, bf16 operands')
    # and it really is the rounded product: the unrounded one is far outside that tolerance
    y_plain = np.concatenate([ops.conv2d_fwd(x[..., g * gw:(g + 1) * gw], wt[g], stride) for g in range(groups)], axis=-1)
    assert np.abs(host(y) - y_plain).max() > 1e-4 * np.abs(y_plain).max()
