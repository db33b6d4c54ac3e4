"""-m gpu: BASELINE.json configs[2] (ResNeXt-50 + center loss, 128 x 112 x 112 per GPU, bf16) and configs[3] (SENet-50 +
batch-hard triplet with online mining, 128 = P x K = 32 x 4 per GPU) AT THEIR WORKLOAD.

  * the 50-layer nets at 112 x 112 against the float64 graph oracle at a small batch: features, losses, EVERY gradient
    (the oracle needs seconds per image at this depth; reference: nets/resnext.py:34-67 as intended, nets/resnet.py:63-92
    + the SE gate of nets/shufflenet_v2.py:79-85, loss.py:29-45, loss.py:47-78);
  * the full per-GPU size through size-independent properties: bit-identical repeats, exact linearity of the whole
    backward pass in the upstream gradient (tower_scale 1 vs 1/2: every gradient halves BIT FOR BIT -- the 1/num_gpus
    pre-scale of data_parallel.py:37 costs nothing in accuracy), the loss heads re-derived in numpy float64 FROM THE HIP
    PATH'S OWN FEATURES (batch-hard indices / losses / feature gradients of loss.py:47-78 at P x K = 32 x 4; center loss,
    its gradient and the scatter_sub update of loss.py:37-41 incl. duplicate labels; softmax-CE on the HIP logits), and
    inference-mode batch independence chained to the oracle (images 0..1 alone == inside the 128-batch == oracle);
  * the ResNeXt step in the bf16 MFMA mode (configs[2]'s precision) at the stated mixed-precision tolerance."""
import numpy as np
import pytest
import torch

from oracle import graphnet as og, ops as oops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host, check_maxabs, check_rell2
    from tf_face_toolbox_amd import net_select, _lib
    from test_gpu_resnet import _run_variant, _kink

H = W = 112


# ------------------------------------------------------------------------------------------------ (a) vs the oracle
def test_resnext50_center_at_112_every_gradient():
    n, ncls = 4, 12
    graph, spec = og.resnet_train_graph(50, 3, ncls, 'resnext')
    rng = np.random.default_rng(31)
    labels = rng.integers(0, ncls, n); labels[3] = labels[0]                     # duplicate label: scatter_sub accumulates
    cen = rng.standard_normal((ncls, 2048)) * 0.1
    net = net_select('ResNeXt-50-center', 'NCHW', 5e-4)
    assert net.head == 'softmax+center' and net.num_block == [3, 4, 6, 3] and net.num_card == 32
    ref, names = _run_variant(net, graph, spec, n, H, W, ncls, labels, 131,
                              center=dict(centers=cen, alpha=net.center_alpha, weight=net.center_weight))
    assert names == ['cross_entropy', 'center_loss', 'reg_loss']
    check_maxabs(host(net.state['centers']), ref[4]['centers'], 1e-5, 'centers after the update')


def test_senet50_triplet_at_112_every_gradient():
    n, ncls = 6, 9
    graph, spec = og.resnet_train_graph(50, 3, ncls, 'senet', classifier=False)
    labels = np.repeat(np.arange(3), 2)                                          # P x K = 3 x 2
    net = net_select('SENet-50-triplet', 'NCHW', 5e-4)
    assert net.head == 'triplet' and net.num_block == [3, 4, 6, 3]
    ref, names = _run_variant(net, graph, spec, n, H, W, ncls, labels, 141, triplet=None)
    assert names == ['triplet_loss', 'reg_loss']


# ------------------------------------------------------------------------------------------------ (b) full per-GPU size
def _train_pass(net, x, y, ncls, scale):
    net.tower_scale = scale
    net.global_step = 0
    net.dropout_seed = 9
    out = net.forward(x, num_classes=ncls, is_training=True)
    losses, names, _ = net.loss_function('T', y, **out)
    net.backward()
    torch.cuda.synchronize()
    return [float(v) for v in losses], names, net.grads[:net.arena_size].clone()


def _eval_features(net, x):
    net.forward(x, num_classes=net.num_classes, is_training=False)
    torch.cuda.synchronize()
    return net.t['features'].clone()


def test_senet50_triplet_full_shard_32x4():
    """configs[3]: 128 images per GPU = 32 identities x 4 images, mining within the shard (loss.py:47-78)."""
    P, K, ncls = 32, 4, 1000
    n = P * K
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, H, W, 3, generator=g) * 2 - 1).cuda()
    ids = torch.randperm(ncls, generator=g)[:P]
    y = ids.repeat_interleave(K).to(torch.int32).cuda()                          # the P x K sampler keeps identities contiguous (data.py:230-242)
    net = net_select('SENet-50-triplet', 'NCHW', 5e-4)
    net.seed = 4
    net.build(H, W, 3, ncls, 'cuda')
    l1, names, g1 = _train_pass(net, x, y, ncls, 1.0)
    feat = host(net.t['features'])
    dfeat = host(net.dfeat)
    rows = host(net.loss_rows[:n])
    st0 = {k: v.clone() for k, v in net.state.items()}
    l2, _, g2 = _train_pass(net, x, y, ncls, 1.0)
    assert names == ['triplet_loss', 'reg_loss'] and l1 == l2 and torch.equal(g1, g2)          # determinism
    # moving statistics moved twice by the same batch statistics: decay^2 consistent (UPDATE_OPS ran once per pass)
    k0 = 'SENet-50/conv1/conv_7x7/BatchNorm/moving_mean'
    m1, m2 = host(st0[k0]), host(net.state[k0])
    np.testing.assert_allclose(m2, m1 * (1 + 0.999), rtol=2e-4, atol=1e-7)
    lh, _, gh = _train_pass(net, x, y, ncls, 0.5)
    assert torch.equal(gh, g1 * 0.5), 'the backward pass is exactly linear in the upstream gradient'
    assert abs(lh[0] - 0.5 * l1[0]) <= 1e-6 * abs(l1[0])
    # the head, re-derived in float64 from the HIP path's own features
    per, df = oops.batch_hard_triplet(feat, host(y).astype(np.int64), None)
    check_maxabs(rows, per, 2e-5, 'per-sample batch-hard losses (softplus form)')
    check_rell2(dfeat, df / n, 2e-5, 'd(mean loss)/d(features)')
    assert abs(l1[0] - per.mean()) <= 1e-5 * per.mean()
    # mining really is hard: every anchor has K-1 positives and 124 negatives, and the mined pairs are not degenerate
    d = np.sqrt(((feat[:, None, :] - feat[None, :, :]) ** 2).sum(-1) + 1e-12)
    same = host(y)[:, None] == host(y)[None, :]
    assert (np.where(same, d, 0).max(1) > 0).all() and (np.where(same, 1e6, d).min(1) < 1e6).all()
    # inference mode is batch-independent: chained to the oracle on two images
    e_full = _eval_features(net, x)
    e_two = _eval_features(net, x[:2])
    check_maxabs(host(e_two), host(e_full[:2]), 2e-5, 'eval features: alone vs inside the 128-batch')
    graph, spec = og.resnet_train_graph(50, 3, ncls, 'senet', classifier=False)
    p = {k: host(net.get_variable(k)) for k in net.variables}
    state = {k: host(net.get_variable(k)) for k in net.state}
    env, _, _ = og.forward(graph, p, host(x[:2]), train=False, state=state)
    check_maxabs(host(e_two), env['features'], 5e-5, 'eval features vs the float64 oracle')


def test_resnext50_center_full_shard_128():
    """configs[2]: 128 images per GPU, softmax + 0.008 x center loss on the 2048-d pooled features."""
    n, ncls = 128, 10575
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, H, W, 3, generator=g) * 2 - 1).cuda()
    y = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).cuda() * 7           # 40 classes among 128 rows: duplicates
    net = net_select('ResNeXt-50-center', 'NCHW', 5e-4)
    net.seed = 6
    net.build(H, W, 3, ncls, 'cuda')
    cen0 = (torch.randn(ncls, 2048, generator=g) * 0.05).cuda()
    net._centers().copy_(cen0)
    l1, names, g1 = _train_pass(net, x, y, ncls, 1.0)
    feat, dfe, logits = host(net.t['features']), host(net.dfeat), host(net.t['logits'])[:, :ncls]
    cen1 = host(net._centers())
    assert names == ['cross_entropy', 'center_loss', 'reg_loss']
    yl = host(y).astype(np.int64)
    cl, dfeat_ref, newc = oops.center_loss(feat, yl, host(cen0), net.center_alpha)
    assert abs(l1[1] - cl) <= 1e-5 * cl
    check_rell2(dfe, dfeat_ref * net.center_weight, 2e-5, 'center-loss gradient wrt the features')
    check_maxabs(cen1, newc, 1e-5, 'centers after scatter_sub (duplicates accumulate)')
    ce, _ = oops.softmax_ce(logits, yl)
    assert abs(l1[0] - ce) <= 1e-5 * ce
    net._centers().copy_(cen0)
    l2, _, g2 = _train_pass(net, x, y, ncls, 1.0)
    assert l1 == l2 and torch.equal(g1, g2)                                             # determinism (same dropout seed/step)
    net._centers().copy_(cen0)
    lh, _, gh = _train_pass(net, x, y, ncls, 0.5)
    assert torch.equal(gh, g1 * 0.5), 'the backward pass is exactly linear in the upstream gradient'
    e_full = _eval_features(net, x)
    e_two = _eval_features(net, x[:2])
    check_maxabs(host(e_two), host(e_full[:2]), 2e-5, 'eval features: alone vs inside the 128-batch')
    graph, spec = og.resnet_train_graph(50, 3, ncls, 'resnext')
    p = {k: host(net.get_variable(k)) for k in net.variables}
    state = {k: host(net.get_variable(k)) for k in net.state if k != 'centers'}
    env, _, _ = og.forward(graph, p, host(x[:2]), train=False, state=state)
    check_maxabs(host(e_two), env['features'], 5e-5, 'eval features vs the float64 oracle')


# ------------------------------------------------------------------------------------------------ (c) bf16 mode
def test_resnext50_center_step_in_bf16_mode():
    """configs[2]'s precision: bf16 MFMA operands, fp32 accumulate / storage, on ResNeXt-50 + center at 112 x 112.

    A randomly initialised 50-layer BN net amplifies ANY perturbation of its early layers (the float64 oracle with
    bf16-rounded operands moves the pooled features of this input by 13 % against the unrounded oracle, and its own float32
    evaluation by 5 %), so an end-to-end gradient comparison says nothing about the kernels.  What is checked instead:
      1. layer by layer, teacher-forced: every convolution / dense product of the net, fed the HIP path's OWN input
         tensor, equals the float64 oracle on the same bf16-rounded operands to 2e-5 (fp32 accumulation error only) --
         51 convolutions incl. the 7x7 stem and all 1x1s, the classifier, and the 16 grouped 3x3s (13 of stride 1, three of
         stride 2: on the bf16 MFMA in this mode);
      2. end to end the HIP path deviates from the unrounded oracle no more than bf16-operand arithmetic must on this
         input: features rel-L2 <= 1.5 x the deviation of the oracle's own bf16-operand evaluation, losses likewise;
      3. every gradient is finite and optimizer steps run in this mode.
    The gradient kernels themselves are pinned per entry point against the rounded-operand oracle in test_gpu_bf16.py."""
    from tf_face_toolbox_amd import Singular
    n, ncls = 4, 12
    graph, spec = og.resnet_train_graph(50, 3, ncls, 'resnext')
    p, state = og.init_params(spec, 151)
    p = og.perturb(p, 152)
    rng = np.random.default_rng(153)
    x = rng.uniform(-1, 1, (n, H, W, 3)); labels = rng.integers(0, ncls, n)
    cen = rng.standard_normal((ncls, 2048)) * 0.1
    net = net_select('ResNeXt-50-center', 'NCHW', 5e-4)
    net.build(H, W, 3, ncls, 'cuda')
    net.load_params(p)
    net._centers().copy_(torch.tensor(cen, dtype=torch.float32))
    _lib.set_mfma_dtype('bf16')
    try:
        out = net.forward(dev(x), num_classes=ncls, is_training=True)
        losses, names, _ = net.loss_function('T', dev(labels, torch.int32), **out)
        net.backward()
        torch.cuda.synchronize()
        got_losses = [float(v) for v in losses]
        grads_finite = bool(torch.isfinite(net.grads).all())
        # ---- 1. teacher-forced, layer by layer ----
        worst, checked, grouped = 0.0, 0, 0
        with oops.operand_rounding('bf16'):
            for op in graph:
                if op[0] == 'conv':
                    xin = x if op[2] == 'images' else host(net.t[op[2]])
                    ref = oops.conv2d_fwd(xin, p[op[3]], op[4])
                    worst = max(worst, check_maxabs(host(net.t[op[1]])[..., :ref.shape[-1]], ref, 2e-5, 'conv ' + op[1]))
                    checked += 1
                elif op[0] == 'fc':
                    ref = oops.fc_fwd(host(net.t[op[2]]), p[op[3]])
                    worst = max(worst, check_maxabs(host(net.t[op[1]])[:, :ncls], ref, 2e-5, 'fc ' + op[1]))
                    checked += 1
                elif op[0] == 'gconv':                         # grouped 3x3: on the bf16 MFMA in this mode
                    xin = host(net.t[op[2]])
                    gw = xin.shape[-1] // op[5]
                    ref = np.concatenate([oops.conv2d_fwd(xin[..., g * gw:(g + 1) * gw], p[op[3]][g], op[4]) for g in range(op[5])], axis=-1)
                    worst = max(worst, check_maxabs(host(net.t[op[1]]), ref, 2e-5, 'gconv ' + op[1]))
                    grouped += 1
        assert checked == 1 + 16 * 2 + 4 + 1, checked          # stem + (conv1, conv3) x 16 blocks + 4 projection shortcuts + classifier
        assert grouped == 16, grouped                            # 16 grouped 3x3 layers, three of them stride 2
        # ---- 2. end to end against the unrounded oracle ----
        masks = {'features_drop': host(net.t['features_drop/mask'])}
        center = dict(centers=cen, alpha=net.center_alpha, weight=net.center_weight)
        plain = og.loss_and_grads(graph, p, x, labels, 5e-4, masks=masks, state=state, center=center)
        with oops.operand_rounding('bf16'):
            rounded = og.loss_and_grads(graph, p, x, labels, 5e-4, masks=masks, state=state, center=center)

        def rel(a, b):
            return float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
        dev_hip = rel(host(net.t['features']), plain[2]['features'])
        dev_orc = rel(rounded[2]['features'], plain[2]['features'])
        assert 1e-4 < dev_hip <= 1.5 * dev_orc, (dev_hip, dev_orc)
        for a, b, c in zip(got_losses, plain[0], rounded[0]):
            assert abs(a - b) <= 3 * abs(c - b) + 2e-2 * abs(b), (names, a, b, c)      # a scalar's deviation can be small by chance
        assert grads_finite
        print('bf16 ResNeXt-50 + center: %d products teacher-forced, worst max-abs %.1e; features vs fp32 oracle %.3f (oracle with bf16 operands: %.3f)'
              % (checked, worst, dev_hip, dev_orc))
        # ---- 3. optimizer steps run in this mode (convergence in the bf16 mode is tested on SphereNet, test_gpu_convergence.py;
        # a 4-image batch through 50 BN layers is too noisy to assert a falling loss on) ----
        net2 = net_select('ResNeXt-50-center', 'NCHW', 5e-4)
        step, ls, _, _ = Singular(net2, 1e-3, 'Momentum')({'images': dev(x), 'labels': dev(labels, torch.int32), 'num_classes': ncls, 'num_examples': n})
        w0 = net2.params.clone()
        for _ in range(3):
            step()
        assert all(np.isfinite(float(v)) for v in ls) and bool(torch.isfinite(net2.params).all()) and not torch.equal(w0, net2.params)
    finally:
        _lib.set_mfma_dtype('f32')
