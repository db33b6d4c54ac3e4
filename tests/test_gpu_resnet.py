"""-m gpu: ResNet (conv-BN-ReLU bottlenecks, max-pool, GAP, dropout) through the reference-shaped API and the
C ABI against the float64 graph oracle."""
import numpy as np
import pytest
import torch

from oracle import graphnet as og

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host, check_maxabs, check_rell2
    from tf_face_toolbox_amd import net_select, Singular
    from tf_face_toolbox_amd.nets.resnet import ResNet


def _kink(net):
    k = {}
    for op in net.graph:
        if op[0] == 'relu':
            k[op[1]] = host(net.t[op[1]])
        elif op[0] == 'maxpool':
            k[op[1] + '/idx'] = net.t[op[1] + '/idx'].cpu().numpy()
            k[op[1]] = host(net.t[op[1]])
    return k


@pytest.mark.parametrize('num_layers,n,h,w,ncls', [(26, 8, 64, 64, 10), (50, 6, 64, 48, 300), (26, 4, 112, 112, 10)])
def test_resnet_forward_loss_and_every_gradient(num_layers, n, h, w, ncls):
    graph, spec = og.resnet_train_graph(num_layers, 3, ncls)
    p, state = og.init_params(spec, 71)
    p = og.perturb(p, 72)
    rng = np.random.default_rng(73)
    x = rng.uniform(-1, 1, (n, h, w, 3)); y = rng.integers(0, ncls, n)
    net = ResNet(num_layers, data_format='NCHW', weight_decay=5e-4)
    net.build(h, w, 3, ncls, 'cuda')
    assert [op for op in net.graph] == graph and sorted(net.variables) == sorted(p)     # same op list, same variable names
    net.load_params(p)
    net.dropout_seed = 5
    xd, yd = dev(x), dev(y, torch.int32)
    logits = net.forward(xd, num_classes=ncls, is_training=True)
    losses, names, _ = net.loss_function('TOWER', yd, **logits)
    net.backward()
    torch.cuda.synchronize()
    mask = host(net.t['features_drop/mask'])
    kink = _kink(net)
    bands = og.noise_bands(graph, p, x, {'features_drop': mask}, state)      # the oracle's own fp32 noise: decision bands
    l_ref, g_ref, env, new_state = og.loss_and_grads(graph, p, x, y, 5e-4, masks={'features_drop': mask}, state=state, kink=kink, bands=bands)
    # fp32's own noise floor on THIS input: the same oracle evaluated in float32.  Batch norm over few samples
    # amplifies rounding noise layer after layer (ResNet-50 at 6x64x48 ends with 24 samples per channel and the
    # float32 oracle is off by 7e-5 at the features), so each tensor is held to max(base, 2 x that floor).
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    s32 = {k: v.astype(np.float32) for k, v in state.items()}
    _, g32, env32, _ = og.loss_and_grads(graph, p32, x.astype(np.float32), y, np.float32(5e-4),
                                         masks={'features_drop': mask.astype(np.float32)}, state=s32, kink=kink, bands=bands)

    def rel(a, b):
        return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
    for name in ('features', 'logits'):
        got = host(net.t[name])[:, :env[name].shape[1]]
        assert rel(got, env[name]) <= max(2e-5, 2 * rel(env32[name], env[name])), name
    assert abs(float(losses[0]) - l_ref[0]) <= 1e-4 * max(1, l_ref[0]) and abs(float(losses[1]) - l_ref[1]) <= 1e-5 * max(1, l_ref[1])
    for k in p:
        got = host(net.get_variable(k, net.grads)) + (5e-4 * p[k] if k.endswith('/weights') else 0)     # + wd*w (folded into the optimizer)
        assert rel(got, g_ref[k]) <= max(1e-4, 2 * rel(g32[k], g_ref[k])), ('grad ' + k, rel(got, g_ref[k]), rel(g32[k], g_ref[k]))
    for k in new_state:                                          # moving statistics (decay 0.999, unbiased variance)
        got = host(net.get_variable(k))
        # (1-decay) * batch statistic; a batch MEAN of zero-centred activations is a cancelling sum, so its
        # error is judged against the activation scale (1e-3 * 1e-6 of O(1) values), not against itself
        # ... and (1 - 0.999f) = 0.00099998713 in fp32 (-1.29e-5 relative), as in TF's own fp32 kernel
        assert np.abs(got - new_state[k]).max() <= 3e-5 * np.abs(new_state[k]).max() + 1e-9, k


def test_resnet_training_steps_and_eval_mode():
    ncls, n, h, w = 10, 8, 64, 64
    net = net_select('ResNet-50', 'NCHW', 5e-4)
    rng = np.random.default_rng(1)
    x = dev(rng.uniform(-1, 1, (n, h, w, 3))); y = dev(rng.integers(0, ncls, n), torch.int32)
    step, losses, names, _ = Singular(net, 0.01, 'Momentum')({'images': x, 'labels': y, 'num_classes': ncls, 'num_examples': n})
    w0 = net.params.clone()
    hist = []
    for i in range(40):
        step()
        hist.append(float(losses[0]))
    assert all(np.isfinite(hist)) and not torch.equal(w0, net.params)
    assert np.mean(hist[-8:]) < np.mean(hist[:8])                             # it fits the batch (dropout makes single steps noisy)
    assert names == ['cross_entropy', 'reg_loss'] and [len(g) for g in net.param_list(True, True)] == [159, 1]
    out = net.forward(x, num_classes=ncls, is_training=False)['logits']      # moving statistics, no dropout
    assert out.shape == (n, ncls) and torch.isfinite(out).all()
    mm = net.get_variable('ResNet-50/conv1/conv_7x7/BatchNorm/moving_mean')
    assert float(mm.abs().max()) > 0                                          # UPDATE_OPS ran


def _run_variant(net, graph, spec, n, h, w, ncls, labels, seed, center=None, triplet='off'):
    p, state = og.init_params(spec, seed)
    p = og.perturb(p, seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = rng.uniform(-1, 1, (n, h, w, 3))
    net.build(h, w, 3, ncls, 'cuda')
    assert net.graph == graph and sorted(net.variables) == sorted(p)
    net.load_params(p)
    if center is not None:
        net._centers().copy_(torch.tensor(center['centers'], dtype=torch.float32))
    xd, yd = dev(x), dev(labels, torch.int32)
    out = net.forward(xd, num_classes=ncls, is_training=True)
    losses, names, _ = net.loss_function('TOWER', yd, **out)
    net.backward()
    torch.cuda.synchronize()
    masks = {'features_drop': host(net.t['features_drop/mask'])} if net.has_classifier else None
    kink = _kink(net)
    bands = og.noise_bands(graph, p, x, masks, state)
    ref = og.loss_and_grads(graph, p, x, labels, 5e-4, masks=masks, state=state, kink=kink, center=center, triplet_margin=triplet, bands=bands)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    s32 = {k: v.astype(np.float32) for k, v in state.items()}
    c32 = None if center is None else dict(center, centers=center['centers'].astype(np.float32))
    m32 = None if masks is None else {k: v.astype(np.float32) for k, v in masks.items()}
    r32 = og.loss_and_grads(graph, p32, x.astype(np.float32), labels, np.float32(5e-4), masks=m32, state=s32, kink=kink, center=c32, triplet_margin=triplet, bands=bands)

    def rel(a, b):
        return float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
    assert rel(host(net.t['features']), ref[2]['features']) <= max(2e-5, 2 * rel(r32[2]['features'], ref[2]['features']))
    got_losses = [float(v) for v in losses]
    assert len(got_losses) == len(ref[0])
    for a, b in zip(got_losses, ref[0]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (names, got_losses, ref[0])
    for k in p:
        got = host(net.get_variable(k, net.grads)) + (5e-4 * p[k] if k.endswith('/weights') else 0)
        assert rel(got, ref[1][k]) <= max(1e-4, 2 * rel(r32[1][k], ref[1][k])), ('grad ' + k, rel(got, ref[1][k]), rel(r32[1][k], ref[1][k]))
    return ref, names


def test_resnext_with_center_loss():
    from tf_face_toolbox_amd.nets.resnet import ResNeXt
    n, h, w, ncls = 8, 64, 64, 12
    graph, spec = og.resnet_train_graph(26, 3, ncls, 'resnext')
    rng = np.random.default_rng(5)
    labels = rng.integers(0, ncls, n); labels[1] = labels[0]                 # a duplicate label: scatter_sub accumulates
    cen = rng.standard_normal((ncls, 2048)) * 0.1
    net = ResNeXt(26, head='softmax+center', center_weight=0.05)
    ref, names = _run_variant(net, graph, spec, n, h, w, ncls, labels, 81, center=dict(centers=cen, alpha=0.99, weight=0.05))
    assert names == ['cross_entropy', 'center_loss', 'reg_loss']
    check_maxabs(host(net.state['centers']), ref[4]['centers'], 1e-5, 'centers after the update')


@pytest.mark.parametrize('head', ['softmax', 'triplet'])
def test_se_resnet(head):
    from tf_face_toolbox_amd.nets.resnet import SENet
    n, h, w, ncls = 8, 64, 48, 9
    graph, spec = og.resnet_train_graph(26, 3, ncls, 'senet', classifier=(head == 'softmax'))
    labels = np.repeat(np.arange(4), 2) if head == 'triplet' else np.random.default_rng(6).integers(0, ncls, n)   # P x K = 4 x 2
    net = SENet(26, head=head)
    ref, names = _run_variant(net, graph, spec, n, h, w, ncls, labels, 91, triplet=(None if head == 'triplet' else 'off'))
    assert names == (['triplet_loss', 'reg_loss'] if head == 'triplet' else ['cross_entropy', 'reg_loss'])


def test_factory_nets_train_through_singular():
    rng = np.random.default_rng(2)
    n, h, w, ncls = 8, 64, 64, 10
    x = dev(rng.uniform(-1, 1, (n, h, w, 3)))
    for name, labels in (('ResNeXt-50-center', rng.integers(0, ncls, n)), ('SENet-50-triplet', np.repeat(np.arange(4), 2))):
        net = net_select(name, 'NCHW', 5e-4)
        step, losses, names, _ = Singular(net, 0.01, 'Momentum')({'images': x, 'labels': dev(labels, torch.int32), 'num_classes': ncls, 'num_examples': n})
        w0 = net.params.clone()
        for _ in range(3):
            step()
        assert all(np.isfinite(float(v)) for v in losses) and not torch.equal(w0, net.params), name
