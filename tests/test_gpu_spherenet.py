"""-m gpu: the whole hot path (SphereNet-20 forward / loss / backward / optimizer through the
reference-shaped Python API and the C ABI) against the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import ops, spherenet as osn

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host, check_maxabs, check_rell2, kink_of
    from tf_face_toolbox_amd import net_select, Singular


def _setup(name, data_format, n, h, w, ch, ncls, seed=21):
    p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    net = net_select(name, data_format, 5e-4)
    net.build(h, w, ch, ncls, 'cuda')
    net.load_params(p)
    return net, p, x, y


@pytest.mark.parametrize('name,data_format,n,h,w,ch,ncls', [
    ('SphereNet', 'NCHW', 4, 32, 32, 3, 10),
    ('SphereNet', 'NHWC', 3, 48, 16, 1, 200),
    ('SphereNet-ASoftmax', 'NCHW', 4, 32, 32, 3, 10),
    ('SphereNet', 'NCHW', 2, 112, 112, 3, 1000),          # BASELINE geometry, small batch
    ('SphereNet-ASoftmax', 'NCHW', 2, 112, 112, 1, 10575),  # config[0] geometry (gray, C = 10575)
])
def test_forward_loss_and_every_gradient(name, data_format, n, h, w, ch, ncls):
    net, p, x, y = _setup(name, data_format, n, h, w, ch, ncls)
    head = 'asoftmax' if 'ASoftmax' in name else 'softmax'
    lam = ops.asoftmax_lambda(0)
    xd, yd = dev(x), dev(y, torch.int32)
    net.tower_scale = 1.0
    if net.needs_labels:
        logits = net.forward(xd, yd, num_classes=ncls, is_training=True)
    else:
        logits = net.forward(xd, num_classes=ncls, is_training=True)
    losses, names, others = net.loss_function('TOWER', yd, **logits)
    net.backward()
    torch.cuda.synchronize()
    losses_ref, g_ref, ex = osn.loss_and_grads(p, x, y, 5e-4, data_format, head, lam, kink=kink_of(net))
    assert names == ['cross_entropy', 'reg_loss']
    check_maxabs(host(net.emb), ex['embedding'], what='embedding')
    check_maxabs(host(logits['logits']), ex['logits'], what='logits')
    assert abs(float(losses[0]) - losses_ref[0]) <= 1e-5 * max(1, abs(losses_ref[0]))
    assert abs(float(losses[1]) - losses_ref[1]) <= 1e-5 * max(1, abs(losses_ref[1]))
    for k in p:
        data_grad = g_ref[k] - (5e-4 * p[k] if k.endswith('/weights') else 0)     # wd*w is folded into the optimizer
        check_rell2(host(net.get_variable(k, net.grads)), data_grad, what='grad ' + k)


def test_config1_at_its_own_batch_64_gray_112():
    """BASELINE.json configs[0] at ITS size -- SphereFaceNet-20 + A-softmax, 64 gray 112x112 images, 10575 classes, one replica:
    embeddings / logits / losses against BOTH float64 oracles (numpy oracle/spherenet.py and the independent torch-autograd
    restatement oracle/torch_ref.py, which bench.py times as the CPU baseline) and every gradient against the numpy oracle
    (kink-band elements resolved as everywhere else).  At 64 images the backward walk runs on two streams (nets/sphere.py)."""
    from oracle import torch_ref
    n, h, w, ch, ncls = 64, 112, 112, 1, 10575
    net, p, x, y = _setup('SphereNet-ASoftmax', 'NCHW', n, h, w, ch, ncls, seed=51)
    lam = ops.asoftmax_lambda(0)
    xd, yd = dev(x), dev(y, torch.int32)
    net.tower_scale = 1.0
    logits = net.forward(xd, yd, num_classes=ncls, is_training=True)
    losses, names, others = net.loss_function('TOWER', yd, **logits)
    net.backward()
    torch.cuda.synchronize()
    tp = torch_ref.to_torch(p, torch.float64, requires_grad=False)
    with torch.no_grad():
        ce_t, reg_t, emb_t, log_t = torch_ref.spherenet_loss(tp, torch.from_numpy(x), torch.from_numpy(y), 5e-4, 'NCHW', 'asoftmax', float(lam))
    check_maxabs(host(net.emb), emb_t.numpy(), what='embedding vs torch_ref')
    check_maxabs(host(logits['logits']), log_t.numpy(), what='logits vs torch_ref')
    assert abs(float(losses[0]) - float(ce_t)) <= 1e-5 * max(1, abs(float(ce_t)))
    assert abs(float(losses[1]) - float(reg_t)) <= 1e-5 * max(1, abs(float(reg_t)))
    losses_ref, g_ref, ex = osn.loss_and_grads(p, x, y, 5e-4, 'NCHW', 'asoftmax', lam, kink=kink_of(net))
    check_maxabs(host(net.emb), ex['embedding'], what='embedding')
    check_maxabs(host(logits['logits']), ex['logits'], what='logits')
    assert abs(float(losses[0]) - losses_ref[0]) <= 1e-5 * max(1, abs(losses_ref[0]))
    for k in p:
        data_grad = g_ref[k] - (5e-4 * p[k] if k.endswith('/weights') else 0)
        check_rell2(host(net.get_variable(k, net.grads)), data_grad, what='grad ' + k)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_filter_gradients_on_the_second_stream_are_bit_identical(mode):
    """SphereNet's backward walk launches wgrad(l) on a second stream beside dgrad(l) (nets/sphere.py _body_walk): the same kernels on
    the same operands, so the whole gradient arena equals the one-stream walk bit for bit -- at a size where kernels really overlap
    (16 x 112 x 112), three backward passes in a row (the dz ping-pong buffers are reused across layers and steps)."""
    from tf_face_toolbox_amd import _lib
    prev = _lib.get_mfma_dtype()
    _lib.set_mfma_dtype(mode)
    try:
        arenas = []
        for two_streams in (True, False):
            net, p, x, y = _setup('SphereNet-ASoftmax', 'NCHW', 16, 112, 112, 3, 100)
            xd, yd = dev(x), dev(y, torch.int32)
            net.tower_scale = 1.0
            for it in range(3):
                logits = net.forward(xd, yd, num_classes=100, is_training=True)
                if not two_streams:
                    assert it > 0 or net._side_stream(16) is not None       # 16 per GPU: the two-stream walk is the default
                    net.side, net.ws_side = None, net.ws
                net.loss_function('TOWER', yd, **logits)
                net.backward()
            torch.cuda.synchronize()
            arenas.append(net.grads.clone())
        assert torch.equal(arenas[0], arenas[1])
        assert float(arenas[0].abs().max()) > 0
    finally:
        _lib.set_mfma_dtype(prev)


@pytest.mark.parametrize('optimizer', ['Momentum', 'Adam'])
def test_three_training_steps_match_oracle(optimizer):
    n, h, w, ch, ncls = 4, 32, 32, 3, 10
    net, p, x, y = _setup('SphereNet', 'NCHW', n, h, w, ch, ncls, seed=31)
    inputs = {'images': dev(x), 'labels': dev(y, torch.int32), 'num_classes': ncls, 'num_examples': n}
    lr = 0.05 if optimizer == 'Momentum' else 1e-3
    step, losses, names, others = Singular(net, lr, optimizer)(inputs)
    slots = osn.zero_slots(p, optimizer)
    for t in range(1, 4):
        step()
        p, slots, l_ref = osn.train_step(p, slots, x, y, lr, optimizer=optimizer, t=t, kink=kink_of(net))
        assert abs(float(losses[0]) - l_ref[0]) <= 1e-5 * max(1, abs(l_ref[0])), (t, float(losses[0]), l_ref)
    for k in p:
        tol = 2e-5 if optimizer == 'Momentum' else 2e-3      # Adam's m/sqrt(v) is ~sign(g): fp32 noise in tiny grads is amplified
        check_maxabs(host(net.get_variable(k)), p[k], tol, what='weights after 3 steps ' + k)


def test_eval_features_flip_average():
    net, p, x, y = _setup('SphereNet', 'NCHW', 3, 32, 32, 3, 10, seed=41)
    f = net.forward(dev(x), is_training=False)
    check_maxabs(host(f), osn.eval_features(p, x, 'NCHW'), what='eval features')


def test_variable_names_and_groups_follow_the_reference():
    net, p, x, y = _setup('SphereNet', 'NCHW', 2, 32, 32, 3, 10)
    assert list(sorted(net.variables)) == list(sorted(p))             # 47 TF variable names
    groups = net.param_list(is_training=True, trainable=True)
    assert len(groups) == 2 and [v.name for v in groups[1]] == ['classifier/fc_classifier/weights']
    assert net.mult_lr_list() == [1.0, 1.0]
    assert all(v.name.startswith('SphereNet/') for v in net.pretrained_param())
    for k in p:                                                       # import/export round trip in reference layout
        np.testing.assert_allclose(host(net.get_variable(k)), p[k].astype(np.float32), rtol=0, atol=0)
