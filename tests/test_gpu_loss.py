"""-m gpu: tf_face_toolbox_amd.loss (the reference's loss.py API) and the focal head, against the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import graphnet as og, ops

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from util_gpu import dev, host, check_maxabs, check_rell2
    from tf_face_toolbox_amd import loss as L


@pytest.mark.parametrize('n,c,ld,gamma,alpha', [(8, 10, 128, 1.0, 2.0), (37, 1000, 1024, 0.5, 1.0), (5, 10575, 10624, 2.0, 3.5)])
def test_focal_loss(n, c, ld, gamma, alpha):
    rng = np.random.default_rng(n + c)
    z = rng.standard_normal((n, c)) * 3; y = rng.integers(0, c, n)
    zp = np.zeros((n, ld)); zp[:, :c] = z
    ref_loss, ref_d = ops.focal_loss(z, y, gamma, alpha)
    loss, d = L.focal_loss(dev(zp), dev(y, torch.int32), gamma, alpha, num_classes=c)
    assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, ref_loss)
    check_rell2(host(d)[:, :c], ref_d, 2e-5, 'dlogits')
    assert float(d[:, c:].abs().max()) == 0.0 if ld > c else True


def test_center_loss_and_triplet_functions():
    rng = np.random.default_rng(3)
    n, d, ncls = 12, 256, 7
    f = rng.standard_normal((n, d)); y = rng.integers(0, ncls, n); y[3] = y[0]
    cen = rng.standard_normal((ncls, d)) * 0.1
    ref_loss, ref_df, ref_c = ops.center_loss(f, y, cen, 0.9)
    cd = dev(cen)
    loss, cout, df = L.center_loss(dev(f), dev(y, torch.int32), ncls, alpha=0.9, weight=0.5, centers=cd)
    assert cout is cd and abs(float(loss) - ref_loss) <= 1e-5 * ref_loss
    check_rell2(host(df), 0.5 * ref_df, 2e-5, 'd(weight*center_loss)/dfeatures')
    check_maxabs(host(cd), ref_c, 1e-5, 'centers after scatter_sub')
    labels = np.repeat(np.arange(4), 3)
    for margin in (None, 0.3, -1.0, -0.25):          # negative margins are hinges (loss.py:76-77), only None is the softplus
        per, dfe = ops.batch_hard_triplet(f, labels, margin)
        rows, dft = L.batch_hard_triplet_loss(dev(f), dev(labels, torch.int32), margin)
        check_maxabs(host(rows), per, 2e-5, 'per-sample triplet loss')
        check_rell2(host(dft), dfe / n, 2e-5, 'd(mean)/dfeatures')
    # `metric` is ignored exactly as the reference ignores it (loss.py:65 never forwards it to cdist): euclidean whatever is passed
    per, dfe = ops.batch_hard_triplet(f, labels, 0.3)
    rows, dft = L.batch_hard_triplet_loss(dev(f), dev(labels, torch.int32), 0.3, metric='cityblock')
    check_maxabs(host(rows), per, 2e-5, 'per-sample triplet loss, metric argument ignored')


def test_focal_head_on_a_graph_net():
    from tf_face_toolbox_amd.nets.resnet import ResNeXt
    from test_gpu_resnet import _kink
    n, h, w, ncls = 8, 48, 48, 9
    graph, spec = og.resnet_train_graph(26, 3, ncls, 'resnext')
    p, state = og.init_params(spec, 5); p = og.perturb(p, 6)
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (n, h, w, 3)); y = rng.integers(0, ncls, n)
    net = ResNeXt(26, head='focal'); net.focal_gamma, net.focal_alpha = 1.5, 2.0
    net.build(h, w, 3, ncls, 'cuda'); net.load_params(p)
    out = net.forward(dev(x), num_classes=ncls, is_training=True)
    losses, names, _ = net.loss_function('T', dev(y, torch.int32), **out)
    net.backward(); torch.cuda.synchronize()
    assert names == ['focal_entropy', 'reg_loss']
    mask = host(net.t['features_drop/mask'])
    kink = _kink(net)
    bands = og.noise_bands(graph, p, x, {'features_drop': mask}, state)
    ref = og.loss_and_grads(graph, p, x, y, 5e-4, masks={'features_drop': mask}, state=state, kink=kink, focal=(1.5, 2.0), bands=bands)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}; s32 = {k: v.astype(np.float32) for k, v in state.items()}
    r32 = og.loss_and_grads(graph, p32, x.astype(np.float32), y, np.float32(5e-4), masks={'features_drop': mask.astype(np.float32)}, state=s32, kink=kink, focal=(1.5, 2.0), bands=bands)
    assert abs(float(losses[0]) - ref[0][0]) <= 1e-4 * max(1.0, ref[0][0])

    def rel(a, b):
        return float(np.sqrt(((np.asarray(a, np.float64) - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
    for k in p:
        got = host(net.get_variable(k, net.grads)) + (5e-4 * p[k] if k.endswith('/weights') else 0)
        assert rel(got, ref[1][k]) <= max(1e-4, 2 * rel(r32[1][k], ref[1][k])), k
