"""CPU: the oracle must keep reproducing the committed golden fixtures (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import ops, spherenet as osn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'sphere_*.npz'))))
def test_oracle_reproduces_spherenet_goldens(path):
    g = np.load(path)
    seed, n, h, w, ch, ncls = [int(v) for v in g['meta']]
    if h * w > 64 * 64 and os.environ.get('FTE_FULL_GOLDEN') != '1':
        pytest.skip('112x112 golden is checked on the GPU box (set FTE_FULL_GOLDEN=1 to run it on CPU too)')
    p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
    x = g['images'].astype(np.float64)          # stored as float32: exactly what the HIP path sees
    rng = np.random.default_rng(seed + 2)
    x64 = rng.uniform(-1, 1, (n, h, w, ch))
    assert np.abs(x64.astype(np.float32) - g['images']).max() == 0
    losses, gr, ex = osn.loss_and_grads(p, x64, g['labels'], 5e-4, str(g['data_format']), str(g['head']), float(g['lam']))
    np.testing.assert_allclose(ex['embedding'], g['embedding'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ex['logits'], g['logits'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(losses, g['losses'], rtol=1e-13)
    for k in p:
        np.testing.assert_allclose(gr[k].reshape(-1)[g['gidx/' + k]], g['gval/' + k], rtol=1e-10, atol=1e-14)
        assert abs(np.sqrt((gr[k] ** 2).sum()) - g['gl2/' + k]) <= 1e-11 * max(1, g['gl2/' + k])


def test_oracle_reproduces_head_goldens():
    g = np.load(os.path.join(GOLD, 'heads.npz'))
    x, w, y = g['x'], g['w'], g['y']
    for lam in (5.0, 1000.0):
        loss, f, dx, dw = ops.asoftmax_fwd_bwd(x, w, y, lam)
        assert abs(loss - g['asm_loss_%g' % lam]) < 1e-13
        np.testing.assert_allclose(dx, g['asm_dx_%g' % lam], atol=1e-14)
        np.testing.assert_allclose(dw, g['asm_dw_%g' % lam], atol=1e-14)
    loss, d = ops.softmax_ce(x @ w, y)
    assert abs(loss - g['ce_loss']) < 1e-13
    cl, cdf, cnew = ops.center_loss(x, y, g['centers'], 0.99)
    np.testing.assert_allclose(cnew, g['center_new'], atol=1e-14)
    for m in (None, 0.3):
        tl, tg = ops.batch_hard_triplet(x, g['tri_labels'], m)
        np.testing.assert_allclose(tl, g['tri_loss_%s' % m], atol=1e-13)
        np.testing.assert_allclose(tg, g['tri_grad_%s' % m], atol=1e-13)
    steps = g['lr_steps']
    np.testing.assert_allclose([ops.lr_step(s, 0.1, 0.1, ['3', '5', '9'], 100) for s in steps], g['lr_step'], rtol=1e-15)
    np.testing.assert_allclose([ops.lr_exp(s, 0.1, 2, 12, 100) for s in steps], g['lr_exp'], rtol=1e-15)
    np.testing.assert_allclose([ops.lr_cosine(s, 0.1, 12, 100) for s in steps], g['lr_cos'], rtol=1e-15, atol=1e-18)


GRAPH_CASES = ['resnet26', 'resnext26_center', 'senet26_triplet', 'shufflenet_small_focal']


def graph_case_setup(g, tag):
    """Shared with tests/test_gpu_golden.py: rebuilds graph / params / inputs of one graphnets.npz case from its seed."""
    from oracle import graphnet as og
    seed, n, h, w, ncls = [int(v) for v in g[tag + '/meta']]
    graph, spec = {'resnet26': lambda: og.resnet_train_graph(26, 3, ncls),
                   'resnext26_center': lambda: og.resnet_train_graph(26, 3, ncls, 'resnext'),
                   'senet26_triplet': lambda: og.resnet_train_graph(26, 3, ncls, 'senet', classifier=False),
                   'shufflenet_small_focal': lambda: og.shufflenet_train_graph('small', 3, ncls, 'NCHW', blocks_override=[1, 1, 1])}[tag]()
    p, state = og.init_params(spec, seed)
    p = og.perturb(p, seed + 1)
    kw = {}
    if tag == 'resnext26_center':
        kw['center'] = dict(centers=g[tag + '/centers'], alpha=0.99, weight=0.05)
    if tag == 'senet26_triplet':
        kw['triplet_margin'] = None
    if tag == 'shufflenet_small_focal':
        kw['focal'] = (1.0, 2.0)
    masks = {'features_drop': g[tag + '/mask'].astype(np.float64)} if tag + '/mask' in g.files else None
    return graph, spec, p, state, g[tag + '/images'].astype(np.float64), g[tag + '/labels'].astype(np.int64), masks, kw


@pytest.mark.parametrize('tag', GRAPH_CASES)
def test_oracle_reproduces_graphnet_goldens(tag):
    from oracle import graphnet as og
    g = np.load(os.path.join(GOLD, 'graphnets.npz'))
    graph, spec, p, state, x, y, masks, kw = graph_case_setup(g, tag)
    res = og.loss_and_grads(graph, p, x, y, 5e-4, masks=masks, state=state, **kw)
    np.testing.assert_allclose(np.array(res[0]), g[tag + '/losses'], rtol=1e-9)        # images were stored as float32
    np.testing.assert_allclose(res[2]['features'], g[tag + '/features'], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose([np.sqrt((res[1][k] ** 2).sum()) for k in sorted(res[1])], g[tag + '/gl2'], rtol=1e-6, atol=1e-12)   # atol: exact-zero gradients (a BN beta in front of conv -> BN) are 1e-18 noise
