"""Pins the graph oracle (BN, max-pool, GAP, dropout, bottleneck wiring) against an independent
torch-CPU float64 ResNet written in NCHW with autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import graphnet as og, ops


def _same_pad(x, k, s):
    def pads(n):
        out = (n + s - 1) // s
        tot = max((out - 1) * s + k - n, 0)
        return tot // 2, tot - tot // 2
    pt, pb = pads(x.shape[2]); pl, pr = pads(x.shape[3])
    return F.pad(x, (pl, pr, pt, pb)), (pt, pb, pl, pr)


def torch_resnet(tp, images_nhwc, labels, num_layers, mask, wd):
    name = 'ResNet-%d' % num_layers
    x = images_nhwc.permute(0, 3, 1, 2)

    def conv_bn(scope, x, k, stride, relu):
        w = tp[scope + '/weights'].permute(3, 2, 0, 1)
        xp, _ = _same_pad(x, k, stride)
        z = F.conv2d(xp, w, None, stride=stride)
        y = F.batch_norm(z, None, None, tp[scope + '/BatchNorm/gamma'], tp[scope + '/BatchNorm/beta'], True, 0.0, 1e-3)
        return torch.relu(y) if relu else y
    x = conv_bn(name + '/conv1/conv_7x7', x, 7, 2, True)
    xp = F.pad(x, (0, 1, 0, 1), value=float('-inf')) if x.shape[2] % 2 == 0 else F.pad(x, (1, 1, 1, 1), value=float('-inf'))
    x = F.max_pool2d(xp, 3, 2)
    cin = 64
    for si, nb in enumerate(og.RESNET_BLOCKS[num_layers]):
        cout = og.RESNET_OUTPUTS[si]
        for b in range(nb):
            stride = 2 if (b == 0 and si > 0) else 1
            sc = '%s/conv%d/resBlock_%d' % (name, si + 2, b)
            shortcut = x
            if stride != 1 or cin != cout:
                shortcut = conv_bn(sc + '/conv_shortcut_1x1', x, 1, stride, False)
            y = conv_bn(sc + '/conv1_1x1', x, 1, 1, True)
            y = conv_bn(sc + '/conv2_3x3', y, 3, stride, True)
            y = conv_bn(sc + '/conv3_1x1', y, 1, 1, False)
            x = torch.relu(y + shortcut)
            cin = cout
    feat = x.mean(dim=(2, 3))
    logits = (feat * mask / 0.5) @ tp['classifier/fc_classifier/weights']
    ce = F.cross_entropy(logits, labels)
    reg = sum(wd * (v ** 2).sum() / 2 for k, v in tp.items() if k.endswith('/weights'))
    return ce, reg, feat, logits


@pytest.mark.parametrize('h,w', [(32, 32), (40, 24)])
def test_resnet26_grads_match_torch_autograd(h, w):
    nl, n, ncls = 26, 3, 5
    graph, spec = og.resnet_train_graph(nl, 3, ncls)
    p, state = og.init_params(spec, 7)
    p = og.perturb(p, 8)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (n, h, w, 3)); y = rng.integers(0, ncls, n)
    mask = (rng.random((n, 2048)) < 0.5).astype(np.float64)
    losses, g, env, new_state = og.loss_and_grads(graph, p, x, y, 5e-4, masks={'features_drop': mask}, state=state)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    ce, reg, feat, logits = torch_resnet(tp, torch.tensor(x), torch.tensor(y), nl, torch.tensor(mask), 5e-4)
    (ce + reg).backward()
    assert abs(losses[0] - ce.item()) < 1e-11 and abs(losses[1] - reg.item()) < 1e-11
    np.testing.assert_allclose(env['features'], feat.detach().numpy(), atol=1e-11)
    assert set(g) == set(tp)
    for k in tp:
        ref = tp[k].grad.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k
    # moving statistics: decay 0.999, unbiased variance
    k0 = 'ResNet-26/conv1/conv_7x7/BatchNorm'
    z = env['conv1/z']
    cnt = z.shape[0] * z.shape[1] * z.shape[2]
    np.testing.assert_allclose(new_state[k0 + '/moving_mean'], 0.001 * z.mean((0, 1, 2)), atol=1e-14)
    np.testing.assert_allclose(new_state[k0 + '/moving_variance'], 0.999 + 0.001 * z.var((0, 1, 2)) * cnt / (cnt - 1), atol=1e-14)


def test_resnet50_graph_shape_facts():
    g, spec = og.resnet_train_graph(50, 3, 10)
    convs = [op for op in g if op[0] == 'conv']
    assert len(convs) == 1 + 16 * 3 + 4                      # stem + 16 bottlenecks + 4 projection shortcuts
    nparams = sum(int(np.prod(s)) for n, s, k in spec if k == 'conv_w')
    assert nparams == 23454912                               # 23.45 M conv params
    p, st = og.init_params(spec, 0)
    x = np.zeros((1, 112, 112, 3))
    env, _, _ = og.forward(g[:-2], p, x, train=True, state=st)
    assert env['s5b2'].shape == (1, 4, 4, 2048) and env['pool1'].shape == (1, 28, 28, 64)


def test_maxpool_matches_torch_with_ties():
    x = np.random.default_rng(0).integers(0, 3, (2, 7, 6, 4)).astype(np.float64)     # many ties
    y, c = ops.maxpool3x3s2_fwd(x)
    tx = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    xp = F.pad(tx, (0, 1, 1, 1), value=float('-inf'))        # h=7: pads (1,1); w=6: pads (0,1)
    ty = F.max_pool2d(xp, 3, 2)
    np.testing.assert_array_equal(y, ty.detach().permute(0, 2, 3, 1).numpy())
    dy = np.random.default_rng(1).standard_normal(y.shape)
    ty.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    np.testing.assert_allclose(ops.maxpool3x3s2_bwd(dy, c), tx.grad.permute(0, 2, 3, 1).numpy(), atol=1e-14)


def test_gconv_and_se_ops_match_torch_autograd():
    """Oracle 'gconv' (split / conv / concat) and 'se' (squeeze-excitation) ops vs torch (groups=, autograd)."""
    rng = np.random.default_rng(3)
    n, h, w, c, groups = 2, 7, 6, 64, 8
    gw = c // groups
    x = rng.standard_normal((n, h, w, c)); wt = rng.standard_normal((groups, 3, 3, gw, gw)) * 0.3
    p = {'w': wt, 'se/fc1/weights': rng.standard_normal((c, c // 2)) * 0.2, 'se/fc1/biases': rng.standard_normal(c // 2) * 0.1,
         'se/fc2/weights': rng.standard_normal((c // 2, c)) * 0.2, 'se/fc2/biases': rng.standard_normal(c) * 0.1}
    graph = [('gconv', 'a', 'images', 'w', 2, groups), ('se', 'b', 'a', 'se')]
    env, cache, _ = og.forward(graph, p, x)
    dy = rng.standard_normal(env['b'].shape)
    gp, gt = og.backward(graph + [], p, env, cache, {'b': dy})
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    tx = torch.tensor(x, requires_grad=True)
    xin = tx.permute(0, 3, 1, 2)
    xp, _ = _same_pad(xin, 3, 2)
    w_t = tp['w'].permute(0, 4, 3, 1, 2).reshape(c, gw, 3, 3)          # [G,3,3,in,out] -> [G*out, in, 3, 3]
    a = F.conv2d(xp, w_t, None, stride=2, groups=groups)
    sq = a.mean(dim=(2, 3))
    gate = torch.sigmoid(torch.relu(sq @ tp['se/fc1/weights'] + tp['se/fc1/biases']) @ tp['se/fc2/weights'] + tp['se/fc2/biases'])
    b = a * gate[:, :, None, None]
    np.testing.assert_allclose(env['b'], b.detach().permute(0, 2, 3, 1).numpy(), atol=1e-12)
    b.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    for k in tp:
        np.testing.assert_allclose(gp[k], tp[k].grad.numpy(), atol=1e-11, err_msg=k)
    # 'images' never receives a gradient from a conv in the oracle (need_dx is off for the stem); check through a proxy
    graph2 = [('relu', 'r', 'images')] + [('gconv', 'a', 'r', 'w', 2, groups), ('se', 'b', 'a', 'se')]
    env2, cache2, _ = og.forward(graph2, p, np.abs(x) + 0.1)
    _, gt2 = og.backward(graph2, p, env2, cache2, {'b': dy})
    tx2 = torch.tensor(np.abs(x) + 0.1, requires_grad=True)
    xp2, _ = _same_pad(tx2.permute(0, 3, 1, 2), 3, 2)
    a2 = F.conv2d(xp2, w_t.detach(), None, stride=2, groups=groups)
    g2 = torch.sigmoid(torch.relu(a2.mean(dim=(2, 3)) @ tp['se/fc1/weights'].detach() + tp['se/fc1/biases'].detach()) @ tp['se/fc2/weights'].detach() + tp['se/fc2/biases'].detach())
    (a2 * g2[:, :, None, None]).backward(torch.tensor(dy).permute(0, 3, 1, 2))
    np.testing.assert_allclose(gt2['images'], tx2.grad.numpy(), atol=1e-11)
