"""Pins the ShuffleNet-v2 oracle graph (depthwise conv, channel split / concat / shuffle wiring) against an
independent torch-CPU float64 net written in NCHW with autograd, following nets/shufflenet_v2.py literally
(tf.split -> branches -> tf.concat -> reshape/transpose/reshape)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import graphnet as og, ops
from test_oracle_resnet import _same_pad


def torch_shufflenet(tp, images_nhwc, labels, variant, mask, wd, blocks, fmt='NCHW'):
    name, stem_c, stages, final_c = og.SHUFFLENET[variant]
    se = variant == 'large'
    x = images_nhwc.permute(0, 3, 1, 2)

    def bn(scope, z, relu):
        y = F.batch_norm(z, None, None, tp[scope + '/BatchNorm/gamma'], tp[scope + '/BatchNorm/beta'], True, 0.0, 1e-3)
        return torch.relu(y) if relu else y

    def conv(scope, x, k, stride):
        xp, _ = _same_pad(x, k, stride)
        return bn(scope, F.conv2d(xp, tp[scope + '/weights'].permute(3, 2, 0, 1), None, stride=stride), True)

    def sep(scope, x, stride):
        c = x.shape[1]
        xp, _ = _same_pad(x, 3, stride)
        d = F.conv2d(xp, tp[scope + '/depthwise_weights'].permute(2, 3, 0, 1), None, stride=stride, groups=c)
        return bn(scope, F.conv2d(d, tp[scope + '/pointwise_weights'].permute(3, 2, 0, 1)), False)

    def shuffle(x):
        n, c, h, w = x.shape
        if fmt == 'NCHW':
            return x.reshape(n, 2, c // 2, h, w).permute(0, 2, 1, 3, 4).reshape(n, c, h, w)
        xh = x.permute(0, 2, 3, 1)                      # the reference's NHWC branch, evaluated on an NHWC view
        xh = xh.reshape(n, h, w, c // 2, 2).permute(0, 1, 2, 4, 3).reshape(n, h, w, c)
        return xh.permute(0, 3, 1, 2)

    if variant == 'large':
        x = conv(name + '/conv1/conv3_3x3', x, 3, 1)
    else:
        x = conv(name + '/conv1/conv_3x3', x, 3, 2)
    xp = F.pad(x, (0, 1, 0, 1), value=float('-inf')) if x.shape[2] % 2 == 0 else F.pad(x, (1, 1, 1, 1), value=float('-inf'))
    x = F.max_pool2d(xp, 3, 2)
    for (scope, _, c), nb in zip(stages, blocks):
        for b in range(nb):
            stride = 2 if b == 0 else 1
            sc = '%s/%s/resBlock_%d' % (name, scope, b)
            ch = x.shape[1]
            shortcut, y = torch.split(x, [int(0.5 * ch), ch - int(0.5 * ch)], dim=1)
            if stride != 1:
                shortcut = sep(sc + '/separable_conv_shortcut_3x3', shortcut, stride)
                shortcut = conv(sc + '/conv_shortcut_1x1', shortcut, 1, 1)
            y = conv(sc + '/conv1_1x1', y, 1, 1)
            y = sep(sc + '/separable_conv2_3x3', y, stride)
            y = conv(sc + '/conv3_1x1', y, 1, 1)
            if se:
                sq = y.mean(dim=(2, 3), keepdim=True)
                hid = torch.relu(F.conv2d(sq, tp[sc + '/Conv/weights'].permute(3, 2, 0, 1), tp[sc + '/Conv/biases']))
                gate = torch.sigmoid(F.conv2d(hid, tp[sc + '/Conv_1/weights'].permute(3, 2, 0, 1), tp[sc + '/Conv_1/biases']))
                y = y * gate
            x = shuffle(torch.cat([shortcut, y], dim=1))
    x = conv(name + '/conv5/conv_1x1', x, 1, 1)
    feat = x.mean(dim=(2, 3))
    logits = (feat * mask / 0.5) @ tp['classifier/fc_classifier/weights']
    ce = F.cross_entropy(logits, labels)
    reg = sum(wd * (v ** 2).sum() / 2 for k, v in tp.items() if k.endswith('weights'))
    return ce, reg, feat, logits


@pytest.mark.parametrize('variant,blocks,fmt,hw', [('small', [2, 2, 2], 'NCHW', (32, 32)), ('small', [2, 1, 2], 'NHWC', (40, 24)),
                                                   ('small_x1', [2, 2, 1], 'NCHW', (32, 32)), ('middle', [2, 1, 1, 2], 'NCHW', (32, 32)),
                                                   ('large', [2, 1, 1, 1], 'NCHW', (16, 16))])
def test_shufflenet_grads_match_torch_autograd(variant, blocks, fmt, hw):
    n, ncls = 3, 5
    graph, spec = og.shufflenet_train_graph(variant, 3, ncls, fmt, blocks_override=blocks)
    p, state = og.init_params(spec, 7)
    p = og.perturb(p, 8)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (n,) + hw + (3,)); y = rng.integers(0, ncls, n)
    fdim = og.SHUFFLENET[variant][3]
    mask = (rng.random((n, fdim)) < 0.5).astype(np.float64)
    losses, g, env, new_state = og.loss_and_grads(graph, p, x, y, 5e-4, masks={'features_drop': mask}, state=state)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    ce, reg, feat, logits = torch_shufflenet(tp, torch.tensor(x), torch.tensor(y), variant, torch.tensor(mask), 5e-4, blocks, fmt)
    (ce + reg).backward()
    assert abs(losses[0] - ce.item()) < 1e-11 and abs(losses[1] - reg.item()) < 1e-11
    np.testing.assert_allclose(env['features'], feat.detach().numpy(), atol=1e-11)
    dead = [k for k in tp if tp[k].grad is None]          # the large net's unused stem BN (gamma / beta see no loss term)
    assert set(g) == set(tp) - set(dead) and all('conv1/conv1_3x3' in k or 'conv1/conv2_3x3' in k for k in dead)
    for k in g:
        ref = tp[k].grad.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k


def test_shufflenet_small_shape_facts():
    """net_base.py:37-42: alpha = 2.0 -> widths 122 / 244 / 488, final 2048 (nets/shufflenet_v2.py:48-50)."""
    g, spec = og.shufflenet_train_graph('small', 3, 10)
    shapes = dict((n, s) for n, s, _ in spec)
    nm = 'ShuffleNet_v2_small_x2'
    assert shapes[nm + '/conv1/conv_3x3/weights'] == (3, 3, 3, 24)
    assert shapes[nm + '/conv2/resBlock_0/separable_conv_shortcut_3x3/depthwise_weights'] == (3, 3, 12, 1)
    assert shapes[nm + '/conv2/resBlock_0/separable_conv_shortcut_3x3/pointwise_weights'] == (1, 1, 12, 122)
    assert shapes[nm + '/conv2/resBlock_1/conv1_1x1/weights'] == (1, 1, 122, 122)
    assert shapes[nm + '/conv3/resBlock_0/conv1_1x1/weights'] == (1, 1, 122, 244)
    assert shapes[nm + '/conv5/conv_1x1/weights'] == (1, 1, 976, 2048)
    assert sum(1 for op in g if op[0] == 'dwconv') == 16 + 3
    p, st = og.init_params(spec, 0)
    env, _, _ = og.forward(g[:-2], p, np.zeros((1, 112, 112, 3)), train=True, state=st)
    assert env['pool1'].shape == (1, 28, 28, 24) and env['conv4b3'].shape == (1, 4, 4, 976) and env['features'].shape == (1, 2048)


def test_shuffle_then_split_is_an_interleave_of_half_branches():
    """The fused op the engine runs: split(shuffle(concat(a, b))) with the NCHW permutation gives
    s = interleave(a[:c/2], b[:c/2]), x = interleave(a[c/2:], b[c/2:])."""
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((2, 3, 3, 6)), rng.standard_normal((2, 3, 3, 6))
    s, x = ops.channel_split(ops.channel_shuffle(np.concatenate([a, b], -1), 'NCHW'))
    np.testing.assert_array_equal(s[..., 0::2], a[..., :3]); np.testing.assert_array_equal(s[..., 1::2], b[..., :3])
    np.testing.assert_array_equal(x[..., 0::2], a[..., 3:]); np.testing.assert_array_equal(x[..., 1::2], b[..., 3:])
