"""Pins the oracle (oracle/) against an INDEPENDENT second implementation and
finite differences.  The reference ships no golden vectors, so
this file is what stands between the oracle and "I restated it wrong":
  * torch-CPU float64 conv2d with explicit TF-SAME padding + autograd, written
    in NCHW (the reference's default data_format) without using oracle code;
  * central finite differences for the loss heads.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops, spherenet as sn


# ---------------------------------------------------------------- independent torch model
# oracle/torch_ref.py: written in NCHW from the reference's layer list, shares no code with the numpy oracle
from oracle.torch_ref import spherenet_loss as torch_spherenet_loss, train_step as torch_train_step, to_torch      # noqa: E402


@pytest.mark.parametrize('in_ch,hw,data_format', [(3, (32, 32), 'NCHW'), (1, (48, 16), 'NHWC')])
def test_spherenet_grads_match_torch_autograd(in_ch, hw, data_format):
    h, w = hw
    ncls = 7
    p = sn.perturb_params(sn.init_params(5, in_ch, ncls, h, w), 6)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (3, h, w, in_ch))
    y = rng.integers(0, ncls, 3)
    losses, g, ex = sn.loss_and_grads(p, x, y, 5e-4, data_format)

    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    ce, reg, emb, logits = torch_spherenet_loss(tp, torch.tensor(x), torch.tensor(y), 5e-4, data_format)
    (ce + reg).backward()
    assert abs(losses[0] - ce.item()) < 1e-12
    assert abs(losses[1] - reg.item()) < 1e-12 * max(1, reg.item())
    np.testing.assert_allclose(ex['embedding'], emb.detach().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(ex['logits'], logits.detach().numpy(), rtol=0, atol=1e-12)
    assert set(g) == set(tp)
    for k in tp:
        ref = tp[k].grad.numpy()
        err = np.abs(g[k] - ref).max()
        assert err <= 1e-11 * max(1.0, np.abs(ref).max()), (k, err)


@pytest.mark.parametrize('lam', [1000.0, 5.0, 0.0])
def test_asoftmax_step_matches_torch_autograd(lam):
    """The A-softmax head (hand-derived gradient through both norms, oracle/ops.py) and one Momentum step against
    torch autograd on oracle/torch_ref.py's independent restatement of SURVEY App. A.9."""
    h = w = 16
    ncls = 9
    p = sn.perturb_params(sn.init_params(15, 1, ncls, h, w), 16)
    p['classifier/fc_classifier/weights'] = p['classifier/fc_classifier/weights'] * 300      # cos(theta) spread over the k branches
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (6, h, w, 1))
    y = rng.integers(0, ncls, 6)
    losses, g, ex = sn.loss_and_grads(p, x, y, 5e-4, 'NCHW', 'asoftmax', lam)
    tp = to_torch(p, torch.float64)
    ce, reg, emb, logits = torch_spherenet_loss(tp, torch.tensor(x), torch.tensor(y), 5e-4, 'NCHW', 'asoftmax', lam)
    (ce + reg).backward()
    assert abs(losses[0] - ce.item()) < 1e-11 and abs(losses[1] - reg.item()) < 1e-11 * max(1, reg.item())
    np.testing.assert_allclose(ex['logits'], logits.detach().numpy(), rtol=0, atol=1e-11 * max(1, np.abs(ex['logits']).max()))
    for k in tp:
        ref = tp[k].grad.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), k
    # one full step (data_parallel.py:45-79): same new weights
    p2, s2, _ = sn.train_step(p, sn.zero_slots(p), x, y, 0.05, head='asoftmax', lam=lam)
    tp = to_torch(p, torch.float64)
    slots = {k: torch.zeros_like(v) for k, v in tp.items()}
    torch_train_step(tp, slots, torch.tensor(x), torch.tensor(y), 0.05, 5e-4, 'NCHW', 'asoftmax', lam)
    for k in p2:
        assert np.abs(tp[k].detach().numpy() - p2[k]).max() <= 1e-11 * max(1.0, np.abs(p2[k]).max()), k


def test_same_padding_is_asymmetric_for_stride2():
    # Appendix A.1: k=3 s=2 on even sizes pads (0,1); s=1 pads (1,1); 7 -> 4 pads (1,1)
    assert ops.same_pads(112, 3, 2) == (56, 0, 1)
    assert ops.same_pads(56, 3, 1) == (56, 1, 1)
    assert ops.same_pads(7, 3, 2) == (4, 1, 1)
    assert ops.same_pads(112, 7, 2) == (56, 2, 3)
    assert sn.feature_hw(112, 112) == (7, 7)


def test_algorithmic_flops_match_survey():
    tr, fw = sn.train_flops_per_image(3, 10575)
    assert fw == 2 * (2041360384 + 512 * 10575)
    assert tr == 2 * (3 * (2041360384 + 512 * 10575) - 5419008)
    names = sn.conv_layer_names()
    assert len(names) == 20 and sum(1 for n in names if n[3]) == 4
    assert len(sn.init_params(0, 1, 10, 16, 16)) == 47


def test_eval_features_flip_average():
    p = sn.init_params(3, 3, 5, 16, 16)
    x = np.random.default_rng(0).uniform(-1, 1, (2, 16, 16, 3))
    f = sn.eval_features(p, x)
    f2 = sn.eval_features(p, x[:, :, ::-1, :])
    np.testing.assert_allclose(f, f2, atol=1e-13)       # symmetric under the flip by construction
    e, _ = sn.backbone_fwd(p, x)
    assert np.abs(f - e).max() > 1e-6                   # and not equal to the plain embedding


# ---------------------------------------------------------------- train step / towers
def test_tower_split_equals_single_tower():
    """data_parallel.py:37,179: (1/n)*grad summed over n equal shards == full-batch mean grad."""
    p = sn.perturb_params(sn.init_params(7, 1, 6, 16, 16), 8)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (4, 16, 16, 1)); y = rng.integers(0, 6, 4)
    s = sn.zero_slots(p)
    p1, s1, l1 = sn.train_step(p, s, x, y, 0.1, num_towers=1)
    p2, s2, l2 = sn.train_step(p, s, x, y, 0.1, num_towers=2)
    p4, s4, l4 = sn.train_step(p, s, x, y, 0.1, num_towers=4)
    for k in p:
        np.testing.assert_allclose(p1[k], p2[k], rtol=0, atol=1e-13)
        np.testing.assert_allclose(p1[k], p4[k], rtol=0, atol=1e-13)
    np.testing.assert_allclose(l1, l2, atol=1e-13)
    # momentum: second step uses acc = 0.9*acc + g
    p1b, s1b, _ = sn.train_step(p1, s1, x, y, 0.1)
    k = 'SphereNet/conv1/Conv/alpha'
    _, g, _ = sn.loss_and_grads(p1, x, y)
    np.testing.assert_allclose(s1b[k], 0.9 * s1[k] + g[k], atol=1e-15)
    np.testing.assert_allclose(p1b[k], p1[k] - 0.1 * s1b[k], atol=1e-15)


def test_optimizers_match_torch():
    rng = np.random.default_rng(0)
    w = rng.standard_normal(50); g1 = rng.standard_normal(50); g2 = rng.standard_normal(50)
    tw = torch.tensor(w.copy(), requires_grad=True)
    opt = torch.optim.SGD([tw], lr=0.1, momentum=0.9)
    acc = np.zeros(50); wn = w
    for g in (g1, g2):
        tw.grad = torch.tensor(g); opt.step()
        wn, acc = ops.momentum_step(wn, acc, g, 0.1)
    np.testing.assert_allclose(wn, tw.detach().numpy(), atol=1e-14)
    # TF Adam (eps outside the bias correction, Appendix A.7) -- closed form for step 1
    wa, m, v = ops.adam_step(w, np.zeros(50), np.zeros(50), g1, 0.01, 1)
    m1 = 0.5 * g1; v1 = 0.001 * g1 * g1
    lr_t = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.5)
    np.testing.assert_allclose(wa, w - lr_t * m1 / (np.sqrt(v1) + 1e-8), atol=1e-15)


def test_lr_schedules():
    bpe = 100
    # train.py:127-129 step: boundaries (epoch-1)*bpe, piecewise_constant keeps the old value AT the boundary
    assert ops.lr_step(0, 0.1, 0.1, ['3', '5'], bpe) == 0.1
    assert ops.lr_step(200, 0.1, 0.1, ['3', '5'], bpe) == 0.1
    assert abs(ops.lr_step(201, 0.1, 0.1, ['3', '5'], bpe) - 0.01) < 1e-15
    assert abs(ops.lr_step(401, 0.1, 0.1, ['3', '5'], bpe) - 0.001) < 1e-15
    # train.py:131-138 exp
    assert ops.lr_exp(199, 0.1, 2, 10, bpe) == 0.1
    assert abs(ops.lr_exp(200, 0.1, 2, 10, bpe) - 0.1) < 1e-15
    assert abs(ops.lr_exp(1001, 0.1, 2, 10, bpe) - 0.1 * 0.001) < 1e-12
    # train.py:140 cosine
    assert abs(ops.lr_cosine(0, 0.1, 10, bpe) - 0.1) < 1e-15
    assert abs(ops.lr_cosine(500, 0.1, 10, bpe) - 0.05) < 1e-15
    assert abs(ops.lr_cosine(5000, 0.1, 10, bpe)) < 1e-15


# ---------------------------------------------------------------- loss heads
def _fd(fun, x, eps=1e-6):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=['multi_index'])
    for _ in it:
        i = it.multi_index
        o = x[i]
        x[i] = o + eps; fp = fun()
        x[i] = o - eps; fm = fun()
        x[i] = o
        g[i] = (fp - fm) / (2 * eps)
    return g


def test_softmax_ce_matches_torch():
    rng = np.random.default_rng(0)
    z = rng.standard_normal((5, 9)) * 3; y = rng.integers(0, 9, 5)
    loss, d = ops.softmax_ce(z, y)
    tz = torch.tensor(z, requires_grad=True)
    tl = F.cross_entropy(tz, torch.tensor(y)); tl.backward()
    assert abs(loss - tl.item()) < 1e-13
    np.testing.assert_allclose(d, tz.grad.numpy(), atol=1e-14)


@pytest.mark.parametrize('lam', [5.0, 1000.0, 0.0])
def test_asoftmax_grad_finite_difference(lam):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 8)); w = rng.standard_normal((8, 5)); y = rng.integers(0, 5, 6)
    loss, f, dx, dw = ops.asoftmax_fwd_bwd(x, w, y, lam)
    gx = _fd(lambda: ops.asoftmax_fwd_bwd(x, w, y, lam)[0], x)
    gw = _fd(lambda: ops.asoftmax_fwd_bwd(x, w, y, lam)[0], w)
    np.testing.assert_allclose(dx, gx, atol=2e-7)
    np.testing.assert_allclose(dw, gw, atol=2e-7)


def test_asoftmax_psi_known_answers():
    """psi(theta) = (-1)^k cos(4 theta) - 2k is continuous and monotone decreasing on [0, pi]."""
    d = 4
    w = np.eye(d)[:, :2].copy()
    vals = []
    for theta in np.linspace(0, np.pi, 181):
        x = np.zeros((1, d)); x[0, 0] = 2 * np.cos(theta); x[0, 2] = 2 * np.sin(theta)
        f, t = ops.asoftmax_logits(x, w, np.array([0]), 0.0)
        k = int(np.floor(4 * theta / np.pi)) if theta < np.pi else 3
        want = (-1) ** k * np.cos(4 * theta) - 2 * k
        if min(abs(4 * theta / np.pi - r) for r in (1, 2, 3)) > 1e-9:   # away from the k boundaries
            assert abs(f[0, 0] / 2 - want) < 1e-9
        vals.append(f[0, 0] / 2)
        assert abs(f[0, 1] - 0.0) < 1e-12                                # non-target logit = |x| cos = 0
    assert all(a >= b - 1e-9 for a, b in zip(vals, vals[1:]))
    assert abs(vals[0] - 1) < 1e-12 and abs(vals[-1] + 7) < 1e-9
    # lambda annealing (Appendix A.9)
    assert ops.asoftmax_lambda(0) == 1000.0
    assert abs(ops.asoftmax_lambda(100) - 1000.0 / 13.0) < 1e-12
    assert ops.asoftmax_lambda(10 ** 6) == 5.0


def test_center_loss_known_answer_and_duplicates():
    feats = np.array([[1., 2.], [3., 4.], [5., 6.]])
    labels = np.array([1, 1, 0])
    centers = np.array([[1., 1.], [0., 0.]])
    loss, df, newc = ops.center_loss(feats, labels, centers, alpha=0.5)
    want = ((feats - centers[labels]) ** 2).mean()
    assert abs(loss - want) < 1e-15
    np.testing.assert_allclose(df, 2 * (feats - centers[labels]) / 6)
    # scatter_sub accumulates duplicates, no count normalisation (loss.py:37-39)
    np.testing.assert_allclose(newc[1], 0 - 0.5 * ((0 - 1) + (0 - 3), (0 - 2) + (0 - 4))[0] * np.array([1, 0])
                               - 0.5 * np.array([0, (0 - 2) + (0 - 4)]))
    np.testing.assert_allclose(newc[0], np.array([1., 1.]) - 0.5 * (np.array([1., 1.]) - feats[2]))


@pytest.mark.parametrize('margin', [None, 0.3])
def test_triplet_matches_torch_restatement(margin):
    rng = np.random.default_rng(4)
    f = rng.standard_normal((8, 5)); y = np.array([0, 0, 1, 1, 2, 2, 2, 3])   # label 3 has no positive
    loss, df = ops.batch_hard_triplet(f, y, margin)
    tf_ = torch.tensor(f, requires_grad=True)
    diff = tf_[:, None, :] - tf_[None, :, :]
    dist = torch.sqrt((diff ** 2).sum(-1) + 1e-12)
    same = torch.tensor(y[:, None] == y[None, :])
    pos = (same ^ torch.eye(8, dtype=torch.bool)).double()
    neg = (~same).double()
    hp = (dist * pos).max(dim=1).values
    hn = (dist * neg + 1e6 * same.double()).min(dim=1).values
    tl = F.softplus(hp - hn) if margin is None else torch.clamp(hp - hn + margin, min=0)
    tl.sum().backward()
    np.testing.assert_allclose(loss, tl.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(df, tf_.grad.numpy(), atol=1e-10)


def test_focal_loss_matches_torch_autograd():
    """loss.py:18-27 restated in oracle.ops.focal_loss vs the literal formula under torch autograd (the gradient flows
    through BOTH the cross-entropy and the softmax-score factor)."""
    import torch
    rng = np.random.default_rng(0)
    z = rng.standard_normal((6, 11)) * 2
    y = rng.integers(0, 11, 6)
    for g, a in ((1.0, 2.0), (0.5, 1.0), (2.0, 3.5)):
        loss, d = ops.focal_loss(z, y, g, a)
        t = torch.tensor(z, requires_grad=True)
        ce = torch.nn.functional.cross_entropy(t, torch.tensor(y), reduction='none')
        sc = torch.softmax(t, 1)[torch.arange(6), torch.tensor(y)]
        ref = (g * (1 - sc) ** a * ce).mean()
        ref.backward()
        assert abs(loss - ref.item()) < 1e-12
        np.testing.assert_allclose(d, t.grad.numpy(), atol=1e-14)
