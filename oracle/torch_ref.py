"""Oracle, second implementation: the reference graph restated on torch-CPU (F.conv2d + autograd).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED.  Two uses:
  * tests/test_oracle.py pins the numpy oracle (oracle/spherenet.py, hand-written backward) against this
    file in float64: it is written in NCHW (the reference's default data_format, nets/sphere.py:53-54)
    straight from the reference's layer list and shares no code with the numpy oracle;
  * bench.py's `cpu_baseline` leg times it in float32 on the GPU box's host cores at BASELINE.json
    configs[0] (SphereFaceNet-20 + A-softmax, 112x112 gray, batch 64, Singular path) -- "the reference's CPU
    path" (TF-CPU cannot be installed here: SURVEY.md 8c/8d) -- and prints the embeddings / logits parity of the
    HIP path against it on the same inputs.
Follows nets/sphere.py:29-36 (PReLU, verbatim formula), :38-45 (resBlock), :47-76 (backbone), :78-95 (forward),
:103-118 (loss), nets/net_base.py:103-107 (reg_loss), data_parallel.py:45-79 (Singular step, Momentum 0.9),
SURVEY.md App. A.1 (TF SAME padding: the extra pixel goes to the bottom / right), A.9 (A-softmax, m = 4).
"""
import math

import torch
import torch.nn.functional as F


def tf_same_pad(x, k, stride):
    """x NCHW.  TF-1.x SAME: out = ceil(in/stride), pad_before = total // 2 (asymmetric for stride 2 on even sizes)."""
    h, w = x.shape[2], x.shape[3]

    def pads(n):
        out = (n + stride - 1) // stride
        tot = max((out - 1) * stride + k - n, 0)
        return tot // 2, tot - tot // 2
    pt, pb = pads(h)
    pl, pr = pads(w)
    return F.pad(x, (pl, pr, pt, pb))


def prelu(x, alpha):
    a = alpha.view(1, -1, 1, 1)
    return torch.relu(x) + a * (x - torch.abs(x)) * 0.5          # nets/sphere.py:36 verbatim formula


def conv3x3(x, w_hwio, stride, bias):
    w = w_hwio.permute(3, 2, 0, 1)                                # HWIO -> OIHW
    return F.conv2d(tf_same_pad(x, 3, stride), w, bias, stride=stride)


def backbone(tp, images_nhwc, data_format='NCHW'):
    """nets/sphere.py:47-76: images NHWC -> embedding [N,512]."""
    x = images_nhwc.permute(0, 3, 1, 2)                           # sphere.py:53-54

    def conv(name, x, stride):
        z = conv3x3(x, tp[name + '/weights'], stride, tp.get(name + '/biases'))
        return prelu(z, tp[name + '/alpha'])

    def block(scope, x):
        return x + conv(scope + '/Conv_1', conv(scope + '/Conv', x, 1), 1)
    x = conv('SphereNet/conv1/Conv', x, 2)
    x = block('SphereNet/conv1/resBlock', x)
    x = conv('SphereNet/conv2/Conv', x, 2)
    for i in (1, 2):
        x = block('SphereNet/conv2/Repeat/resBlock_%d' % i, x)
    x = conv('SphereNet/conv3/Conv', x, 2)
    for i in (1, 2, 3, 4):
        x = block('SphereNet/conv3/Repeat/resBlock_%d' % i, x)
    x = conv('SphereNet/conv4/Conv', x, 2)
    x = block('SphereNet/conv4/resBlock', x)
    if data_format == 'NHWC':
        x = x.permute(0, 2, 3, 1)
    flat = x.reshape(x.shape[0], -1)
    return flat @ tp['SphereNet/fully_connected/weights'] + tp['SphereNet/fully_connected/biases']


def asoftmax_logits(emb, wc, labels, lam):
    """SURVEY.md App. A.9 (SphereFace, m = 4): non-target logits |x| cos(theta_j); target |x| (lam cos + psi) / (1 + lam),
    psi = (-1)^k cos(4 theta) - 2k, k = #{j in 1..3 : cos <= cos(j pi / 4)}."""
    n = emb.shape[0]
    idx = torch.arange(n)
    xn = emb.norm(dim=1)
    wn = wc.norm(dim=0)
    s = emb @ wc
    f = s / wn
    c = s[idx, labels] / (xn * wn[labels])
    cd = c.detach()
    k = (cd <= math.cos(math.pi / 4)).to(c.dtype) + (cd <= 0.0).to(c.dtype) + (cd <= math.cos(3 * math.pi / 4)).to(c.dtype)
    sign = 1.0 - 2.0 * torch.remainder(k, 2.0)
    c2 = c * c
    psi = sign * (8 * c2 * c2 - 8 * c2 + 1) - 2 * k
    fy = xn * (lam * c + psi) / (1 + lam)
    return f.scatter(1, labels.view(-1, 1), fy.view(-1, 1))


def spherenet_loss(tp, images_nhwc, labels, wd, data_format='NCHW', head='softmax', lam=None):
    """One tower: returns (cross_entropy, reg_loss, embedding, logits)."""
    emb = backbone(tp, images_nhwc, data_format)
    wc = tp['classifier/fc_classifier/weights']
    if head == 'softmax':
        logits = emb @ wc                                         # nets/sphere.py:84-90, no bias
    elif head == 'asoftmax':
        logits = asoftmax_logits(emb, wc, labels, lam)
    else:
        raise ValueError(head)
    ce = F.cross_entropy(logits, labels)                          # tf.losses.sparse_softmax_cross_entropy: mean over the shard
    reg = sum(wd * (v ** 2).sum() / 2 for k, v in tp.items() if k.endswith('/weights'))      # nets/net_base.py:105
    return ce, reg, emb, logits


def to_torch(params, dtype=torch.float32, requires_grad=True):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}


def train_step(tp, slots, images_nhwc, labels, lr, wd=5e-4, data_format='NCHW', head='softmax', lam=None, momentum=0.9):
    """data_parallel.py:45-79 with MomentumOptimizer(lr, 0.9): acc <- 0.9 acc + g ; w <- w - lr acc (App. A.7).
    tp: name -> leaf tensor (requires_grad); slots: name -> tensor.  Updates both in place; returns (ce, reg, emb, logits)."""
    for v in tp.values():
        v.grad = None
    ce, reg, emb, logits = spherenet_loss(tp, images_nhwc, labels, wd, data_format, head, lam)
    (ce + reg).backward()
    with torch.no_grad():
        for k, v in tp.items():
            slots[k].mul_(momentum).add_(v.grad)
            v.sub_(lr * slots[k])
    return float(ce.detach()), float(reg.detach()), emb.detach(), logits.detach()
