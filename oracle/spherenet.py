"""Oracle: SphereNet-20 ("SphereFaceNet-20") forward / loss / backward / one training step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED.
Follows nets/sphere.py:29-36 (PReLU), :38-45 (resBlock), :47-76 (backbone),
:78-101 (forward, train and flip-averaged eval), :103-118 (loss),
nets/net_base.py:103-107 (reg_loss), data_parallel.py:32-43 (grad scaling),
:45-79 (Singular step), :203-256 + :175-200 (shard / all_sum / per-replica
optimizer).  Parameters use the reference's TF variable names and layouts
(SURVEY.md Appendix D): conv `weights` HWIO, FC `weights` [in,out] with the
flatten order of the chosen data_format (nets/sphere.py:72).
"""
from collections import OrderedDict

import numpy as np

from . import ops

NUM_OUTPUTS = (64, 128, 256, 512)      # nets/sphere.py:26
NUM_BLOCKS = (1, 2, 4, 1)              # nets/sphere.py:58,62,66,70
EMBED = 512                            # nets/sphere.py:73


def conv_layer_names():
    """Conv scopes in forward order with (name, stride, has_bias, is_block_second)."""
    out = []
    for si, nb in enumerate(NUM_BLOCKS):
        stage = 'SphereNet/conv%d' % (si + 1)
        out.append((stage + '/Conv', si, 2, True, None))
        for b in range(nb):
            if nb == 1:
                blk = stage + '/resBlock'                       # self.resBlock(...) direct call
            else:
                blk = stage + '/Repeat/resBlock_%d' % (b + 1)   # layers.repeat scopes
            out.append((blk + '/Conv', si, 1, False, 0))
            out.append((blk + '/Conv_1', si, 1, False, 1))
    return out


def feature_hw(h, w):
    for _ in range(4):
        h, _, _ = ops.same_pads(h, 3, 2)
        w, _, _ = ops.same_pads(w, 3, 2)
    return h, w


def init_params(seed, in_ch, num_classes, height=112, width=112, dtype=np.float64):
    """Reference initialisers (nets/sphere.py:34,41-42,87; Appendix A.2/A.3)."""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    cin = in_ch
    for name, si, stride, has_bias, _ in conv_layer_names():
        cout = NUM_OUTPUTS[si]
        if has_bias:      # stage-entry conv: layers.conv2d default Xavier-uniform, zero bias
            lim = np.sqrt(6.0 / (9 * cin + 9 * cout))
            p[name + '/weights'] = rng.uniform(-lim, lim, (3, 3, cin, cout)).astype(dtype)
            p[name + '/biases'] = np.zeros(cout, dtype)
        else:             # resBlock conv: N(0, 0.01), no bias
            p[name + '/weights'] = (0.01 * rng.standard_normal((3, 3, cin, cout))).astype(dtype)
        p[name + '/alpha'] = np.full(cout, 0.25, dtype)
        cin = cout
    fh, fw = feature_hw(height, width)
    fin = fh * fw * NUM_OUTPUTS[3]
    lim = np.sqrt(6.0 / (fin + EMBED))
    p['SphereNet/fully_connected/weights'] = rng.uniform(-lim, lim, (fin, EMBED)).astype(dtype)
    p['SphereNet/fully_connected/biases'] = np.zeros(EMBED, dtype)
    p['classifier/fc_classifier/weights'] = (0.001 * rng.standard_normal((EMBED, num_classes))).astype(dtype)
    return p


def perturb_params(p, seed, scale=0.05):
    """Test helper: move biases/alphas off their constant initial values so that
    bias / alpha gradient paths are exercised with non-trivial numbers."""
    rng = np.random.default_rng(seed)
    q = OrderedDict()
    for k, v in p.items():
        if k.endswith('/biases') or k.endswith('/alpha'):
            q[k] = (v + scale * rng.standard_normal(v.shape)).astype(v.dtype)
        else:
            q[k] = v
    return q


def regularized_names(p):
    return [k for k in p if k.endswith('/weights')]          # conv + fc weights only (A.4)


def _flatten(y, data_format):
    n = y.shape[0]
    if data_format == 'NCHW':                                 # nets/sphere.py:53-54,72
        return y.transpose(0, 3, 1, 2).reshape(n, -1)
    return y.reshape(n, -1)


def _unflatten(d, shape, data_format):
    n, h, w, c = shape
    if data_format == 'NCHW':
        return d.reshape(n, c, h, w).transpose(0, 2, 3, 1)
    return d.reshape(n, h, w, c)


def backbone_fwd(p, images, data_format='NCHW', keep=True):
    """images NHWC in [-1,1] -> embedding [N,512]; cache holds what backward needs.
    Precision modes of the engine (ops.operand_rounding / ops.storage_rounding): the FIRST conv (K = 9*Cin <= 27) is not on the
    bf16 MFMA path in any mode -- its operands are never rounded; under bf16 storage every layer's z and block output are
    rounded where they are stored, except the LAST conv layer's (the dense layer reads them from an fp32 buffer)."""
    cache = []
    x = images
    shortcut = None
    specs = conv_layer_names()
    for li, (name, si, stride, has_bias, second) in enumerate(specs):
        if second == 0:
            shortcut = x
        if li == 0:
            with ops.operand_rounding(None):
                z = ops.conv2d_fwd(x, p[name + '/weights'], stride, p.get(name + '/biases'))
        else:
            z = ops.conv2d_fwd(x, p[name + '/weights'], stride, p.get(name + '/biases'))
        y = ops.prelu_fwd(z, p[name + '/alpha'])
        out = y + shortcut if second == 1 else y
        last = li == len(specs) - 1
        if keep:
            cache.append((name, x, z if last else ops.stored(z)))
        x = out if last else ops.stored(out)
    feat_shape = x.shape
    flat = _flatten(x, data_format)
    emb = ops.fc_fwd(flat, p['SphereNet/fully_connected/weights'], p['SphereNet/fully_connected/biases'])
    return emb, dict(layers=cache, feat_shape=feat_shape, flat=flat, data_format=data_format)


KINK_BAND = 1e-5      # |z| < KINK_BAND * rms(z): fp32 cannot tell which side of PReLU's kink z is on
BF16_KINK_MULT = 4    # mixed-precision checks: the band is this many times the ORACLE's own bf16 noise on the tensor (bf16_noise)


def bf16_noise(p, images, data_format='NCHW'):
    """name -> rms(z evaluated with bf16 operand rounding (and the ambient storage rounding) - z evaluated exactly) of every conv
    layer, on THIS input, from the oracle alone: how far bf16 operands (unit roundoff 2^-9 per operand, ~1.6e-3 * rms per layer,
    compounding with depth) move a pre-activation.  The mixed-precision kink band is a multiple of it -- never a function of the
    tensors of the implementation under test (a kernel bug that perturbs z cannot widen its own acceptance band)."""
    with ops.operand_rounding('bf16'):
        _, ca = backbone_fwd(p, images, data_format)
    with ops.no_rounding():
        _, cb = backbone_fwd(p, images, data_format)
    return {a[0]: float(np.sqrt(((a[2] - b[2]) ** 2).mean())) for a, b in zip(ca['layers'], cb['layers'])}


def kink_resolved(z, z_other, mode='fp32', noise=None):
    """PReLU's derivative jumps at z = 0.  Where the float64 z lies within the band of the
    kink, either one-sided slope is a valid answer for a lower-precision evaluation, so the oracle adopts
    the side the checked implementation took (`z_other`, its own z) -- there and only there.

    The band never depends on the implementation under test:
    mode 'fp32': the FIXED KINK_BAND*rms(z);
    mode 'bf16' (the mixed-precision checks only): max(that, BF16_KINK_MULT * noise) with `noise` = bf16_noise()'s figure for
    this layer -- the oracle's own rounded-vs-exact discrepancy on this input; the forward tensors are held to their own
    (stated, looser) tolerance by the same tests."""
    thr = KINK_BAND * np.sqrt((z * z).mean())
    if mode == 'bf16':
        assert noise is not None, "the bf16 kink band needs the oracle's own noise figure (bf16_noise)"
        thr = max(thr, BF16_KINK_MULT * noise)
    elif mode != 'fp32':
        raise ValueError(mode)
    return np.where(np.abs(z) < thr, z_other.astype(z.dtype), z)


def backbone_bwd(p, cache, demb, trace=None, kink=None, kink_mode='fp32', noise=None):
    """`trace`, when a dict, receives the per-layer gradient wrt the pre-activation (name -> dz).
    `kink`, when a dict name -> z of the implementation under test, resolves kink-band elements
    (`kink_mode`: see kink_resolved)."""
    g = OrderedDict()
    dflat, g['SphereNet/fully_connected/weights'], g['SphereNet/fully_connected/biases'] = ops.fc_bwd(
        cache['flat'], p['SphereNet/fully_connected/weights'], demb, True)
    dx = _unflatten(dflat, cache['feat_shape'], cache['data_format'])
    specs = conv_layer_names()
    dskip = None
    for li in range(len(specs) - 1, -1, -1):
        name, si, stride, has_bias, second = specs[li]
        _, x, z = cache['layers'][li]
        if second == 1:
            dskip = ops.stored(dx)                       # out = shortcut + prelu(z2): the skip-path gradient waits in HBM
        zs = kink_resolved(z, kink[name], kink_mode, None if noise is None else noise[name]) if kink is not None and name in kink else None
        dz, g[name + '/alpha'] = ops.prelu_bwd(z, p[name + '/alpha'], dx, zs)
        if trace is not None:
            trace[name] = dz
        if has_bias:
            g[name + '/biases'] = dz.sum(axis=(0, 1, 2))
        dz = ops.stored(dz)                              # what the data / filter gradient kernels read
        if li == 0:
            with ops.operand_rounding(None):             # the first conv's filter gradient is an fp32 product in every mode
                dxl, g[name + '/weights'] = ops.conv2d_bwd(x, p[name + '/weights'], dz, stride, need_dx=False)
        else:
            dxl, g[name + '/weights'] = ops.conv2d_bwd(x, p[name + '/weights'], dz, stride, need_dx=True)
        if second == 0:
            dxl = dxl + dskip
        dx = dxl
    return g


def eval_features(p, images, data_format='NCHW'):
    """nets/sphere.py:97-101: mean of the embedding of x and of its horizontal flip."""
    f1, _ = backbone_fwd(p, images, data_format, keep=False)
    f2, _ = backbone_fwd(p, images[:, :, ::-1, :], data_format, keep=False)
    return (f1 + f2) / 2


def loss_and_grads(p, images, labels, weight_decay=5e-4, data_format='NCHW',
                   head='softmax', lam=None, grad_scale=None, trace=None, kink=None, kink_mode='fp32'):
    """One tower of data_parallel.py:45-63 / :215-236.

    Returns (losses=[ce, reg], grads incl. the L2 term, extras).  `grad_scale`
    is the factor on d(ce)/d(logits) rows (default 1/N: shard mean); the
    reg-loss gradient wd*w is always added unscaled by it.
    """
    emb, cache = backbone_fwd(p, images, data_format)
    wc = p['classifier/fc_classifier/weights']
    if head == 'softmax':
        logits = ops.fc_fwd(emb, wc)
        ce, dlogits = ops.softmax_ce(logits, labels, grad_scale)
        demb, dwc, _ = ops.fc_bwd(emb, wc, dlogits, False)
    elif head == 'asoftmax':
        ce, logits, demb, dwc = ops.asoftmax_fwd_bwd(emb, wc, labels, lam, grad_scale)
    else:
        raise ValueError(head)
    noise = bf16_noise(p, images, data_format) if (kink is not None and kink_mode == 'bf16') else None
    g = backbone_bwd(p, cache, demb, trace, kink, kink_mode, noise)
    g['classifier/fc_classifier/weights'] = dwc
    reg_names = regularized_names(p)
    reg = ops.l2_reg([p[k] for k in reg_names], weight_decay)
    for k in reg_names:
        g[k] = g[k] + weight_decay * p[k]
    return [ce, reg], g, dict(embedding=emb, logits=logits)


def train_step(p, slots, images, labels, lr, num_towers=1, weight_decay=5e-4,
               data_format='NCHW', head='softmax', lam=None, optimizer='Momentum', t=1, kink=None, kink_mode='fp32'):
    """One global step as data_parallel.py builds it: split the batch into
    `num_towers` equal shards (:206-207), per-tower loss+grads scaled by
    1/num_towers (:37), sum over towers (:179), same update on every replica
    (:186-196).  Returns (new_params, new_slots, mean losses)."""
    n = images.shape[0]
    assert n % num_towers == 0
    sh = n // num_towers
    total = None
    losses = np.zeros(2)
    for r in range(num_towers):
        kr = None if kink is None else {k: v[r * sh:(r + 1) * sh] for k, v in kink.items()}
        ls, g, _ = loss_and_grads(p, images[r * sh:(r + 1) * sh], labels[r * sh:(r + 1) * sh],
                                  weight_decay, data_format, head, lam, kink=kr, kink_mode=kink_mode)
        losses += np.array(ls) / num_towers
        if total is None:
            total = OrderedDict((k, v / num_towers) for k, v in g.items())
        else:
            for k, v in g.items():
                total[k] = total[k] + v / num_towers
    newp, news = OrderedDict(), OrderedDict()
    for k in p:
        if optimizer == 'Momentum':
            newp[k], news[k] = ops.momentum_step(p[k], slots[k], total[k], lr)
        else:
            m, v = slots[k]
            w2, m2, v2 = ops.adam_step(p[k], m, v, total[k], lr, t)
            newp[k], news[k] = w2, (m2, v2)
    return newp, news, list(losses)


def zero_slots(p, optimizer='Momentum'):
    if optimizer == 'Momentum':
        return OrderedDict((k, np.zeros_like(v)) for k, v in p.items())
    return OrderedDict((k, (np.zeros_like(v), np.zeros_like(v))) for k, v in p.items())


def train_flops_per_image(in_ch, num_classes, height=112, width=112):
    """Algorithmic MACs*2 of fwd + dgrad + wgrad (SURVEY.md section 8d / BASELINE.md section 4)."""
    h, w, cin = height, width, in_ch
    fwd = 0
    first = None
    for name, si, stride, has_bias, _ in conv_layer_names():
        cout = NUM_OUTPUTS[si]
        h, _, _ = ops.same_pads(h, 3, stride)
        w, _, _ = ops.same_pads(w, 3, stride)
        mac = h * w * 9 * cin * cout
        if first is None:
            first = mac
        fwd += mac
        cin = cout
    fwd += h * w * cin * EMBED + EMBED * num_classes
    return 2 * (3 * fwd - first), 2 * fwd
