"""Oracle primitives: numpy restatements of the TF-1.x ops the reference graph uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function works in the
dtype of its inputs (float64 for parity checks, float32 for the timed
`cpu_baseline` leg of bench.py).  Layouts are the reference's boundary
layouts: activations NHWC, conv weights HWIO `[kh,kw,Cin,Cout]`, FC weights
`[in,out]`.  TF semantics follow SURVEY.md Appendix A.
"""
import contextlib

import numpy as np


# --------------------------------------------------------------------------
# MFMA operand precision (the build's bf16 mode, BASELINE.json configs[2]: "bf16").  The reference computes in fp32
# (dtype=tf.float32 throughout); the engine's bf16 mode rounds the two OPERANDS of every matrix product that runs on
# the MFMA kernel family (convolutions and dense products, forward and both gradients) to bfloat16 -- round to nearest
# even -- and keeps products, accumulation, storage and everything else in fp32.  `operand_rounding('bf16')` makes the
# oracle's products do exactly that, so the bf16 mode has an EXACT counterpart to be checked against instead of a loose
# mixed-precision tolerance.  Default: no rounding.
# --------------------------------------------------------------------------
_OPERAND_ROUND = None


def bf16_round(a):
    """Nearest-even rounding of the float32 value of `a` to bfloat16, returned in a's dtype."""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(a.dtype) if a.dtype != np.float32 else r


@contextlib.contextmanager
def operand_rounding(mode):
    """mode None | 'bf16': precision of the operands of conv2d_* / fc_* / mfma_matmul products inside the block."""
    global _OPERAND_ROUND
    assert mode in (None, 'bf16'), mode
    prev = _OPERAND_ROUND
    _OPERAND_ROUND = bf16_round if mode == 'bf16' else None
    try:
        yield
    finally:
        _OPERAND_ROUND = prev


def _r(a):
    return a if _OPERAND_ROUND is None else _OPERAND_ROUND(a)


# --------------------------------------------------------------------------
# bf16 STORAGE (the build's 'bf16s' mode; include/fte.h "bf16 STORAGE"): the activations the backward
# pass keeps (z, y) and the gradients that travel between layers (dz, the skip-path gradient) live in HBM as bfloat16.  Every
# such tensor is rounded ONCE, to nearest even, where it is written, and every consumer sees the rounded value; sums
# (dalpha, dbias, filter gradients) are formed from the unrounded fp32 terms.  `storage_rounding('bf16')` makes the nets'
# oracles (oracle/spherenet.py) round at exactly those points via stored().  Default: no rounding.
# --------------------------------------------------------------------------
_STORAGE_ROUND = None


@contextlib.contextmanager
def storage_rounding(mode):
    global _STORAGE_ROUND
    assert mode in (None, 'bf16'), mode
    prev = _STORAGE_ROUND
    _STORAGE_ROUND = bf16_round if mode == 'bf16' else None
    try:
        yield
    finally:
        _STORAGE_ROUND = prev


@contextlib.contextmanager
def no_rounding():
    """Both rounding modes off inside the block (the exact float64 evaluation beside a rounded one: the oracles' own bf16 noise
    figures, oracle/spherenet.py bf16_noise, oracle/graphnet.py noise_bands16)."""
    global _OPERAND_ROUND, _STORAGE_ROUND
    prev = (_OPERAND_ROUND, _STORAGE_ROUND)
    _OPERAND_ROUND = _STORAGE_ROUND = None
    try:
        yield
    finally:
        _OPERAND_ROUND, _STORAGE_ROUND = prev


def stored(a):
    """`a` as its consumers see it after a round trip through HBM in the active storage precision."""
    return a if _STORAGE_ROUND is None else _STORAGE_ROUND(a)


def storage_active():
    return _STORAGE_ROUND is not None


def rounding_active():
    """True inside operand_rounding('bf16')"""
    return _OPERAND_ROUND is not None


def mfma_matmul(a, b):
    """a @ b as the engine's MFMA kernels compute it: operands rounded per operand_rounding(), fp32/64 accumulate."""
    return _r(a) @ _r(b)


# --------------------------------------------------------------------------
# TF 'SAME' padding  (Appendix A.1; used implicitly by every layers.conv2d in
# nets/sphere.py:41-42,57,61,65,69 -- padding is never overridden there).
# --------------------------------------------------------------------------
def same_pads(in_size, k, stride):
    out = -(-in_size // stride)                      # ceil(in/stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def _pad_nhwc(x, pt, pb, pl, pr):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def _im2col(xp, kh, kw, stride, ho, wo):
    """[N,Hp,Wp,C] -> [N*ho*wo, kh*kw*C] with k ordered (r, s, c) == HWIO rows."""
    n, hp, wp, c = xp.shape
    s0, s1, s2, s3 = xp.strides
    view = np.lib.stride_tricks.as_strided(
        xp, shape=(n, ho, wo, kh, kw, c),
        strides=(s0, s1 * stride, s2 * stride, s1, s2, s3), writeable=False)
    return np.ascontiguousarray(view).reshape(n * ho * wo, kh * kw * c)


def conv2d_fwd(x, w, stride=1, bias=None):
    """layers.conv2d(padding='SAME') linear part: NHWC x HWIO -> NHWC (nets/sphere.py:41,57)."""
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c
    ho, pt, pb = same_pads(h, kh, stride)
    wo, pl, pr = same_pads(wd, kw, stride)
    cols = _im2col(_pad_nhwc(x, pt, pb, pl, pr), kh, kw, stride, ho, wo)
    z = mfma_matmul(cols, w.reshape(kh * kw * ci, co))
    if bias is not None:
        z = z + bias
    return z.reshape(n, ho, wo, co)


def conv2d_bwd(x, w, dz, stride=1, need_dx=True, need_dw=True):
    """Gradients of conv2d_fwd wrt x and w (what tf.gradients lowers to:
    Conv2DBackpropInput / Conv2DBackpropFilter; data_parallel.py:33).  `need_dw=False` skips the filter gradient
    (x then only gives the shape)."""
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    ho, pt, pb = same_pads(h, kh, stride)
    wo, pl, pr = same_pads(wd, kw, stride)
    xp = _pad_nhwc(x, pt, pb, pl, pr)
    dz2 = dz.reshape(n * ho * wo, co)
    dw = None
    if need_dw:
        cols = _im2col(xp, kh, kw, stride, ho, wo)
        dw = mfma_matmul(cols.T, dz2).reshape(kh, kw, ci, co)
    dx = None
    if need_dx:
        dcols = mfma_matmul(dz2, w.reshape(kh * kw * ci, co).T).reshape(n, ho, wo, kh, kw, ci)
        dxp = np.zeros_like(xp)
        for r in range(kh):
            for s in range(kw):
                dxp[:, r:r + stride * ho:stride, s:s + stride * wo:stride, :] += dcols[:, :, :, r, s, :]
        dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    return dx, dw


# --------------------------------------------------------------------------
# PReLU  (nets/sphere.py:29-36): relu(x) + alpha*(x-|x|)*0.5, alpha per channel.
# --------------------------------------------------------------------------
def prelu_fwd(z, alpha):
    return np.maximum(z, 0) + alpha * (z - np.abs(z)) * z.dtype.type(0.5)


def prelu_bwd(z, alpha, dy, zsign=None):
    """Gradient of prelu_fwd.  TF: d relu(0) = 0 and d|x|(0) = sign(0) = 0, so the slope at exactly
    z == 0 is alpha/2.  `zsign` (optional, same shape) replaces z in the three-way branch only: the
    derivative is discontinuous at 0 and an fp32 evaluation of z may land on the other side of it
    for |z| below fp32 resolution (see spherenet.kink_resolved)."""
    zs = z if zsign is None else zsign
    one = z.dtype.type(1)
    slope = np.where(zs > 0, one, alpha * np.ones_like(z))
    slope = np.where(zs == 0, alpha * z.dtype.type(0.5) * np.ones_like(z), slope)
    dz = dy * slope
    axes = tuple(range(z.ndim - 1))
    dalpha = (dy * np.where(zs <= 0, z, 0)).sum(axis=axes)
    return dz, dalpha


# --------------------------------------------------------------------------
# fully_connected (nets/sphere.py:73-74, 86-90)
# --------------------------------------------------------------------------
def fc_fwd(x, w, b=None):
    y = mfma_matmul(x, w)
    return y if b is None else y + b


def fc_bwd(x, w, dy, has_bias):
    dx = mfma_matmul(dy, w.T)
    dw = mfma_matmul(x.T, dy)
    db = dy.sum(axis=0) if has_bias else None
    return dx, dw, db


# --------------------------------------------------------------------------
# tf.losses.sparse_softmax_cross_entropy: mean over examples (nets/sphere.py:109)
# --------------------------------------------------------------------------
def softmax_ce(logits, labels, grad_scale=None):
    """Returns (loss, dlogits) with dlogits = (softmax - onehot) * grad_scale,
    grad_scale defaulting to 1/N (mean reduction)."""
    n = logits.shape[0]
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    s = e.sum(axis=1, keepdims=True)
    logp_y = (logits - m)[np.arange(n), labels] - np.log(s[:, 0])
    loss = -logp_y.mean()
    d = e / s
    d[np.arange(n), labels] -= 1
    d *= logits.dtype.type(1.0 / n if grad_scale is None else grad_scale)
    return loss, d


def focal_loss(logits, labels, gamma=1.0, alpha=2.0, grad_scale=None):
    """loss.py:18-27: mean_i gamma * (1 - p_y)^alpha * CE_i (the reference's parameter names kept as written).
    Returns (loss, dlogits) with dlogits scaled by grad_scale (default 1/N: the reduce_mean)."""
    n = logits.shape[0]
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    s = e.sum(axis=1, keepdims=True)
    p = e / s
    logq = (logits - m)[np.arange(n), labels] - np.log(s[:, 0])
    q = np.exp(logq)
    per = gamma * (1 - q) ** alpha * (-logq)
    coef = gamma * ((1 - q) ** alpha - alpha * q * (1 - q) ** (alpha - 1) * logq)
    d = p.copy()
    d[np.arange(n), labels] -= 1
    d *= coef[:, None] * (1.0 / n if grad_scale is None else grad_scale)
    return per.mean(), d


# --------------------------------------------------------------------------
# l2_regularizer(s)(w) = s * sum(w^2)/2   (Appendix A.4, nets/net_base.py:105)
# --------------------------------------------------------------------------
def l2_reg(ws, wd):
    return wd * sum(float((w.astype(np.float64) ** 2).sum()) for w in ws) / 2


# --------------------------------------------------------------------------
# Optimizers (data_parallel.py:65-69,191-196; Appendix A.7)
# --------------------------------------------------------------------------
def momentum_step(w, acc, g, lr, mom=0.9):
    acc = mom * acc + g
    return w - lr * acc, acc


def adam_step(w, m, v, g, lr, t, b1=0.5, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


# --------------------------------------------------------------------------
# LR schedules (train.py:122-144).  step is the global step (0-based).
# --------------------------------------------------------------------------
def lr_step(step, init_lr, decay_rate, decay_epochs, batches_per_epoch):
    """tf.train.piecewise_constant: value[i] for boundaries[i-1] < step <= boundaries[i]
    (the boundary step itself still takes the EARLIER value)."""
    bounds = [(int(e) - 1) * batches_per_epoch for e in decay_epochs]
    vals = [init_lr] + [init_lr * decay_rate ** (p + 1) for p in range(len(bounds))]
    k = 0
    while k < len(bounds) and step > bounds[k]:
        k += 1
    return vals[k]


def lr_exp(step, init_lr, decay_epoch, max_epoches, batches_per_epoch):
    decay_step = int(decay_epoch) * batches_per_epoch
    if step < decay_step:
        return init_lr
    decay_steps = int(max_epoches) * batches_per_epoch + 1 - decay_step
    return init_lr * 0.001 ** ((step - decay_step) / decay_steps)


def lr_cosine(step, init_lr, max_epoches, batches_per_epoch):
    total = max_epoches * batches_per_epoch
    s = min(step, total)
    return init_lr * 0.5 * (1 + np.cos(np.pi * s / total))


# --------------------------------------------------------------------------
# A-softmax (SphereFace, m=4; SURVEY.md Appendix A.9 -- NOT in the reference tree)
# --------------------------------------------------------------------------
def asoftmax_lambda(it, lambda_base=1000.0, gamma=0.12, power=1.0, lambda_min=5.0):
    return max(lambda_min, lambda_base * (1.0 + gamma * it) ** (-power))


_COS_K = (np.cos(np.pi / 4), 0.0, np.cos(3 * np.pi / 4))


def asoftmax_logits(x, w, labels, lam):
    """x [N,D], w [D,C] -> margin logits f [N,C] plus the intermediates backward needs."""
    n = x.shape[0]
    xn = np.sqrt((x * x).sum(axis=1))                 # |x_i|
    wn = np.sqrt((w * w).sum(axis=0))                 # |W_j|
    s = mfma_matmul(x, w)
    f = s / wn                                        # |x| cos(theta_ij)
    sy = s[np.arange(n), labels]
    c = sy / (xn * wn[labels])
    k = (c <= _COS_K[0]).astype(np.int64) + (c <= _COS_K[1]) + (c <= _COS_K[2])
    sign = np.where(k % 2 == 0, 1.0, -1.0).astype(x.dtype)
    c2 = c * c
    psi = sign * (8 * c2 * c2 - 8 * c2 + 1) - 2 * k
    dpsi = sign * (32 * c2 * c - 16 * c)
    phi = lam * c + psi
    f = f.copy()
    f[np.arange(n), labels] = xn * phi / (1 + lam)
    return f, dict(xn=xn, wn=wn, s=s, c=c, phi=phi, dphi=lam + dpsi)


def asoftmax_fwd_bwd(x, w, labels, lam, grad_scale=None):
    """Mean softmax-CE over the margin logits and its EXACT gradient wrt x and w
    (differentiating through both norms)."""
    n = x.shape[0]
    idx = np.arange(n)
    f, t = asoftmax_logits(x, w, labels, lam)
    loss, g = softmax_ce(f, labels, grad_scale)
    xn, wn, s, c, phi, dphi = t['xn'], t['wn'], t['s'], t['c'], t['phi'], t['dphi']
    G = g / wn                                        # coefficient of ds_ij, j != y
    G[idx, labels] = g[idx, labels] * dphi / ((1 + lam) * wn[labels])
    rowcoef = g[idx, labels] * (phi - dphi * c) / ((1 + lam) * xn)
    colcoef = -(G * s).sum(axis=0) / (wn * wn)
    dx = mfma_matmul(G, w.T) + rowcoef[:, None] * x
    dw = mfma_matmul(x.T, G) + colcoef[None, :] * w
    return loss, f, dx, dw


# --------------------------------------------------------------------------
# center loss (loss.py:29-45) and batch-hard triplet (loss.py:47-78)
# --------------------------------------------------------------------------
def center_loss(features, labels, centers, alpha=0.99):
    """Returns (loss, dfeatures, new_centers).  Gather happens BEFORE the
    scatter_sub (the loss uses the pre-update centers: loss.py:37,41)."""
    cb = centers[labels]
    diffs = (1 - alpha) * (cb - features)
    new_centers = centers.copy()
    np.subtract.at(new_centers, labels, diffs)        # duplicates accumulate (scatter_sub)
    d = features - cb
    loss = (d * d).mean()
    dfeat = 2 * d / d.size
    return loss, dfeat, new_centers


def batch_hard_triplet(features, labels, margin=None):
    """Per-sample loss vector [N] (unreduced, loss.py:78) and d(sum)/dfeatures."""
    n = features.shape[0]
    diff = features[:, None, :] - features[None, :, :]
    d2 = (diff * diff).sum(-1)
    dist = np.sqrt(d2 + 1e-12)
    same = labels[:, None] == labels[None, :]
    pos_mask = (same ^ np.eye(n, dtype=bool)).astype(features.dtype)
    neg_mask = (~same).astype(features.dtype)
    pm = dist * pos_mask
    nm = dist * neg_mask + 1e6 * same.astype(features.dtype)
    ip = pm.argmax(axis=1)
    ineg = nm.argmin(axis=1)
    hp = pm[np.arange(n), ip]
    hn = nm[np.arange(n), ineg]
    v = hp - hn
    if margin is None:
        loss = np.logaddexp(0, v)
        dv = 1 / (1 + np.exp(-v))
    else:
        loss = np.maximum(0, v + margin)
        dv = (v + margin > 0).astype(features.dtype)
    # gradient of sum(loss) wrt features (d dist_ij / d f_i = (f_i - f_j)/dist_ij)
    df = np.zeros_like(features)
    for i in range(n):
        if pos_mask[i, ip[i]] > 0:                    # hardest_pos is a real distance, not a masked 0
            gvec = dv[i] * diff[i, ip[i]] / dist[i, ip[i]]
            df[i] += gvec
            df[ip[i]] -= gvec
        if not same[i, ineg[i]]:                      # hardest_neg is a real distance, not the 1e6 filler
            gvec = dv[i] * diff[i, ineg[i]] / dist[i, ineg[i]]
            df[i] -= gvec
            df[ineg[i]] += gvec
    return loss, df


# --------------------------------------------------------------------------
# layers.batch_norm(fused=True, scale=True, center=True, decay=0.999, epsilon=1e-3)
# (nets/resnet.py:97-99; Appendix A.6).  Statistics over N,H,W of THIS shard.
# --------------------------------------------------------------------------
BN_EPS = 1e-3
BN_DECAY = 0.999


def bn_train_fwd(x, gamma, beta, eps=BN_EPS):
    axes = tuple(range(x.ndim - 1))
    mean = x.mean(axis=axes)
    var = x.var(axis=axes)                       # biased: what normalises the batch
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return gamma * xhat + beta, dict(xhat=xhat, rstd=rstd, mean=mean, var=var)


def bn_moving_update(moving_mean, moving_var, mean, var, count, decay=BN_DECAY):
    """The fused kernel feeds the UNBIASED variance (N/(N-1)) into the moving average."""
    unbiased = var * (count / max(count - 1.0, 1.0))
    return decay * moving_mean + (1 - decay) * mean, decay * moving_var + (1 - decay) * unbiased


def bn_train_bwd(dy, gamma, cache):
    xhat, rstd = cache['xhat'], cache['rstd']
    axes = tuple(range(dy.ndim - 1))
    m = float(np.prod([dy.shape[a] for a in axes]))
    dbeta = dy.sum(axis=axes)
    dgamma = (dy * xhat).sum(axis=axes)
    dx = gamma * rstd * (dy - dbeta / m - xhat * (dgamma / m))
    return dx, dgamma, dbeta


def bn_infer(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    return gamma * (x - moving_mean) / np.sqrt(moving_var + eps) + beta


# --------------------------------------------------------------------------
# max_pool2d(kernel 3, stride 2, padding='SAME')  (nets/resnet.py:115); padded cells never win.
# --------------------------------------------------------------------------
def maxpool3x3s2_fwd(x):
    n, h, w, c = x.shape
    ho, pt, pb = same_pads(h, 3, 2)
    wo, pl, pr = same_pads(w, 3, 2)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    s0, s1, s2, s3 = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, shape=(n, ho, wo, 3, 3, c), strides=(s0, 2 * s1, 2 * s2, s1, s2, s3), writeable=False)
    flat = win.reshape(n, ho, wo, 9, c)
    arg = flat.argmax(axis=3)                    # first maximum wins (row-major window order), like TF's MaxPoolGrad
    y = np.take_along_axis(flat, arg[:, :, :, None, :], axis=3)[:, :, :, 0, :]
    return y, dict(arg=arg, shape=x.shape, pads=(pt, pl))


def maxpool3x3s2_bwd(dy, cache):
    n, h, w, c = cache['shape']
    pt, pl = cache['pads']
    arg = cache['arg']
    _, ho, wo, _ = dy.shape
    dx = np.zeros((n, h + 3, w + 3, c), dy.dtype)
    oh = np.arange(ho)[None, :, None, None]
    ow = np.arange(wo)[None, None, :, None]
    ih = oh * 2 + arg // 3                       # coordinates in the padded frame
    iw = ow * 2 + arg % 3
    ni = np.arange(n)[:, None, None, None]
    ci = np.arange(c)[None, None, None, :]
    np.add.at(dx, (ni, ih, iw, ci), dy)
    return dx[:, pt:pt + h, pl:pl + w, :]


def gap_fwd(x):
    return x.mean(axis=(1, 2))                   # tf.reduce_mean over the spatial axes (nets/resnet.py:142)


def gap_bwd(dy, shape):
    n, h, w, c = shape
    return np.broadcast_to(dy[:, None, None, :] / (h * w), shape).copy()


def dropout_fwd(x, mask, keep_prob=0.5):
    """layers.dropout: inverted dropout; `mask` is the 0/1 keep mask (TF's RNG stream is not reproducible,
    so parity tests feed the implementation's mask to the oracle)."""
    return x * mask / keep_prob


# ------------------------------------------------------------------------------------------------
# ShuffleNet-v2 pieces (nets/shufflenet_v2.py)
# ------------------------------------------------------------------------------------------------
def dwconv3x3_fwd(x, w, stride=1):
    """Depthwise half of layers.separable_conv2d (nets/shufflenet_v2.py:98,104; depth_multiplier=1, :139).
    x [N,H,W,C], w [3,3,C,1], TF-SAME."""
    n, h, wd, c = x.shape
    ho, pt, pb = same_pads(h, 3, stride)
    wo, pl, pr = same_pads(wd, 3, stride)
    xp = _pad_nhwc(x, pt, pb, pl, pr)
    y = np.zeros((n, ho, wo, c), x.dtype)
    for r in range(3):
        for q in range(3):
            y += xp[:, r:r + (ho - 1) * stride + 1:stride, q:q + (wo - 1) * stride + 1:stride, :] * w[r, q, :, 0]
    return y


def dwconv3x3_bwd(x, w, dy, stride=1):
    n, h, wd, c = x.shape
    ho, pt, pb = same_pads(h, 3, stride)
    wo, pl, pr = same_pads(wd, 3, stride)
    xp = _pad_nhwc(x, pt, pb, pl, pr)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for r in range(3):
        for q in range(3):
            sl = (slice(None), slice(r, r + (ho - 1) * stride + 1, stride), slice(q, q + (wo - 1) * stride + 1, stride), slice(None))
            dw[r, q, :, 0] = (xp[sl] * dy).sum(axis=(0, 1, 2))
            dxp[sl] += dy * w[r, q, :, 0]
    return dxp[:, pt:pt + h, pl:pl + wd, :], dw


def channel_split(x):
    """_channel_split (nets/shufflenet_v2.py:60-64): sizes [int(0.5*C), C - int(0.5*C)] on the channel axis."""
    c = x.shape[-1]
    h = int(0.5 * c)
    return x[..., :h], x[..., h:]


def channel_shuffle(x, data_format='NCHW'):
    """_channel_shuffle (nets/shufflenet_v2.py:66-77) restated on an NHWC array.  The two layouts of the reference
    do NOT compute the same permutation: NCHW (the default, :37) views channels as [2, C/2] and transposes --
    out[2i+j] = in[j*C/2+i]; its NHWC branch views them as [C/2, 2] -- out[j*C/2+i] = in[2i+j]."""
    c = x.shape[-1]
    lead = x.shape[:-1]
    if data_format == 'NCHW':
        return x.reshape(lead + (2, c // 2)).swapaxes(-1, -2).reshape(lead + (c,))
    return x.reshape(lead + (c // 2, 2)).swapaxes(-1, -2).reshape(lead + (c,))
