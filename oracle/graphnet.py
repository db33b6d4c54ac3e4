"""Oracle: a tiny static-graph executor (numpy, float64) for the BN / pooling nets of the reference
(nets/resnet.py, nets/resnext.py, nets/shufflenet_v2.py).  TEST INFRASTRUCTURE -- PARITY UNPINNED.

A net is a list of ops over named tensors; forward fills an environment, backward walks the list in
reverse accumulating gradients.  The HIP engine (tf_face_toolbox_amd/nets/graph.py) executes the same
op lists, which is what makes the layer-by-layer parity checks line up."""
from collections import OrderedDict


import numpy as np

from . import ops


def forward(graph, params, images, labels=None, train=True, masks=None, state=None, stored=None):
    """graph: list of tuples.  Returns env (all tensors), caches.
    `stored` (bf16 storage, ops.storage_rounding): names of the tensors the engine keeps in HBM -- the net's `h16` set, i.e. the
    outputs of its FUSED ops (a BN + add + ReLU group stores only the ReLU's output) -- each rounded once, where it is computed,
    before any consumer reads it."""
    env = {'images': images}
    cache = {}
    new_state = {}
    prev_outs = ()
    for op in graph:
        for nm in prev_outs:
            if stored and nm in stored:
                env[nm] = ops.stored(env[nm])
        kind, out = op[0], op[1]
        prev_outs = (out, op[3]) if kind == 'split' else ((out, op[4]) if kind == 'shufsplit' else (out,))
        if kind == 'conv':          # ('conv', out, inp, wname, stride)
            _, _, inp, wname, stride = op
            env[out] = ops.conv2d_fwd(env[inp], params[wname], stride)
        elif kind == 'bn':          # ('bn', out, inp, prefix)
            _, _, inp, pre = op
            g, b = params[pre + '/gamma'], params[pre + '/beta']
            if train:
                env[out], cache[out] = ops.bn_train_fwd(env[inp], g, b)
                cnt = float(np.prod(env[inp].shape[:-1]))
                if state is not None:
                    new_state[pre + '/moving_mean'], new_state[pre + '/moving_variance'] = ops.bn_moving_update(
                        state[pre + '/moving_mean'], state[pre + '/moving_variance'], cache[out]['mean'], cache[out]['var'], cnt)
            else:
                env[out] = ops.bn_infer(env[inp], g, b, state[pre + '/moving_mean'], state[pre + '/moving_variance'])
        elif kind == 'relu':
            env[out] = np.maximum(env[op[2]], 0)
        elif kind == 'add':
            env[out] = env[op[2]] + env[op[3]]
        elif kind == 'maxpool':
            env[out], cache[out] = ops.maxpool3x3s2_fwd(env[op[2]])
        elif kind == 'gap':
            env[out] = ops.gap_fwd(env[op[2]])
        elif kind == 'dropout':     # ('dropout', out, inp, keep_prob)
            if train:
                env[out] = ops.dropout_fwd(env[op[2]], masks[out], op[3])
            else:
                env[out] = env[op[2]]
        elif kind == 'fc':          # ('fc', out, inp, wname, bname-or-None)
            env[out] = ops.fc_fwd(env[op[2]], params[op[3]], params[op[4]] if op[4] else None)
        elif kind == 'gconv':       # ('gconv', out, inp, wname, stride, groups): nets/resnext.py:41-51 split / conv / concat
            _, _, inp, wname, stride, groups = op
            x = env[inp]
            gw = x.shape[-1] // groups
            # bf16 mode: the engine's grouped 3x3 runs on the bf16 MFMA (operands rounded like every other MFMA product)
            env[out] = np.concatenate([ops.conv2d_fwd(x[..., g * gw:(g + 1) * gw], params[wname][g], stride)
                                       for g in range(groups)], axis=-1)
        elif kind == 'se':          # ('se', out, inp, prefix[, scope1, scope2]): nets/shufflenet_v2.py:79-85
            inp, pre = op[2], op[3]
            s1, s2 = (op[4], op[5]) if len(op) > 4 else ('fc1', 'fc2')
            x = env[inp]
            sq = x.mean(axis=(1, 2))
            w1 = params[pre + '/%s/weights' % s1].reshape(params[pre + '/%s/weights' % s1].shape[-2:])
            w2 = params[pre + '/%s/weights' % s2].reshape(params[pre + '/%s/weights' % s2].shape[-2:])
            hid = np.maximum(ops.mfma_matmul(sq, w1) + params[pre + '/%s/biases' % s1], 0)
            gate = 1.0 / (1.0 + np.exp(-(ops.mfma_matmul(hid, w2) + params[pre + '/%s/biases' % s2])))
            env[out] = x * gate[:, None, None, :]
            cache[out] = dict(sq=sq, hid=hid, gate=gate)
        elif kind == 'dwconv':      # ('dwconv', out, inp, wname, stride)
            env[out] = ops.dwconv3x3_fwd(env[op[2]], params[op[3]], op[4])
        elif kind == 'split':       # ('split', out_a, inp, out_b): nets/shufflenet_v2.py:60-64
            env[out], env[op[3]] = ops.channel_split(env[op[2]])
        elif kind == 'shufsplit':   # ('shufsplit', out_s, a, b, out_x, fmt): concat + shuffle (:112-113), then the NEXT block's split (:93)
            env[out], env[op[4]] = ops.channel_split(ops.channel_shuffle(np.concatenate([env[op[2]], env[op[3]]], axis=-1), op[5]))
        elif kind == 'shufcat':     # ('shufcat', out, a, b, fmt): concat + shuffle feeding a conv (the last block)
            env[out] = ops.channel_shuffle(np.concatenate([env[op[2]], env[op[3]]], axis=-1), op[4])
        else:
            raise ValueError(kind)
    for nm in prev_outs:
        if stored and nm in stored:
            env[nm] = ops.stored(env[nm])
    return env, cache, new_state


KINK_BAND = 1e-5


NOISE_MULT = 16      # band = NOISE_MULT x the oracle's OWN float32-vs-float64 forward noise on that tensor (noise_bands)


def noise_bands(graph, params, images, masks=None, state=None):
    """rms(float32 oracle - float64 oracle) of every ReLU / max-pool output on THIS input: the float32 noise floor of
    the decision tensors, computed from the oracle alone (batch norm over few samples amplifies rounding noise layer
    after layer, so a fixed 1e-5*rms band is too narrow deep inside ResNet-50 -- but the band must not depend on the
    implementation under test, or a kernel bug would widen its own acceptance band)."""
    f64 = lambda d: None if d is None else {k: np.asarray(v, np.float64) for k, v in d.items()}
    f32 = lambda d: None if d is None else {k: np.asarray(v, np.float32) for k, v in d.items()}
    e64, _, _ = forward(graph, f64(params), np.asarray(images, np.float64), train=True, masks=f64(masks), state=f64(state))
    e32, _, _ = forward(graph, f32(params), np.asarray(images, np.float32), train=True, masks=f32(masks), state=f32(state))
    out = {}
    for op in graph:
        if op[0] in ('relu', 'maxpool'):
            out[op[1]] = float(np.sqrt(((e32[op[1]].astype(np.float64) - e64[op[1]]) ** 2).mean()))
    return out


BF16_NOISE_MULT = 4  # mixed-precision checks: band = this many times the oracle's OWN bf16-vs-exact discrepancy on that tensor (noise_bands16)


def noise_bands16(graph, params, images, masks=None, state=None, stored=None):
    """rms(oracle with bf16 operand rounding (+ the ambient storage rounding) - exact oracle) of every ReLU / max-pool output on THIS
    input: what bf16 operands do to the decision tensors, from the oracle alone -- the mixed-precision counterpart of noise_bands()."""
    with ops.operand_rounding('bf16'):
        ea, _, _ = forward(graph, params, images, train=True, masks=masks, state=state, stored=stored)
    with ops.no_rounding():
        eb, _, _ = forward(graph, params, images, train=True, masks=masks, state=state)
    return {op[1]: float(np.sqrt(((ea[op[1]] - eb[op[1]]) ** 2).mean())) for op in graph if op[0] in ('relu', 'maxpool')}


def backward(graph, params, env, cache, dout, masks=None, kink=None, kink_mode='fp32', bands=None, stored=None, stored_grad=None):
    """dout: {tensor name: gradient}.  Returns (param grads, tensor grads).
    `kink` (optional): tensors of the implementation under test.  ReLU's derivative jumps at 0 and a
    max-pool routes its gradient to ONE of several near-equal candidates; where the float64 values are
    closer than the band to such a decision boundary either choice is valid for a float32
    evaluation, and the oracle adopts the choice the checked implementation made (there and only there):
    kink[relu_out] = its ReLU output, kink[pool_out + '/idx'] = its arg-max window positions.
    The band is never a function of the implementation's tensors: kink_mode 'fp32' -> max(KINK_BAND*rms, NOISE_MULT*bands[out])
    with `bands` = noise_bands() of the ORACLE; kink_mode 'bf16' (mixed-precision checks only) -> max(KINK_BAND*rms,
    BF16_NOISE_MULT*bands[out]) with `bands` = noise_bands16() of the ORACLE (loss_and_grads computes it when not given)."""
    if kink_mode not in ('fp32', 'bf16'):
        raise ValueError(kink_mode)
    if kink_mode == 'bf16' and kink is not None:
        assert bands is not None, "the bf16 kink band needs the oracle's own noise figures (noise_bands16)"

    def band_of(out, ref_rms):
        thr = KINK_BAND * ref_rms
        if kink_mode == 'bf16':
            return max(thr, BF16_NOISE_MULT * (bands or {}).get(out, 0.0))
        return max(thr, NOISE_MULT * (bands or {}).get(out, 0.0))
    gt = dict(dout)
    gp = OrderedDict()
    if stored_grad is None:
        stored_grad = stored                         # gradients of tensors that are never stored themselves (a BN folded into a gather) may be stored too

    def acc(d, k, v):
        d[k] = v if k not in d else d[k] + v
        if d is gt and stored_grad and k in stored_grad:       # bf16 storage: the gradient of a stored tensor is itself stored, rounded where it is written
            d[k] = ops.stored(d[k])
    for op in reversed(graph):
        kind, out = op[0], op[1]
        if out not in gt:
            continue
        dy = gt[out]
        if kind in ('split', 'shufsplit', 'shufcat', 'dwconv'):
            _backward_shuffle_ops(op, params, env, gt, gp, acc)
            continue
        if kind == 'conv':
            _, _, inp, wname, stride = op
            dx, dw = ops.conv2d_bwd(env[inp], params[wname], dy, stride, need_dx=inp != 'images')
            acc(gp, wname, dw)
            if dx is not None:
                acc(gt, inp, dx)
        elif kind == 'bn':
            _, _, inp, pre = op
            dx, dg, db = ops.bn_train_bwd(dy, params[pre + '/gamma'], cache[out])
            acc(gp, pre + '/gamma', dg)
            acc(gp, pre + '/beta', db)
            acc(gt, inp, dx)
        elif kind == 'relu':
            pre = env[op[2]]
            on = pre > 0
            if kink is not None and out in kink:
                thr = band_of(out, np.sqrt((pre * pre).mean()))
                band = np.abs(pre) < thr
                on = np.where(band, kink[out] > 0, on)
            acc(gt, op[2], dy * on)
        elif kind == 'add':
            acc(gt, op[2], dy)
            acc(gt, op[3], dy)
        elif kind == 'maxpool':
            c = cache[out]
            if kink is not None and out + '/idx' in kink:
                their = kink[out + '/idx'].astype(np.int64)
                diff = their != c['arg']
                if diff.any():                       # accept another window position only if its value ties the maximum
                    x = env[op[2]]
                    n_, ho, wo, ch = dy.shape
                    pt, pl = c['pads']
                    ii = np.argwhere(diff)
                    ih = ii[:, 1] * 2 + their[diff] // 3 - pt
                    iw = ii[:, 2] * 2 + their[diff] % 3 - pl
                    ok = (ih >= 0) & (ih < x.shape[1]) & (iw >= 0) & (iw < x.shape[2])
                    assert ok.all(), 'arg-max outside the image'
                    gap = env[out][diff] - x[ii[:, 0], ih, iw, ii[:, 3]]
                    tie = band_of(out, np.sqrt((x * x).mean()))
                    if kink_mode == 'bf16':          # candidates that are ONE value in the implementation's bf16 tensor differ here by at
                        tie = tie + 2.0 ** -7 * np.abs(env[out][diff])      # most one bf16 ulp (<= 2^-7 |x|: the rounding) plus the noise band
                    assert (np.abs(gap) <= tie).all(), 'arg-max differs beyond the tie band: %d of %d, worst gap %.3e at value %.3e, band %.3e' % (
                        int((np.abs(gap) > tie).sum()), gap.size, float(np.abs(gap).max()), float(np.abs(env[out][diff])[np.argmax(np.abs(gap))]),
                        float(band_of(out, np.sqrt((x * x).mean()))))
                    c = dict(c, arg=their)
            acc(gt, op[2], ops.maxpool3x3s2_bwd(dy, c))
        elif kind == 'gap':
            acc(gt, op[2], ops.gap_bwd(dy, env[op[2]].shape))
        elif kind == 'dropout':
            acc(gt, op[2], dy * masks[out] / op[3])
        elif kind == 'gconv':
            _, _, inp, wname, stride, groups = op
            x = env[inp]
            gw = x.shape[-1] // groups
            dx = np.zeros_like(x)
            dw = np.zeros_like(params[wname])
            # all three products on the bf16 MFMA in the bf16 mode (rounded operands)
            for g in range(groups):
                dxg, dwg = ops.conv2d_bwd(x[..., g * gw:(g + 1) * gw], params[wname][g], dy[..., g * gw:(g + 1) * gw], stride)
                dx[..., g * gw:(g + 1) * gw] = dxg
                dw[g] = dwg
            acc(gp, wname, dw)
            acc(gt, inp, dx)
        elif kind == 'se':
            inp, pre = op[2], op[3]
            s1, s2 = (op[4], op[5]) if len(op) > 4 else ('fc1', 'fc2')
            n1, n2 = pre + '/%s/weights' % s1, pre + '/%s/weights' % s2
            w1, w2 = params[n1].reshape(params[n1].shape[-2:]), params[n2].reshape(params[n2].shape[-2:])
            x = env[inp]
            c = cache[out]
            hw = x.shape[1] * x.shape[2]
            dgate = (dy * x).sum(axis=(1, 2))
            dpre2 = dgate * c['gate'] * (1 - c['gate'])
            acc(gp, n2, ops.mfma_matmul(c['hid'].T, dpre2).reshape(params[n2].shape))
            acc(gp, pre + '/%s/biases' % s2, dpre2.sum(0))
            dpre1 = ops.mfma_matmul(dpre2, w2.T) * (c['hid'] > 0)
            acc(gp, n1, ops.mfma_matmul(c['sq'].T, dpre1).reshape(params[n1].shape))
            acc(gp, pre + '/%s/biases' % s1, dpre1.sum(0))
            dsq = ops.mfma_matmul(dpre1, w1.T)
            acc(gt, inp, dy * c['gate'][:, None, None, :] + dsq[:, None, None, :] / hw)
        elif kind == 'fc':
            dx, dw, db = ops.fc_bwd(env[op[2]], params[op[3]], dy, op[4] is not None)
            acc(gp, op[3], dw)
            if op[4]:
                acc(gp, op[4], db)
            acc(gt, op[2], dx)
    return gp, gt


def _unshuffle(d, fmt):
    """gradient of channel_shuffle = the inverse permutation"""
    c = d.shape[-1]
    idx = ops.channel_shuffle(np.arange(c), fmt)
    out = np.empty_like(d)
    out[..., idx] = d
    return out


def _backward_shuffle_ops(op, params, env, gt, gp, acc):
    kind, out = op[0], op[1]
    dy = gt[out]
    if kind == 'dwconv':
        dx, dw = ops.dwconv3x3_bwd(env[op[2]], params[op[3]], dy, op[4])
        acc(gp, op[3], dw)
        acc(gt, op[2], dx)
    elif kind == 'split':
        acc(gt, op[2], np.concatenate([dy, gt[op[3]]], axis=-1))
    elif kind == 'shufsplit':
        dcat = _unshuffle(np.concatenate([dy, gt[op[4]]], axis=-1), op[5])
        ca = env[op[2]].shape[-1]
        acc(gt, op[2], dcat[..., :ca])
        acc(gt, op[3], dcat[..., ca:])
    elif kind == 'shufcat':
        dcat = _unshuffle(dy, op[4])
        ca = env[op[2]].shape[-1]
        acc(gt, op[2], dcat[..., :ca])
        acc(gt, op[3], dcat[..., ca:])


# ------------------------------------------------------------------------------------------------
# ShuffleNet-v2 (nets/shufflenet_v2.py:32-420)
# ------------------------------------------------------------------------------------------------
SHUFFLENET = {   # variant -> (name, stem ops, [(scope, blocks, width)], final width)
    'small_x0_5': ('ShuffleNet_v2_small_x0_5', 24, [('conv2', 4, 24), ('conv3', 8, 48), ('conv4', 4, 96)], 1024),      # :40-42
    'small_x1': ('ShuffleNet_v2_small', 24, [('conv2', 4, 58), ('conv3', 8, 116), ('conv4', 4, 232)], 1024),            # :43-44
    'small_x1_5': ('ShuffleNet_v2_small_x1_5', 24, [('conv2', 4, 88), ('conv3', 8, 176), ('conv4', 4, 352)], 1024),     # :45-47
    'small': ('ShuffleNet_v2_small_x2', 24, [('conv2', 4, 122), ('conv3', 8, 244), ('conv4', 4, 488)], 2048),           # :48-50, net_base.py:37-42 (alpha=2.0)
    'middle': ('ShuffleNet_v2_middle', 64, [('conv2', 3, 244), ('conv3', 4, 488), ('conv4', 6, 976), ('conv5', 3, 1952)], 2048),   # :234,264-290
    'large': ('ShuffleNet_v2_large_se_res', 128, [('conv2', 10, 340), ('conv3', 10, 680), ('conv4', 23, 1360), ('conv5', 10, 2720)], 2048),  # :305,345-371
}


def shufflenet_graph(variant='small', in_ch=3, data_format='NCHW', blocks_override=None):
    """Returns (graph, spec, feature tensor, net name).  `data_format` only selects WHICH channel permutation
    _channel_shuffle applies (the reference's two branches differ, see ops.channel_shuffle); tensors are NHWC.
    The 'large' variant is built with se=True, residual=True (:305): the SE gate is applied; the residual flag can
    never fire because :91 compares num_outputs with the channel count BEFORE the split (always 2x for stride 1).
    Its stem applies conv1_3x3, conv2_3x3 and conv3_3x3 all to `inputs` (:336-340), so only conv3_3x3 (stride 1,
    128 channels) reaches the max-pool; the other two exist as variables that only see weight decay."""
    name, stem_c, stages, final_c = SHUFFLENET[variant]
    if blocks_override is not None:
        stages = [(s, nb, c) for (s, _, c), nb in zip(stages, blocks_override)]
    se = variant == 'large'
    g, spec = [], []

    def bn(scope, out, cout, relu):
        spec.append((scope + '/BatchNorm/gamma', (cout,), 'gamma'))
        spec.append((scope + '/BatchNorm/beta', (cout,), 'beta'))
        g.append(('bn', out + '/bn', out + '/z', scope + '/BatchNorm'))
        if relu:
            g.append(('relu', out, out + '/bn'))
            return out
        return out + '/bn'

    def conv_bn(scope, out, inp, cin, cout, k, stride):                       # layers.conv2d arg_scope :130-135: BN + ReLU
        spec.append((scope + '/weights', (k, k, cin, cout), 'conv_w'))
        g.append(('conv', out + '/z', inp, scope + '/weights', stride))
        return bn(scope, out, cout, True)

    def sep_bn(scope, out, inp, cin, cout, stride):                           # layers.separable_conv2d arg_scope :136-143: BN, no activation
        spec.append((scope + '/depthwise_weights', (3, 3, cin, 1), 'dw_w'))
        spec.append((scope + '/pointwise_weights', (1, 1, cin, cout), 'conv_w'))
        g.append(('dwconv', out + '/dw', inp, scope + '/depthwise_weights', stride))
        g.append(('conv', out + '/z', out + '/dw', scope + '/pointwise_weights', 1))
        return bn(scope, out, cout, False)

    if variant == 'large':
        spec.append((name + '/conv1/conv1_3x3/weights', (3, 3, in_ch, 64), 'conv_w'))     # dead: :336-339
        spec.extend([(name + '/conv1/conv1_3x3/BatchNorm/gamma', (64,), 'gamma'), (name + '/conv1/conv1_3x3/BatchNorm/beta', (64,), 'beta')])
        spec.append((name + '/conv1/conv2_3x3/weights', (3, 3, in_ch, 64), 'conv_w'))
        spec.extend([(name + '/conv1/conv2_3x3/BatchNorm/gamma', (64,), 'gamma'), (name + '/conv1/conv2_3x3/BatchNorm/beta', (64,), 'beta')])
        x = conv_bn(name + '/conv1/conv3_3x3', 'conv1', 'images', in_ch, stem_c, 3, 1)    # :340
    else:
        x = conv_bn(name + '/conv1/conv_3x3', 'conv1', 'images', in_ch, stem_c, 3, 2)      # :149 / :260
    g.append(('maxpool', 'pool1', x))                                                      # :151
    g.append(('split', 'pool1/s', 'pool1', 'pool1/x'))                                     # first block's _channel_split (:93)
    s_in, x_in, half = 'pool1/s', 'pool1/x', (int(0.5 * stem_c), stem_c - int(0.5 * stem_c))
    nstage = len(stages)
    for si, (scope, nb, c) in enumerate(stages):
        for b in range(nb):
            stride = 2 if b == 0 else 1                                                    # :157-158
            sc = '%s/%s/resBlock_%d' % (name, scope, b)
            t = '%sb%d' % (scope, b)
            shortcut = s_in
            if stride != 1:                                                                # :94-98
                shortcut = sep_bn(sc + '/separable_conv_shortcut_3x3', t + '/ss', s_in, half[0], c, stride)
                shortcut = conv_bn(sc + '/conv_shortcut_1x1', t + '/sc', shortcut, c, c, 1, 1)
            y = conv_bn(sc + '/conv1_1x1', t + '/c1', x_in, half[1], c, 1, 1)              # :101
            y = sep_bn(sc + '/separable_conv2_3x3', t + '/c2', y, c, c, stride)            # :102
            y = conv_bn(sc + '/conv3_1x1', t + '/c3', y, c, c, 1, 1)                       # :103
            if se:                                                                         # :104-105, :79-85 (1x1 convs with biases == FCs on the squeezed map)
                pre = sc
                spec.extend([(pre + '/Conv/weights', (1, 1, c, c // 2), 'fc_w'), (pre + '/Conv/biases', (c // 2,), 'bias'),
                             (pre + '/Conv_1/weights', (1, 1, c // 2, c), 'fc_w'), (pre + '/Conv_1/biases', (c,), 'bias')])
                g.append(('se', t + '/se', y, pre, 'Conv', 'Conv_1'))
                y = t + '/se'
            last = si == nstage - 1 and b == nb - 1
            if last:
                g.append(('shufcat', t, shortcut, y, data_format))                         # :112-113 feeding conv5/conv_1x1
                x = t
            else:
                g.append(('shufsplit', t + '/s', shortcut, y, t + '/x', data_format))      # :112-113 + next block's :93
                s_in, x_in = t + '/s', t + '/x'
            # after a block the tensor has 2c channels; every later block of the stage splits it c / c
            half = (c, c)
    c_last = 2 * stages[-1][2]
    x = conv_bn(name + '/conv5/conv_1x1', 'conv_last', x, c_last, final_c, 1, 1)            # :175-177 (the middle / large nets reuse scope 'conv5', :288,369)
    g.append(('gap', 'features', x))                                                       # :180
    return g, spec, 'features', name


def shufflenet_train_graph(variant, in_ch, num_classes, data_format='NCHW', blocks_override=None):
    g, spec, feat, name = shufflenet_graph(variant, in_ch, data_format, blocks_override)
    final_c = SHUFFLENET[variant][3]
    g = g + [('dropout', 'features_drop', feat, 0.5),                                      # :191
             ('fc', 'logits', 'features_drop', 'classifier/fc_classifier/weights', None)]  # :192-196
    spec = spec + [('classifier/fc_classifier/weights', (final_c, num_classes), 'cls_w')]
    return g, spec


# ------------------------------------------------------------------------------------------------
# ResNet (nets/resnet.py:24-161): bottleneck, conv-BN-ReLU, projection shortcut, pre_act=False
# ------------------------------------------------------------------------------------------------
RESNET_BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 26: [2, 2, 2, 2]}   # nets/resnet.py:35-41
RESNET_OUTPUTS = [256, 512, 1024, 2048]                                                          # nets/resnet.py:43


def resnet_graph(num_layers=50, in_ch=3, variant='resnet', cardinality=32):
    """Returns (graph, weight specs [(name, shape, kind)], feature tensor name, net name).
    variant 'resnet'  : nets/resnet.py as written (mid = C/4, dense 3x3)
            'resnext' : the INTENDED nets/resnext.py (mid = C/2, 3x3 grouped x`cardinality`, conv-BN-ReLU like its
                        base class; the snapshot's override drops BN/ReLU and cannot run -- SURVEY Appendix C)
            'senet'   : 'resnet' + the SE gate of nets/shufflenet_v2.py:79-85 on the block output before the residual
                        add (SE-ResNet-50; the build's composition, SURVEY 8 "SENet-50")."""
    name = {'resnet': 'ResNet', 'resnext': 'ResNeXt', 'senet': 'SENet'}[variant] + '-%d' % num_layers
    g, spec = [], []

    def bn_relu(scope, out, cout, relu):
        spec.append((scope + '/BatchNorm/gamma', (cout,), 'gamma'))
        spec.append((scope + '/BatchNorm/beta', (cout,), 'beta'))
        g.append(('bn', out + '/bn', out + '/z', scope + '/BatchNorm'))
        if relu:
            g.append(('relu', out, out + '/bn'))
            return out
        return out + '/bn'

    def conv_bn(scope, out, inp, cin, cout, k, stride, relu):
        spec.append((scope + '/weights', (k, k, cin, cout), 'conv_w'))
        g.append(('conv', out + '/z', inp, scope + '/weights', stride))
        return bn_relu(scope, out, cout, relu)

    def gconv_bn(scope, out, inp, c, stride, relu):
        gw = c // cardinality
        spec.append((scope + '/weights', (cardinality, 3, 3, gw, gw), 'gconv_w'))
        g.append(('gconv', out + '/z', inp, scope + '/weights', stride, cardinality))
        return bn_relu(scope, out, c, relu)

    x = conv_bn(name + '/conv1/conv_7x7', 'conv1', 'images', in_ch, 64, 7, 2, True)               # nets/resnet.py:109-113
    g.append(('maxpool', 'pool1', x))                                                            # :115
    x, cin = 'pool1', 64
    sc_scope = 'conv_1x1_shortcut' if variant == 'resnext' else 'conv_shortcut_1x1'               # resnext.py:57 / resnet.py:77
    for si, nb in enumerate(RESNET_BLOCKS[num_layers]):
        cout = RESNET_OUTPUTS[si]
        mid = cout // 2 if variant == 'resnext' else cout // 4                                    # resnext.py:60 / resnet.py:82
        for b in range(nb):
            stride = 2 if (b == 0 and si > 0) else 1                                             # :126-139
            sc = '%s/conv%d/resBlock_%d' % (name, si + 2, b)
            t = 's%db%d' % (si + 2, b)
            shortcut = x
            if stride != 1 or cin != cout:                                                       # :72-78
                shortcut = conv_bn(sc + '/' + sc_scope, t + '/sc', x, cin, cout, 1, stride, False)
            y = conv_bn(sc + '/conv1_1x1', t + '/c1', x, cin, mid, 1, 1, True)                   # :82
            if variant == 'resnext':
                y = gconv_bn(sc + '/conv2_3x3', t + '/c2', y, mid, stride, True)                 # resnext.py:61
            else:
                y = conv_bn(sc + '/conv2_3x3', t + '/c2', y, mid, mid, 3, stride, True)          # :83
            y = conv_bn(sc + '/conv3_1x1', t + '/c3', y, mid, cout, 1, 1, False)                 # :84-87
            if variant == 'senet':
                pre = sc + '/se'
                spec.extend([(pre + '/fc1/weights', (cout, cout // 2), 'fc_w'), (pre + '/fc1/biases', (cout // 2,), 'bias'),
                             (pre + '/fc2/weights', (cout // 2, cout), 'fc_w'), (pre + '/fc2/biases', (cout,), 'bias')])
                g.append(('se', t + '/se', y, pre))
                y = t + '/se'
            g.append(('add', t + '/sum', y, shortcut))                                           # :88
            g.append(('relu', t, t + '/sum'))                                                    # :89-90
            x, cin = t, cout
    g.append(('gap', 'features', x))                                                             # :142
    return g, spec, 'features', name


def resnet_train_graph(num_layers, in_ch, num_classes, variant='resnet', classifier=True):
    g, spec, feat, name = resnet_graph(num_layers, in_ch, variant)
    if classifier:
        g = g + [('dropout', 'features_drop', feat, 0.5),                                        # nets/resnet.py:152
                 ('fc', 'logits', 'features_drop', 'classifier/fc_classifier/weights', None)]    # :153-157
        spec = spec + [('classifier/fc_classifier/weights', (RESNET_OUTPUTS[3], num_classes), 'cls_w')]
    return g, spec


def init_params(spec, seed, dtype=np.float64):
    """layers.conv2d default Xavier-uniform weights; BN gamma 1 / beta 0; classifier N(0, 1e-3)."""
    rng = np.random.default_rng(seed)
    p, state = OrderedDict(), OrderedDict()
    for name, shape, kind in spec:
        if kind == 'conv_w':
            k, _, cin, cout = shape
            lim = np.sqrt(6.0 / (k * k * cin + k * k * cout))
            p[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        elif kind == 'gconv_w':
            gw = shape[3]
            lim = np.sqrt(6.0 / (9 * gw + 9 * gw))
            p[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        elif kind == 'fc_w':
            lim = np.sqrt(6.0 / (shape[-2] + shape[-1]))
            p[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        elif kind == 'dw_w':
            lim = np.sqrt(6.0 / (9 * shape[2] + 9))        # Xavier-uniform: fan_in 9*C, fan_out 9*1 (TF's fan rule on [3,3,C,1])
            p[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        elif kind == 'bias':
            p[name] = np.zeros(shape, dtype)
        elif kind == 'cls_w':
            p[name] = (0.001 * rng.standard_normal(shape)).astype(dtype)
        elif kind == 'gamma':
            p[name] = np.ones(shape, dtype)
            pre = name[:-len('/gamma')]
            state[pre + '/moving_mean'] = np.zeros(shape, dtype)
            state[pre + '/moving_variance'] = np.ones(shape, dtype)
        elif kind == 'beta':
            p[name] = np.zeros(shape, dtype)
    return p, state


def perturb(p, seed, scale=0.1):
    rng = np.random.default_rng(seed)
    q = OrderedDict()
    for k, v in p.items():
        q[k] = (v + scale * rng.standard_normal(v.shape)).astype(v.dtype) if (k.endswith('/gamma') or k.endswith('/beta') or k.endswith('/biases')) else v
    return q


def loss_and_grads(graph, params, images, labels, weight_decay=5e-4, masks=None, grad_scale=None, state=None, kink=None,
                   center=None, triplet_margin='off', focal=None, kink_mode='fp32', bands=None, stored=None, stored_grad=None):
    """softmax-CE (+ center loss) or batch-hard triplet, + L2 on conv / fc weights (gamma, beta, biases are not
    regularised).  center = dict(centers=[C,D], alpha=, weight=): loss.py:29-45 on the pooled features, added to the
    total loss with `weight` (the reference leaves the wiring to the caller, loss.py:43).  triplet_margin != 'off':
    the net has no classifier; the loss is the MEAN of loss.py:47-78's per-sample vector.
    Returns (losses, grads incl. wd*w, env, new moving stats [, new centers])."""
    env, cache, new_state = forward(graph, params, images, train=True, masks=masks, state=state, stored=stored)
    n = images.shape[0]
    dout, losses, extra = {}, [], {}
    if triplet_margin != 'off':
        per, df = ops.batch_hard_triplet(env['features'], labels, triplet_margin)
        losses.append(per.mean())
        dout['features'] = df * (1.0 / n if grad_scale is None else grad_scale)
    else:
        if focal is not None:                             # (gamma, alpha): loss.py:18-27 in place of the cross-entropy
            ce, dlogits = ops.focal_loss(env['logits'], labels, focal[0], focal[1], grad_scale)
        else:
            ce, dlogits = ops.softmax_ce(env['logits'], labels, grad_scale)
        losses.append(ce)
        dout['logits'] = dlogits
        if center is not None:
            cl, dfe, newc = ops.center_loss(env['features'], labels, center['centers'], center['alpha'])
            losses.append(cl)
            scale = center['weight'] * (1.0 if grad_scale is None else grad_scale * n)
            dout['features'] = dfe * scale
            extra['centers'] = newc
    if kink is not None and kink_mode == 'bf16' and bands is None:
        bands = noise_bands16(graph, params, images, masks=masks, state=state, stored=stored)
    gp, _ = backward(graph, params, env, cache, dout, masks=masks, kink=kink, kink_mode=kink_mode, bands=bands, stored=stored, stored_grad=stored_grad)
    reg_names = [k for k in params if k.endswith('weights')]      # weights, depthwise_weights, pointwise_weights
    reg = ops.l2_reg([params[k] for k in reg_names], weight_decay)
    for k in reg_names:
        gp[k] = gp.get(k, 0.0) + weight_decay * params[k]      # variables no op reads (ShuffleNet-large's dead stem convs) see only decay
    losses.append(reg)
    if extra:
        return losses, gp, env, new_state, extra
    return losses, gp, env, new_state
