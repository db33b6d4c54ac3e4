"""Static-graph engine for the BN / pooling nets (ResNet family) on libfte.so.

A net is a list of ops over named NHWC tensors (the same description the oracle executes,
oracle/graphnet.py).  `GraphNet` compiles it once -- shapes, one flat parameter arena
[gamma+beta | conv W | classifier W], a state arena for the BN moving statistics, BN+add+ReLU
fusion -- and then forward / backward are fixed sequences of kernel launches on the current stream.
It replaces the TF graph that nets/resnet.py builds and `tf.gradients` differentiates
(data_parallel.py:33); every FLOP runs in libfte.so.
"""
import os
from collections import OrderedDict

import torch

from .. import _lib
from .net_base import Network, side_stream
from .sphere import Variable, same_pads

BN_EPS = 1e-3          # nets/resnet.py:97-99 via layers.batch_norm defaults
BN_DECAY = 0.999


def stem_kpad(k, cin):
    """rows of the im2col'ed stem weight, zero-padded to a multiple of 32 (7*7*3 = 147 -> 160, 3*3*3 = 27 -> 32)"""
    return (k * k * cin + 31) // 32 * 32


STEM_KPAD = stem_kpad(7, 3)


def shuffle_perm(c, data_format):
    """_channel_shuffle (nets/shufflenet_v2.py:66-77) as an index vector: out[k] = in[perm[k]].  The reference's NCHW
    branch views channels as [2, C/2] and transposes; its NHWC branch views them as [C/2, 2] -- a different permutation."""
    idx = torch.arange(c)
    if data_format == 'NCHW':
        return idx.reshape(2, c // 2).t().reshape(-1).tolist()
    return idx.reshape(c // 2, 2).t().reshape(-1).tolist()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Activations(dict):
    """name -> stored activation.  A BN output that was folded into the channel gather that consumes it is never stored;
    asking for it (tests, debugging) recomputes it from z and the kept scale / shift."""

    def __init__(self, net):
        super(_Activations, self).__init__()
        self.net = net

    def __missing__(self, name):
        net = self.net
        if name in getattr(net, 'se_fused', {}):         # the BN output / gated tensor of a fused SE block (never stored)
            which, out, z = net.se_fused[name]
            b = net.bn[out]
            zz = self[z]
            if zz.dtype == torch.int16:
                zz = zz.view(torch.bfloat16).float()
            y = torch.addcmul(b['shift'], zz, b['scale'])
            return y if which == 'y' else y * self[out + '/gate'][:, None, None, :]
        if name not in net.folded:
            raise KeyError(name)
        z, relu = net.folded[name]
        b = net.bn[name]
        zz = self[z]
        if zz.dtype == torch.int16:                      # bf16 storage
            zz = zz.view(torch.bfloat16).float()
        y = torch.addcmul(b['shift'], zz, b['scale'])
        return torch.relu(y) if relu else y


class GraphNet(Network):
    """Network whose body is an op list: ('conv', out, inp, wname, stride) | ('bn', out, inp, prefix) |
    ('relu', out, inp) | ('add', out, a, b) | ('maxpool', out, inp) | ('gap', out, inp) |
    ('dropout', out, inp, keep) | ('fc', out, inp, wname, None) | ('gconv', out, inp, wname, stride, groups) |
    ('se', out, inp, prefix[, scope1, scope2]) | ('dwconv', out, inp, wname, stride) | ('split', out_a, inp, out_b) |
    ('shufsplit', out_s, a, b, out_x, fmt) | ('shufcat', out, a, b, fmt).
    `channel_pad` > 1 (ShuffleNet: 64) stores every activation and weight with its channel count rounded up to that
    multiple; the padding channels are exactly zero in the forward and backward pass (zero weight rows / columns, BN of
    a constant-zero channel with beta 0 stays 0, and every gradient flowing into them is 0), so results equal the
    unpadded net's while all kernels keep their float4 / MFMA-tile granularity.  Heads: 'softmax' (CE on the classifier), 'focal' (loss.py:18-27 instead of CE), 'softmax+center' (CE + weight * center loss
    on the pooled features, loss.py:29-45) and 'triplet' (batch-hard triplet on the pooled features, loss.py:47-78, no
    classifier)."""

    head = 'softmax'
    channel_pad = 1

    def _pc(self, c):
        p = self.channel_pad
        return (c + p - 1) // p * p

    def __init__(self, weight_decay, data_format, name, seed=0):
        super(GraphNet, self).__init__(weight_decay, data_format, name)
        self.seed = seed
        self.built = False
        self._views = {}
        self._gpacks = {}
        self.tower_scale = 1.0
        self.global_step = 0
        self.update_moving_stats = True      # data_parallel.py:242-243: UPDATE_OPS of tower 0 only
        self.dropout_seed = 0
        self._act_n = None
        self.center_weight = 0.0          # 'softmax+center': total loss = CE + center_weight * center_loss
        self.center_alpha = 0.99          # loss.py:29 default
        self.update_centers = True        # False: the loss is evaluated without running centers_update_op (loss.py:39,43 returns it to the caller)
        self.center_comm = None           # set by DataParallel(sync_centers=True): all-gather the scatter rows, one table for all replicas
        self.triplet_margin = None        # 'triplet': None = soft-margin (softplus), loss.py:47
        self.focal_gamma, self.focal_alpha = 1.0, 2.0     # 'focal': loss.py:18 defaults (names as in the reference)

    # ---- to be provided by the subclass ----------------------------------------------------------
    def build_graph(self, in_ch, num_classes):
        """-> (graph, spec) with spec = [(variable name, reference shape, kind)], kind in
        {'conv_w', 'gamma', 'beta', 'cls_w'}."""
        raise NotImplementedError

    # ---- construction -----------------------------------------------------------------------------
    def build(self, height, width, channels, num_classes, device='cuda'):
        _lib.load()
        self.device = torch.device(device)
        self.in_hwc = (height, width, channels)
        self.num_classes = int(num_classes)
        self.cpad = (self.num_classes + 127) // 128 * 128
        self.graph, spec = self.build_graph(channels, num_classes)
        self.spec = OrderedDict((n, (s, k)) for n, s, k in spec)
        # A 3x3 stem of at most 32 filters (ShuffleNet-v2 small: 24) is stored 32 channels wide, not channel_pad wide: its
        # 56x56 output is the largest tensor of the net, and every pass over it (BN statistics / apply, max-pool, their
        # gradients) is pure HBM traffic -- 64-wide storage made 62 % of those bytes padding.  `narrow` = the variables
        # (filter, BN gamma / beta) that follow that width.
        self.narrow = set()
        if self.channel_pad > 32 and os.environ.get('FTE_DIRECT_STEM', '1') != '0':
            for op in self.graph:
                if op[0] == 'conv':
                    k, _, cin, cout = self.spec[op[3]][0]
                    if cin <= 4 and k == 3 and cout <= 32:
                        self.narrow.add(op[3])
                        for o2 in self.graph:
                            if o2[0] == 'bn' and o2[2] == op[1]:
                                self.narrow.update([o2[3] + '/gamma', o2[3] + '/beta'])
        self._infer_shapes()
        small = [(n, s, k) for n, s, k in spec if k in ('gamma', 'beta', 'bias')]
        convs = [(n, s, k) for n, s, k in spec if k in ('conv_w', 'gconv_w', 'fc_w', 'dw_w')]
        cls = [(n, s, k) for n, s, k in spec if k == 'cls_w']
        self.variables = OrderedDict()
        off = 0
        self.ishape = {}
        for n, s, k in small + convs + cls:
            self.ishape[n] = self._internal_shape(s, k, n)
            size = 1
            for d in self.ishape[n]:
                size *= d
            self.variables[n] = Variable(n, k, s, off, size)
            off += (size + 3) // 4 * 4
        self.small_end = self.variables[convs[0][0]].offset
        self.cls_start = self.variables[cls[0][0]].offset if cls else off
        self.arena_size = off
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off + 4, dtype=torch.float32, device=dev)
        self.loss_slots = self.grads[off:off + 4]
        # BN moving statistics: non-trainable state, never all-reduced (each replica keeps its own; only
        # tower 0's are saved: saver.py:36-40)
        self.state = OrderedDict()
        self.state_ref = {}
        for n, s, k in small:
            if k == 'gamma':
                pre = n[:-len('/gamma')]
                cp = self.ishape[n][0]
                self.state[pre + '/moving_mean'] = torch.zeros(cp, dtype=torch.float32, device=dev)
                self.state[pre + '/moving_variance'] = torch.ones(cp, dtype=torch.float32, device=dev)
                self.state_ref[pre + '/moving_mean'] = self.state_ref[pre + '/moving_variance'] = s[0]
        self._init_params()
        self._compile()
        self.built = True
        return self

    def _internal_shape(self, shape, kind, name=None):
        """arena layout of a variable whose reference shape is `shape`"""
        pc = (lambda c: (c + 31) // 32 * 32) if name in self.narrow else self._pc
        if kind == 'cls_w':
            assert pc(shape[0]) == shape[0]
            return (shape[0], self.cpad)
        if kind == 'conv_w':
            k, _, cin, cout = shape
            if cin <= 4:                                       # the stem (image channels): [k*k*cin -> kpad, cout]
                return (stem_kpad(k, cin), pc(cout))
            return (k, k, pc(cin), pc(cout))
        if kind == 'dw_w':
            return (3, 3, pc(shape[2]))
        if kind == 'fc_w':
            return (pc(shape[-2]), pc(shape[-1]))
        if kind in ('gamma', 'beta', 'bias'):
            return (pc(shape[0]),)
        assert self.channel_pad == 1 or kind != 'gconv_w'
        return tuple(shape)

    def view(self, name, arena=None):
        """flat view of a variable in the parameter arena (or in `arena`, e.g. the gradient arena).  Views of the two arenas that
        live as long as the net are cached: the step makes ~1000 of these lookups and is host-bound at small shards."""
        a = self.params if arena is None else arena
        if a is self.params or a is self.grads:
            key = (name, a is self.grads)
            t = self._views.get(key)
            if t is None or t.data_ptr() != a.data_ptr() + self.variables[name].offset * 4:
                v = self.variables[name]
                t = self._views[key] = a[v.offset:v.offset + v.size]
            return t
        v = self.variables[name]
        return a[v.offset:v.offset + v.size]

    def _init_params(self):
        """layers.conv2d default Xavier-uniform; BN gamma 1 / beta 0; classifier N(0, 1e-3) (nets/resnet.py:153-157)."""
        g = torch.Generator().manual_seed(self.seed)
        for n, v in self.variables.items():
            if v.kind == 'conv_w':
                k, _, cin, cout = v.ref_shape
                lim = (6.0 / (k * k * cin + k * k * cout)) ** 0.5
                self.set_variable(n, (torch.rand(v.ref_shape, generator=g) * 2 - 1) * lim)
            elif v.kind == 'gconv_w':
                gw = v.ref_shape[3]
                lim = (6.0 / (18 * gw)) ** 0.5
                self.set_variable(n, (torch.rand(v.ref_shape, generator=g) * 2 - 1) * lim)
            elif v.kind == 'fc_w':
                lim = (6.0 / (v.ref_shape[-2] + v.ref_shape[-1])) ** 0.5
                self.set_variable(n, (torch.rand(v.ref_shape, generator=g) * 2 - 1) * lim)
            elif v.kind == 'dw_w':
                lim = (6.0 / (9 * v.ref_shape[2] + 9)) ** 0.5          # Xavier on [3,3,C,1]: fan_in 9C, fan_out 9
                self.set_variable(n, (torch.rand(v.ref_shape, generator=g) * 2 - 1) * lim)
            elif v.kind == 'cls_w':
                self.set_variable(n, torch.randn(v.ref_shape, generator=g) * 0.001)
            elif v.kind == 'gamma':
                self.set_variable(n, torch.ones(v.ref_shape))

    def get_variable(self, name, arena=None):
        if name in self.state:
            t = self.state[name]
            return t[:self.state_ref[name]].clone() if name in self.state_ref else t.clone()
        v = self.variables[name]
        t = self.view(name, arena).reshape(self.ishape[name])
        ref = v.ref_shape
        if v.kind == 'cls_w':
            return t[:, :self.num_classes].clone()
        if v.kind == 'conv_w':
            k, _, cin, cout = ref
            if cin <= 4:
                return t[:k * k * cin, :cout].reshape(ref).clone()
            return t[:, :, :cin, :cout].clone()
        if v.kind == 'dw_w':
            return t[:, :, :ref[2]].reshape(ref).clone()
        if v.kind == 'fc_w':
            return t[:ref[-2], :ref[-1]].reshape(ref).clone()
        if v.kind in ('gamma', 'beta', 'bias'):
            return t[:ref[0]].clone()
        return t.reshape(ref).clone()

    def set_variable(self, name, value, arena=None):
        if name in self.state:
            t = torch.as_tensor(value, dtype=torch.float32)
            if name in self.state_ref:
                self.state[name][:self.state_ref[name]].copy_(t)
            else:
                self.state[name].copy_(t)
            return
        v = self.variables[name]
        t = torch.as_tensor(value, dtype=torch.float32).to(self.device)
        ref = v.ref_shape
        assert tuple(t.shape) == ref, (name, tuple(t.shape), ref)
        buf = torch.zeros(self.ishape[name], device=self.device)
        if v.kind == 'cls_w':
            buf[:, :self.num_classes] = t
        elif v.kind == 'conv_w':
            k, _, cin, cout = ref
            if cin <= 4:
                buf[:k * k * cin, :cout] = t.reshape(k * k * cin, cout)
            else:
                buf[:, :, :cin, :cout] = t
        elif v.kind == 'dw_w':
            buf[:, :, :ref[2]] = t.reshape(3, 3, ref[2])
        elif v.kind == 'fc_w':
            buf[:ref[-2], :ref[-1]] = t.reshape(ref[-2], ref[-1])
        elif v.kind in ('gamma', 'beta', 'bias'):
            buf[:ref[0]] = t
        else:
            buf = t
        self.view(name, arena).copy_(buf.reshape(-1))

    def load_params(self, params):
        for k, val in params.items():
            self.set_variable(k, val)

    # ---- static analysis ----------------------------------------------------------------------------
    def _infer_shapes(self):
        """self.shapes: stored (channel-padded) shape of every tensor; self.real_c: its true channel count."""
        h, w, c = self.in_hwc
        shp = {'images': (h, w, c)}
        real = {'images': c}
        pc = self._pc

        def put(name, hh, ww, cc):
            real[name] = cc
            shp[name] = (hh, ww, pc(cc))
        for op in self.graph:
            kind, out = op[0], op[1]
            if kind == 'conv':
                ih, iw, _ = shp[op[2]]
                k, _, cin, cout = self.spec[op[3]][0]
                assert cin == real[op[2]], (op, cin, real[op[2]])
                put(out, same_pads(ih, k, op[4])[0], same_pads(iw, k, op[4])[0], cout)
                if op[3] in self.narrow:
                    shp[out] = shp[out][:2] + ((cout + 31) // 32 * 32,)
            elif kind in ('gconv', 'dwconv'):
                ih, iw, _ = shp[op[2]]
                put(out, same_pads(ih, 3, op[4])[0], same_pads(iw, 3, op[4])[0], real[op[2]])
            elif kind in ('bn', 'relu', 'dropout', 'se', 'add'):
                shp[out] = shp[op[2]]
                real[out] = real[op[2]]
            elif kind == 'maxpool':
                ih, iw, cp = shp[op[2]]
                put(out, same_pads(ih, 3, 2)[0], same_pads(iw, 3, 2)[0], real[op[2]])
                shp[out] = shp[out][:2] + (cp,)                 # keeps its input's stored width
            elif kind == 'gap':
                shp[out] = (shp[op[2]][2],)
                real[out] = real[op[2]]
            elif kind == 'fc':
                shp[out] = (self.cpad,)
                real[out] = self.num_classes
            elif kind == 'split':
                ih, iw, _ = shp[op[2]]
                cc = real[op[2]]
                put(out, ih, iw, int(0.5 * cc))
                put(op[3], ih, iw, cc - int(0.5 * cc))
            elif kind == 'shufsplit':
                ih, iw, _ = shp[op[2]]
                cc = real[op[2]] + real[op[3]]
                put(out, ih, iw, int(0.5 * cc))
                put(op[4], ih, iw, cc - int(0.5 * cc))
            elif kind == 'shufcat':
                ih, iw, _ = shp[op[2]]
                put(out, ih, iw, real[op[2]] + real[op[3]])
            else:
                raise ValueError(kind)
        # A tensor stored narrower than channel_pad (the 32-wide stem) may only feed ops that take their width from the stored
        # input: BN / ReLU / max-pool, the channel gathers, and a 1x1 conv (which reads a valid row prefix).  A 3x3, grouped or
        # depthwise conv, an SE gate or an add would lay out its weights / output for pc(real) channels while the kernel is
        # launched with the stored width -- a silent wrong stride.  No net of the factory does that; a new one must not.
        narrow_stored = {n for n in shp if n != 'images' and len(shp[n]) == 3 and shp[n][2] != pc(real[n])}
        for op in self.graph:
            ins = [a for a in op[2:] if isinstance(a, str) and a in narrow_stored]
            if not ins:
                continue
            ok = op[0] in ('bn', 'relu', 'maxpool', 'split', 'shufsplit', 'shufcat', 'dropout') or \
                (op[0] == 'conv' and self.spec[op[3]][0][0] == 1)
            assert ok, 'op %r consumes %s, which is stored %d channels wide (not %d): unsupported consumer of the narrow stem' % (
                op, ins[0], shp[ins[0]][2], pc(real[ins[0]]))
        self.shapes = shp
        self.real_c = real

    def _table(self, entries, width):
        """device int32 gather table of `width` slots from [(src, channel) or None]"""
        t = [-1] * width
        for k, e in enumerate(entries):
            if e is not None:
                t[k] = (e[0] << 16) | e[1]
        return torch.tensor(t, dtype=torch.int32, device=self.device)

    def _gather_tables(self, op):
        """Forward and backward channel-gather tables of a split / shufsplit / shufcat op (fte_channel_gather)."""
        kind = op[0]
        real, shp = self.real_c, self.shapes
        if kind == 'split':                                   # nets/shufflenet_v2.py:60-64
            cc = real[op[2]]
            h = int(0.5 * cc)
            fwd = [(op[1], self._table([(0, k) for k in range(h)], shp[op[1]][2])),
                   (op[3], self._table([(0, h + k) for k in range(cc - h)], shp[op[3]][2]))]
            bwd = [(op[2], self._table([(0, k) if k < h else (1, k - h) for k in range(cc)], shp[op[2]][2]))]
            return dict(ins=(op[2], None), outs=fwd, gouts=(op[1], op[3]), bwd=bwd)
        a, b = op[2], op[3]
        ca, cb = real[a], real[b]
        cc = ca + cb
        fmt = op[5] if kind == 'shufsplit' else op[4]
        perm = shuffle_perm(cc, fmt)                          # shuffled[k] = cat[perm[k]]
        src = [(0, j) if j < ca else (1, j - ca) for j in perm]
        if kind == 'shufsplit':
            h = int(0.5 * cc)
            fwd = [(op[1], self._table(src[:h], shp[op[1]][2])), (op[4], self._table(src[h:], shp[op[4]][2]))]
            gouts = (op[1], op[4])
            where = lambda k: (0, k) if k < h else (1, k - h)
        else:
            fwd = [(op[1], self._table(src, shp[op[1]][2]))]
            gouts = (op[1], None)
            where = lambda k: (0, k)
        inv = [None] * cc
        for k, j in enumerate(perm):
            inv[j] = where(k)
        bwd = [(a, self._table(inv[:ca], shp[a][2])), (b, self._table(inv[ca:], shp[b][2]))]
        return dict(ins=(a, b), outs=fwd, gouts=gouts, bwd=bwd)

    def _compile(self):
        """Fuse bn -> relu and bn -> add -> relu into one BN-apply launch; build consumer counts."""
        g = self.graph
        users = {}
        for i, op in enumerate(g):
            for inp in self._inputs(op):
                users.setdefault(inp, []).append(i)
        plan, skip = [], set()
        for i, op in enumerate(g):
            if i in skip:
                continue
            if op[0] == 'bn':
                out, res, relu, final = op[1], None, 0, op[1]
                u = users.get(out, [])
                if len(u) == 1 and g[u[0]][0] == 'relu':
                    relu, final = 1, g[u[0]][1]
                    skip.add(u[0])
                elif len(u) == 1 and g[u[0]][0] == 'add':
                    add = g[u[0]]
                    other = add[3] if add[2] == out else add[2]
                    u2 = users.get(add[1], [])
                    if len(u2) == 1 and g[u2[0]][0] == 'relu' and self._defined_before(other, i):
                        res, relu, final = other, 1, g[u2[0]][1]
                        skip.update([u[0], u2[0]])
                plan.append(('bn', final, op[2], op[3], res, relu))
            elif op[0] == 'add':
                u = users.get(op[1], [])
                assert len(u) == 1 and g[u[0]][0] == 'relu', 'a bare add is always followed by a ReLU in these nets'
                skip.add(u[0])
                plan.append(('addrelu', g[u[0]][1], op[2], op[3]))
            elif op[0] in ('split', 'shufsplit', 'shufcat'):
                plan.append(('gather', op[1], self._gather_tables(op)))
            else:
                plan.append(op)
        # A BN(+ReLU) output whose only consumer is a channel gather (conv3_1x1 and the stride-2 shortcut's 1x1 of a
        # ShuffleNet block, nets/shufflenet_v2.py:96-113) is normalised INSIDE the gather: the BN op keeps its statistics
        # pass only ('bnstats'), the gather applies scale / shift / ReLU to that source on the way
        # (fte_channel_gather_affine), and the normalised tensor is never written (FTE_BN_GATHER=0: off, A/B hook).
        # SE residual block: BN (no activation) -> SE gate -> add shortcut -> ReLU becomes ONE plan op whose kernels read the BN's input z
        # and write the block's output; the BN output and the gated tensor never exist (csrc/layers.hip "SE residual block",
        # fte_se_*).  FTE_SE_FUSE=0: the separate ops (A/B hook).
        self.se_fused = {}
        if os.environ.get('FTE_SE_FUSE', '1') != '0':
            plan = self._fuse_se_blocks(plan)
        self.folded = {}
        pusers = {}
        for j, op in enumerate(plan):
            for x in self._plan_inputs(op):
                pusers.setdefault(x, []).append(j)
        if os.environ.get('FTE_BN_GATHER', '1') != '0':
            for j, op in enumerate(plan):
                if op[0] == 'bn' and op[4] is None:
                    u = pusers.get(op[1], [])
                    if len(u) == 1 and plan[u[0]][0] == 'gather' and op[1] != self.feature_name:
                        plan[j] = ('bnstats',) + op[1:]
                        self.folded[op[1]] = (op[2], op[5])          # name -> (z, relu)
        # "BN fusion" (fte.h): a conv / grouped conv whose output feeds ONE batch norm leaves that layer's batch statistics in its
        # epilogue (fuse_fwd: plan index of the conv -> plan index of the BN), and the data gradient that completes the gradient of
        # a BN layer's OUTPUT -- the dgrad of its first consumer in plan order, which runs last in the backward walk and takes the
        # other consumer's contribution through `addin` -- applies the ReLU mask and leaves the two sums of the BN backward
        # (fuse_bwd: name of the BN output -> plan index of the BN).  Which of them can run fused (MFMA conv path, storage
        # mode, grouped conv on the bf16 MFMA) is decided where they run.  FTE_BN_FUSE=0: off (A/B hook).
        self.fuse_fwd, self.fuse_bwd = {}, {}
        self.fuse_3x3 = os.environ.get('FTE_BN_FUSE_3X3', '1') != '0'
        # The backward half is OPT-IN (FTE_BN_FUSE_BWD=1 / FTE_BN_FUSE_GBWD=1).  Measured on MI355X at 128 images per GPU, ms per step,
        # forward only / + conv data gradients / + grouped-conv data gradients / no fusion: ResNeXt-50 7.84 / 7.91 / 8.17 / 8.23, ResNet-50
        # 7.27 / 7.38 / - / 7.66, SE-ResNet-50 9.52 / 9.40 / - / 9.87, ShuffleNet-v2 (fp32, 256) 8.10 / 8.11 / - / 8.60: the tile kernels'
        # epilogue waits for its three extra inputs with 3 blocks per CU, which costs what the separate reduce pass cost.
        self.fuse_bwd_conv = os.environ.get('FTE_BN_FUSE_BWD', '0') == '1'
        self.fuse_bwd_gconv = os.environ.get('FTE_BN_FUSE_GBWD', '0') == '1'
        if os.environ.get('FTE_BN_FUSE', '1') != '0':
            producer = {op[1]: j for j, op in enumerate(plan) if op[0] in ('conv', 'gconv')}
            for j, op in enumerate(plan):
                if op[0] not in ('bn', 'bnstats', 'seblock'):
                    continue
                i = producer.get(op[2])
                if i is not None and pusers.get(op[2], []) == [j]:
                    self.fuse_fwd[i] = j
                us = pusers.get(op[1], [])
                if op[0] == 'bn' and us and op[1] != self.feature_name:
                    first = plan[us[0]]
                    if (first[0] == 'conv' and len(us) <= 2) or (first[0] == 'gconv' and len(us) == 1):
                        self.fuse_bwd[op[1]] = j
        # ... and the normalise pass of a BN + ReLU whose output feeds ONE conv / grouped conv that itself runs fused can move into
        # that consumer's operand loader (fold_apply: plan index of the BN -> plan index of the consumer): the consumer reads the
        # BN's input z, applies scale / shift / ReLU on the way to the matrix cores and writes the normalised tensor back for the
        # filter gradient; the bn_apply launch and its pass over the tensor disappear (fte.h, fte_conv2d_bn_fwd's in_scale).
        # Whether the consumer's kernel takes it (bf16 storage, pointwise stride-1 conv of 64 / 128 / 256 channels, or a
        # stride-1 grouped conv on the bf16 MFMA) is decided where it runs.  FTE_BN_FOLD=0: off (A/B hook).
        self.fold_apply = {}
        if self.fuse_fwd and os.environ.get('FTE_BN_FOLD', '1') != '0':
            for j, op in enumerate(plan):
                if op[0] == 'bn' and op[4] is None and op[5] and op[1] != self.feature_name and j in self.fuse_fwd.values():
                    us = pusers.get(op[1], [])
                    if len(us) == 1 and us[0] in self.fuse_fwd and plan[us[0]][0] in ('conv', 'gconv') and plan[us[0]][2] == op[1]:
                        self.fold_apply[j] = us[0]
        # Shortcut branches of the residual blocks (conv 1x1 -> BN without activation, consumed only as the `res` of the block's last
        # BN or by its add + ReLU): independent of the block's main branch, so the forward walk queues them on the side stream and the
        # consumer waits for their event (shortcut_fwd: plan index -> True for the conv and the BN).  FTE_SHORTCUT_SIDE=0: off (A/B hook).
        self.shortcut_fwd = {}
        if os.environ.get('FTE_SHORTCUT_SIDE', '1') != '0':
            producer = {op[1]: j for j, op in enumerate(plan) if op[0] == 'conv'}
            for j, op in enumerate(plan):
                if op[0] != 'bn' or op[4] is not None or op[5]:
                    continue
                us = pusers.get(op[1], [])
                if len(us) != 1 or op[1] == self.feature_name:
                    continue
                cons = plan[us[0]]
                as_res = (cons[0] == 'bn' and cons[4] == op[1] and cons[2] != op[1]) or (cons[0] == 'addrelu' and op[1] in (cons[2], cons[3])) or \
                    (cons[0] == 'seblock' and cons[4] == op[1] and cons[2] != op[1])
                i = producer.get(op[2])
                if as_res and i is not None and pusers.get(op[2], []) == [j] and i == j - 1:
                    self.shortcut_fwd[i] = True
                    self.shortcut_fwd[j] = True
        self.plan = plan
        self.has_classifier = plan[-1][0] == 'fc'

    @staticmethod
    def _inputs(op):
        if op[0] in ('add', 'shufsplit', 'shufcat'):
            return [op[2], op[3]]
        return [op[2]]

    def _plan_inputs(self, op):
        """tensors a PLAN op reads"""
        if op[0] == 'gather':
            return [x for x in op[2]['ins'] if x is not None]
        if op[0] == 'bn':
            return [op[2]] + ([op[4]] if op[4] is not None else [])
        if op[0] == 'addrelu':
            return [op[2], op[3]]
        if op[0] == 'seblock':
            return [op[2], op[4]]
        return self._inputs(op)

    def _fuse_se_blocks(self, plan):
        """('bn', y, z, pre, None, 0) -> ('se', s, y, ...) -> ('addrelu', out, s, shortcut), each the only user of its input, becomes
        ('seblock', out, z, pre, shortcut, se op, y, s) at the add's place (nets/resnet.py:63-92 with use_se)."""
        users = {}
        for j, op in enumerate(plan):
            for x in self._plan_inputs(op):
                users.setdefault(x, []).append(j)
        drop, repl = set(), {}
        for j, op in enumerate(plan):
            if op[0] != 'bn' or op[4] is not None or op[5] or len(self.shapes[op[1]]) != 3 or op[1] == self.feature_name:
                continue
            u = users.get(op[1], [])
            if len(u) != 1 or plan[u[0]][0] != 'se' or plan[u[0]][2] != op[1]:
                continue
            se = plan[u[0]]
            u2 = users.get(se[1], [])
            if len(u2) != 1 or plan[u2[0]][0] != 'addrelu' or se[1] == self.feature_name:
                continue
            ar = plan[u2[0]]
            sc = ar[3] if ar[2] == se[1] else ar[2]
            if sc == se[1] or self.shapes[sc] != self.shapes[op[1]] or self.shapes[op[1]][-1] % 4:
                continue
            repl[u2[0]] = ('seblock', ar[1], op[2], op[3], sc, se, op[1], se[1])
            drop.update([j, u[0]])
            self.se_fused[op[1]] = ('y', ar[1], op[2])
            self.se_fused[se[1]] = ('s', ar[1], op[2])
        return [repl.get(j, op) for j, op in enumerate(plan) if j not in drop]

    def _defined_before(self, name, idx):
        if name == 'images':
            return True
        for j in range(idx):
            o = self.graph[j]
            if o[1] == name or (o[0] == 'split' and o[3] == name) or (o[0] == 'shufsplit' and o[4] == name):
                return True
        return False

    # ---- buffers --------------------------------------------------------------------------------------
    S16_OPS = ('conv', 'bn', 'bnstats', 'gconv', 'dwconv', 'gather', 'se', 'seblock', 'maxpool', 'addrelu', 'gap', 'dropout', 'fc')

    def _storage16(self):
        """bf16 STORAGE ('bf16s', fte.h): the tensors between the layers live in HBM as bf16 -- implemented for the op sets of the ResNet
        family (conv / BN / grouped 3x3 on the bf16 MFMA / max-pool / add+ReLU / GAP; BASELINE.json configs[2]) and of ShuffleNet-v2
        (depthwise 3x3, channel gathers with folded BN) and SE gates.  A grouped 3x3 that cannot run on the bf16 MFMA (channels per
        group not 4 / 8 / 16 / 32) makes the net fall back to the 'bf16' operand mode: same MFMA precision, fp32 tensors."""
        if not _lib.bf16_storage():
            return False
        ok = all(op[0] in self.S16_OPS for op in self.plan) and \
            all(self._gconv_pack(op) is not None for op in self.plan if op[0] == 'gconv')
        if not ok and not getattr(self, '_s16_note', False):
            self._s16_note = True
            print('%s: bf16 storage is not implemented for one of its ops; this net runs bf16 MFMA operands with fp32 tensors' % self.name)
        return ok

    def _is16(self, name):
        return name in self.h16

    def _alloc(self, n):
        s16 = self._storage16()
        if self._act_n == n and getattr(self, '_act_s16', False) == s16:
            return
        self._act_s16 = s16
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i16 = dict(dtype=torch.int16, device=dev)
        # names of the tensors stored as bf16: conv / BN / grouped-conv / pool / add outputs; the stem conv's output (an fp32 GEMM),
        # the pooled features and everything after them stay fp32
        self.h16 = set()
        if s16:
            for op in self.plan:
                if op[0] in ('gconv', 'dwconv', 'se', 'seblock', 'maxpool', 'addrelu') or (op[0] == 'bn' and len(self.shapes[op[1]]) == 3):
                    self.h16.add(op[1])          # (a batch norm behind the pooling has a rank-1 output: the features stay fp32)
                elif op[0] == 'conv':          # every conv writes bf16: the MFMA convs, the direct 3x3 stem, and the im2col stem (a 1x1 conv
                    self.h16.add(op[1])        # of `kpad` bf16 columns on the bf16-source kernels)
                elif op[0] == 'gather':
                    self.h16.update(name for name, _ in op[2]['outs'])
            self._pack_entries = []
        # tensors whose GRADIENT is stored as bf16: the stored ones and the BN outputs folded into a gather (never stored themselves)
        self.g16 = set(self.h16) | (set(self.folded) if s16 else set())
        if s16:          # ... and the gated tensor of a fused SE block: its gradient g = dy * (out > 0) is stored (the shortcut's gradient too)
            self.g16.update(op[7] for op in self.plan if op[0] == 'seblock')
        if s16:          # every bf16-storage entry point reads its tensor input as bf16: an fp32 input would be misread silently
            for op in self.plan:
                ins = []
                if op[0] == 'conv' and self.shapes[op[2]][-1] >= 32:
                    ins = [op[2]]
                elif op[0] in ('gconv', 'dwconv', 'maxpool', 'se'):
                    ins = [op[2]]
                elif op[0] == 'seblock':
                    ins = [op[2], op[4]]
                for x in ins:
                    assert x in self.h16, 'bf16 storage: %s reads %s, which is stored as fp32' % (op[0] + ' ' + op[1], x)
        self.t = _Activations(self)
        self.bn = {}
        self.ident = {}
        need = 1 << 20
        q = _lib.query
        for op in self.plan:
            kind, out = op[0], op[1]
            if kind == 'gather':
                for name, _ in op[2]['outs']:
                    self.t[name] = torch.empty((n,) + self.shapes[name], **(i16 if s16 else f32))
                continue
            shape = (n,) + self.shapes[out]
            if kind != 'bnstats':
                self.t[out] = torch.empty(shape, **(i16 if out in self.h16 else f32))
            if kind in ('bn', 'bnstats', 'seblock'):
                c = shape[-1]
                self.bn[out] = dict(mean=torch.empty(c, **f32), rstd=torch.empty(c, **f32), scale=torch.empty(c, **f32),
                                    shift=torch.empty(c, **f32), coef=torch.empty(3 * c, **f32))
                need = max(need, q('fte_bn_ws_bytes', c))
            elif kind == 'conv':
                ih, iw, cin = self.shapes[op[2]]
                k = self.spec[op[3]][0][0]
                cout = shape[-1]
                if cin >= 32:
                    need = max(need, q('fte_conv2d_fwd_ws_bytes', n, ih, iw, cin, cout, k, op[4]),
                               q('fte_conv2d_dgrad_ws_bytes', n, ih, iw, cin, cout, k, op[4]),
                               q('fte_conv2d_wgrad_ws_bytes', n, ih, iw, cin, cout, k, op[4]),
                               q('fte_conv2d_bn_fwd_ws_bytes', n, ih, iw, cin, cout, k, op[4]),
                               q('fte_conv2d_dgrad_bn_ws_bytes', n, ih, iw, cin, cout, k, op[4]))
                    if s16:          # bf16 packs of the filter: HWIO (data gradient) and [tap][cout][cin] (forward), refreshed every step
                        self._pack_entries.append((op[3], self.variables[op[3]].offset, k, cin, cout))
                elif self._direct_stem(k, cin, cout):
                    need = max(need, q('fte_conv3x3_first_wgrad_ws_bytes', n, ih, iw, cin, cout, op[4]))
                else:
                    oh, ow, _ = self.shapes[out]
                    kpad = stem_kpad(k, cin)
                    self.cols = torch.empty(n * oh * ow, kpad, **(i16 if s16 else f32))
                    need = max(need, q('fte_gemm_ws_bytes', n * oh * ow, cout, kpad))
                    if s16:          # the stem as a 1x1 conv of kpad bf16 columns: packs like any other conv's, [1, 1, kpad, cout]
                        self._pack_entries.append((op[3], self.variables[op[3]].offset, 1, kpad, cout))
                        need = max(need, q('fte_conv2d_fwd_ws_bytes', n, oh, ow, kpad, cout, 1, 1), q('fte_conv2d_bn_fwd_ws_bytes', n, oh, ow, kpad, cout, 1, 1),
                                   q('fte_conv2d_wgrad_ws_bytes', n, oh, ow, kpad, cout, 1, 1))
            elif kind == 'dwconv':
                ih, iw, cc = self.shapes[op[2]]
                need = max(need, q('fte_dwconv3x3_wgrad_ws_bytes', n, ih, iw, cc, op[4]))
            elif kind == 'gconv':
                ih, iw, cc = self.shapes[op[2]]
                need = max(need, q('fte_gconv3x3_wgrad_ws_bytes', n, ih, iw, cc, op[5], op[4]))
                if cc % 32 == 0 and cc // op[5] in (4, 8, 16, 32):
                    need = max(need, q('fte_gconv3x3_wgrad_bf16_ws_bytes', n, ih, iw, cc, op[5], op[4]),
                               q('fte_gconv3x3_bn_ws_bytes', n, ih, iw, cc, op[4]))
            if kind in ('se', 'seblock'):
                cc = shape[-1]
                hd = self._se_names(op[5] if kind == 'seblock' else op)[4]
                if kind == 'seblock':          # per-image sums of the backward pass, the squeeze in xhat units
                    for nm in ('xm', 's1', 's2'):
                        self.t[out + '/' + nm] = torch.empty(n, cc, **f32)
                self.t[out + '/sq'] = torch.empty(n, cc, **f32)
                self.t[out + '/hid'] = torch.empty(n, hd, **f32)
                self.t[out + '/gate'] = torch.empty(n, cc, **f32)
                need = max(need, q('fte_gemm_ws_bytes', n, cc, hd), q('fte_gemm_ws_bytes', n, hd, cc))
                # backward scratch: dsq is shared by all SE blocks of a width; dgate / dhid are per block -- the gate's weight gradients
                # read them on the side stream while the walk has moved on to the next block
                if ('se', 'dsq', cc) not in self.ident:
                    self.ident[('se', 'dsq', cc)] = torch.empty(n, cc, **f32)
                self.ident[('se', out, 'dgate')] = torch.empty(n, cc, **f32)
                self.ident[('se', out, 'dhid')] = torch.empty(n, hd, **f32)
            elif kind == 'addrelu':
                cc = shape[-1]
                if cc not in self.ident:
                    self.ident[cc] = (torch.ones(cc, **f32), torch.zeros(cc, **f32))
            elif kind == 'maxpool':
                self.t[out + '/idx'] = torch.empty(shape, dtype=torch.uint8, device=dev)
            elif kind == 'dropout':
                self.t[out + '/mask'] = torch.empty(shape, **f32)
            elif kind == 'fc':
                need = max(need, q('fte_gemm_ws_bytes', n, self.cpad, self.shapes[op[2]][0]))
        if s16:
            from ._packs import FilterPacks
            self.packs = FilterPacks(self._pack_entries, dev, head=int(os.environ.get('FTE_PACK_HEAD', '4')))
            self.w16, self.w16t = self.packs.w16, self.packs.w16t
        self.G = torch.empty(n, self.cpad, **f32)
        self.loss_rows = torch.empty(n, **f32)
        fdim = self.shapes[self.feature_name][0]
        self.dfeat = torch.empty(n, fdim, **f32)
        self.ones_n = torch.ones(n, **f32)
        need = max(need, 4 * n * fdim, 12 * n * n)
        self.ws = torch.empty((need + 3) // 4 + 1024, **f32)
        self.ws_bytes = self.ws.numel() * 4
        # filter gradients run on a second HIP stream beside the data-gradient chain (backward_body): their own workspace
        self.side = side_stream(dev, int(os.environ.get('FTE_SIDE_PRIO', '0'))) if os.environ.get('FTE_SIDE_STREAM', '1') != '0' else None
        self.ws_side = torch.empty_like(self.ws) if self.side is not None else self.ws
        self.side_batch = int(os.environ.get('FTE_SIDE_BATCH', '3'))
        self._act_n = n

    # ---- forward ----------------------------------------------------------------------------------------
    def _check_images(self, images):
        if not (isinstance(images, torch.Tensor) and images.is_cuda and images.dtype == torch.float32):
            raise TypeError('images must be a float32 CUDA tensor in NHWC (data.py:275-279 layout)')
        return images.contiguous()

    def _run_forward(self, images, is_training):
        x = self._check_images(images)
        n, h, w, ch = x.shape
        assert (h, w, ch) == self.in_hwc, ((h, w, ch), self.in_hwc)
        self._alloc(n)
        st = _stream()
        call = _lib.call
        T = self.t
        T['images'] = x
        s16 = self._act_s16
        h16 = self.h16
        pack_ev = None
        side_packs = None
        self._reg_ev = None
        if is_training and self.side is not None and os.environ.get('FTE_REG_SIDE', '1') != '0':
            # Network._regularize's sum over the decayed weights (one pass over the arena: 25-40 us) depends on nothing the walk computes:
            # it runs on the side stream under the first layers instead of between the loss head's launches; loss_function waits for it
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            nreg = self.arena_size - self.small_end
            # (into a scratch scalar of its own, not the displayed slot: the previous step's reg_loss stays readable until this step's
            # loss_function copies the new value in; the scale used is remembered -- a tower_scale / weight_decay changed between
            # forward() and loss_function() makes loss_function recompute)
            if getattr(self, '_reg_scratch', None) is None:
                self._reg_scratch = torch.zeros(4, dtype=torch.float32, device=self.params.device)
            self._reg_scale = 0.5 * self.weight_decay * self.tower_scale
            call('fte_sumsq', self.params[self.small_end:], nreg, self._reg_scale,
                 self._reg_scratch[0:1], self.ws_side, self.ws_bytes, self.side.cuda_stream)
            self._reg_ev = self.side.record_event()
        if s16:
            # every filter's bf16 packs, refreshed once per step.  The walk's first layers need only THEIR forward packs: those are
            # made here; the rest -- the other layers' forward packs, every HWIO pack (read by the backward pass only) and the
            # grouped convs' packs -- are made on the side stream under the first layers, and the first conv outside the head
            # waits for them (0.27 ms of launches off the ResNeXt-50 step's critical path at 128 images).
            side = self.side
            if side is not None and self.packs.head_names != set(self.w16t):
                self.packs.refresh_head(self.params, st)

                def side_packs():
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    sst = side.cuda_stream
                    self.packs.refresh_rest(self.params, sst)
                    for op in self.plan:
                        if op[0] == 'gconv':
                            pk = self._gconv_pack(op)
                            if pk is not None:
                                call('fte_gconv3x3_pack_bf16', self.view(op[3]), pk[0], pk[1], self.shapes[op[2]][-1], op[5], sst)
                    return side.record_event()
                # A 7x7 stem's im2col (150-200 MB of strided traffic) and the packs (the same again) choke each other when they run
                # side by side: im2col 74 -> 240-260 us beside ResNet-50's 47 MB of packs (profiles/r5_resnet50_*).  The packs start
                # behind the im2col instead, under the stem's GEMM (FTE_PACK_AFTER_STEM=0: at the start of the walk, as before).
                first = self.plan[0]
                stem_cols = first[0] == 'conv' and self.shapes[first[2]][-1] < 32 and not self._direct_stem(self.spec[first[3]][0][0], self.shapes[first[2]][-1], self.shapes[first[1]][-1])
                if not (stem_cols and os.environ.get('FTE_PACK_AFTER_STEM', '1') != '0'):
                    pack_ev = side_packs()
                    side_packs = None
            else:
                self.packs.refresh(self.params, st)
        stats_done = set()                               # BN plan ops whose statistics came out of the producing conv's epilogue
        folded_in = {}                                   # consumer plan op -> BN plan op whose normalise pass its loader applies
        upd = self.update_moving_stats

        def folds(bj):
            """does the consumer of BN plan op bj take the normalise pass into its loader? (bf16 storage, training, statistics fused)"""
            cj = self.fold_apply.get(bj)
            if cj is None or not (s16 and is_training and bj in stats_done):
                return False
            cop = self.plan[cj]
            ih_, iw_, cin_ = self.shapes[cop[2]]
            if cop[0] == 'gconv':                       # (its fused launch carries the fold: FTE_BN_FUSE_3X3=0 takes both away)
                return cop[4] == 1 and self._gconv_pack(cop) is not None and self.fuse_3x3
            return bool(_lib.query('fte_conv2d_bn_fwd_folds', n, ih_, iw_, cin_, self.shapes[cop[1]][-1], self.spec[cop[3]][0][0], cop[4], 1))

        def fold_args(j):
            """(x, in_scale, in_shift, y_side) of consumer plan op j"""
            bj = folded_in.get(j)
            if bj is None:
                return T[self.plan[j][2]], None, None, None
            bop = self.plan[bj]
            b = self.bn[bop[1]]
            return T[bop[2]], b['scale'], b['shift'], T[bop[1]]

        def bn_args(j):
            bop = self.plan[j]
            b, pre = self.bn[bop[1]], bop[3]
            return (self.view(pre + '/gamma'), self.view(pre + '/beta'), b['mean'], b['rstd'], b['scale'], b['shift'],
                    self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None, BN_EPS, BN_DECAY)
        main_s = torch.cuda.current_stream()
        sc_side = self.side if (self.side is not None and is_training and self.shortcut_fwd) else None
        sc_ev = {}                                       # shortcut tensor -> event of the side stream that completes it
        st_main, ws_main = st, self.ws
        for j, op in enumerate(self.plan):
            kind, out = op[0], op[1]
            on_side = sc_side is not None and j in self.shortcut_fwd
            st, ws_j = (sc_side.cuda_stream, self.ws_side) if on_side else (st_main, ws_main)
            if kind == 'conv':
                _, _, inp, wname, stride = op
                ih, iw, cin = self.shapes[inp]
                k = self.spec[wname][0][0]
                cout = self.shapes[out][-1]
                if on_side:
                    sc_side.wait_event(main_s.record_event())          # the block's input is complete (the packs were made on this stream)
                if not on_side and pack_ev is not None and cin >= 32 and wname not in self.packs.head_names:
                    torch.cuda.current_stream().wait_event(pack_ev)          # the side stream's packs (once: every later layer is behind this wait)
                    pack_ev = None
                if cin >= 32 and is_training and j in self.fuse_fwd and (k == 1 or not s16 or self.fuse_3x3):          # conv + the batch statistics of its output ("BN fusion")
                    xin, isc, ish, yside = fold_args(j)
                    call('fte_conv2d_bn_fwd', xin, self.w16t[wname] if s16 else self.view(wname), T[out], *bn_args(self.fuse_fwd[j]),
                         isc, ish, yside, n, ih, iw, cin, cout, k, stride, 1 if s16 else 0, ws_j, self.ws_bytes, st)
                    stats_done.add(self.fuse_fwd[j])
                elif cin >= 32 and s16:          # bf16 storage: bf16 x in, bf16 z out, filters packed once per step
                    call('fte_conv2d_fwd_s16', T[inp], self.w16t[wname], None, None, None, None, T[out], None, None,
                         n, ih, iw, cin, cout, k, stride, ws_j, self.ws_bytes, st)
                elif cin >= 32:
                    call('fte_conv2d_fwd', T[inp], self.view(wname), None, None, None, None, T[out],
                         n, ih, iw, cin, cout, k, stride, ws_j, self.ws_bytes, st)
                elif self._direct_stem(k, cin, cout):          # 3x3 stem of 32 / 64 stored filters: the direct MFMA kernel
                    if s16:
                        call('fte_conv3x3_first_fwd_s16', T[inp], self.view(wname), None, None, None, T[out], n, ih, iw, cin, cout, stride, st)
                    else:
                        call('fte_conv3x3_first_fwd', T[inp], self.view(wname), None, None, None, T[out], n, ih, iw, cin, cout, stride, st)
                elif s16:                                      # other stems (7x7), bf16 storage: bf16 columns, then a 1x1 conv of kpad channels
                    oh, ow, _ = self.shapes[out]
                    kpad = stem_kpad(k, cin)
                    call('fte_im2col_first_s16', T[inp], self.cols, n, ih, iw, cin, k, stride, kpad, st)
                    if side_packs is not None:                 # the rest of the filter packs, from here on (see above)
                        pack_ev = side_packs()
                        side_packs = None
                    if is_training and j in self.fuse_fwd:
                        call('fte_conv2d_bn_fwd', self.cols, self.w16t[wname], T[out], *bn_args(self.fuse_fwd[j]), None, None, None,
                             n, oh, ow, kpad, cout, 1, 1, 1, self.ws, self.ws_bytes, st)
                        stats_done.add(self.fuse_fwd[j])
                    else:
                        call('fte_conv2d_fwd_s16', self.cols, self.w16t[wname], None, None, None, None, T[out], None, None,
                             n, oh, ow, kpad, cout, 1, 1, self.ws, self.ws_bytes, st)
                else:                                          # other stems (7x7): im2col + dense MFMA GEMM
                    oh, ow, _ = self.shapes[out]
                    kpad = stem_kpad(k, cin)
                    call('fte_im2col_first', T[inp], self.cols, n, ih, iw, cin, k, stride, kpad, st)
                    call('fte_gemm_nn', self.cols, self.view(wname), None, T[out], n * oh * ow, cout, kpad,
                         self.ws, self.ws_bytes, st)
            elif kind == 'dwconv':
                ih, iw, c = self.shapes[op[2]]
                call('fte_dwconv3x3_fwd_s16' if s16 else 'fte_dwconv3x3_fwd', T[op[2]], self.view(op[3]), T[out], n, ih, iw, c, op[4], st)
            elif kind == 'gather':
                a, b = op[2]['ins']
                fa, fb = self.folded.get(a), self.folded.get(b)
                outs = op[2]['outs']
                if fa is None and fb is None and len(outs) == 1:
                    name, table = outs[0]
                    co = self.shapes[name][-1]
                    call('fte_channel_gather_s16' if s16 else 'fte_channel_gather', T[a], T[b] if b else None, T[name], table, T[name].numel() // co,
                         self.shapes[a][-1], self.shapes[b][-1] if b else 0, co, st)
                else:                                          # both halves in one launch, BN applied to a folded source
                    sa = (T[fa[0]], self.bn[a]['scale'], self.bn[a]['shift'], fa[1]) if fa else (T[a], None, None, 0)
                    sb = (T[fb[0]], self.bn[b]['scale'], self.bn[b]['shift'], fb[1]) if fb else (T[b] if b else None, None, None, 0)
                    (n0, t0), (n1, t1) = outs[0], (outs[1] if len(outs) > 1 else (None, None))
                    co0 = self.shapes[n0][-1]
                    call('fte_channel_gather_affine_s16' if s16 else 'fte_channel_gather_affine', sa[0], sb[0], T[n0], t0, co0, T[n1] if n1 else None, t1,
                         self.shapes[n1][-1] if n1 else 0, T[n0].numel() // co0, self.shapes[a][-1],
                         self.shapes[b][-1] if b else 0, sa[1], sa[2], sa[3], sb[1], sb[2], sb[3], st)
            elif kind == 'bnstats':
                _, _, inp, pre, _, _ = op
                b = self.bn[out]
                c = self.shapes[out][-1]
                rows = T[inp].numel() // c
                if j in stats_done:
                    pass
                elif is_training and s16:
                    upd = self.update_moving_stats
                    call('fte_bn_train_stats_s16', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'),
                         b['mean'], b['rstd'], b['scale'], b['shift'],
                         self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None,
                         rows, c, BN_EPS, BN_DECAY, 1 if inp in h16 else 0, self.ws, self.ws_bytes, st)
                elif is_training:
                    upd = self.update_moving_stats
                    call('fte_bn_train_stats', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'),
                         b['mean'], b['rstd'], b['scale'], b['shift'],
                         self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None,
                         rows, c, BN_EPS, BN_DECAY, self.ws, self.ws_bytes, st)
                else:
                    call('fte_bn_infer_coef', self.view(pre + '/gamma'), self.view(pre + '/beta'),
                         self.state[pre + '/moving_mean'], self.state[pre + '/moving_variance'], b['scale'], b['shift'], c, BN_EPS, st)
            elif kind == 'bn':
                _, _, inp, pre, res, relu = op
                b = self.bn[out]
                c = self.shapes[out][-1]
                rows = T[out].numel() // c
                resbuf = T[res] if res is not None else None
                if res is not None and res in sc_ev:
                    main_s.wait_event(sc_ev.pop(res))                  # the shortcut branch (side stream) has written it
                if j in stats_done and folds(j):  # ... which the consumer's operand loader takes over (it also writes T[out])
                    folded_in[self.fold_apply[j]] = j
                elif j in stats_done:            # scale / shift are there already: the normalise pass alone
                    assert res is None or not s16 or res in h16, 'bf16 storage: the shortcut of %s is an fp32 tensor' % out
                    call('fte_bn_apply', T[inp], b['scale'], b['shift'], resbuf, T[out], rows, c, relu, ((1 if inp in h16 else 0) | 2) if s16 else 0, st)
                elif s16:
                    assert res is None or res in h16, 'bf16 storage: the shortcut of %s is an fp32 tensor' % out
                    fl = (1 if inp in h16 else 0) | 2
                    if is_training:
                        upd = self.update_moving_stats
                        call('fte_bn_train_fwd_s16', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'), resbuf, T[out],
                             b['mean'], b['rstd'], b['scale'], b['shift'],
                             self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None,
                             rows, c, BN_EPS, BN_DECAY, relu, fl, ws_j, self.ws_bytes, st)
                    else:
                        call('fte_bn_infer_fwd_s16', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'),
                             self.state[pre + '/moving_mean'], self.state[pre + '/moving_variance'], resbuf, T[out],
                             b['scale'], b['shift'], rows, c, BN_EPS, relu, fl, st)
                elif is_training:
                    upd = self.update_moving_stats
                    call('fte_bn_train_fwd', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'), resbuf, T[out],
                         b['mean'], b['rstd'], b['scale'], b['shift'],
                         self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None,
                         rows, c, BN_EPS, BN_DECAY, relu, ws_j, self.ws_bytes, st)
                else:
                    call('fte_bn_infer_fwd', T[inp], self.view(pre + '/gamma'), self.view(pre + '/beta'),
                         self.state[pre + '/moving_mean'], self.state[pre + '/moving_variance'], resbuf, T[out],
                         b['scale'], b['shift'], rows, c, BN_EPS, relu, st)
                if on_side:
                    sc_ev[out] = sc_side.record_event()
            elif kind == 'gconv':
                ih, iw, c = self.shapes[op[2]]
                pk = self._gconv_pack(op)
                if pk is not None:                             # bf16 MFMA mode: block-diagonal slices on the matrix cores
                    if not s16 or self.side is None or self.packs.head_names == set(self.w16t):
                        call('fte_gconv3x3_pack_bf16', self.view(op[3]), pk[0], pk[1], c, op[5], st)
                    elif pack_ev is not None:                  # (packed on the side stream at the start of the walk)
                        torch.cuda.current_stream().wait_event(pack_ev)
                        pack_ev = None
                    if s16 and is_training and j in self.fuse_fwd and self.fuse_3x3:
                        xin, isc, ish, yside = fold_args(j)
                        call('fte_gconv3x3_bn_fwd_bf16_s16', xin, pk[0], T[out], *bn_args(self.fuse_fwd[j]), isc, ish, yside, n, ih, iw, c, op[4],
                             self.ws, self.ws_bytes, st)
                        stats_done.add(self.fuse_fwd[j])
                    else:
                        call('fte_gconv3x3_bf16_s16' if s16 else 'fte_gconv3x3_bf16', T[op[2]], pk[0], T[out], n, ih, iw, c, op[4], 0, st)
                else:
                    call('fte_gconv3x3_fwd', T[op[2]], self.view(op[3]), T[out], n, ih, iw, c, op[5], op[4], st)
            elif kind == 'se':
                inp = op[2]
                w1, b1, w2, b2, hd = self._se_names(op)
                ih, iw, c = self.shapes[inp]
                sq, hid, gate = T[out + '/sq'], T[out + '/hid'], T[out + '/gate']
                call('fte_gap_fwd_s16' if s16 else 'fte_gap_fwd', T[inp], sq, n, ih * iw, c, st)
                if os.environ.get('FTE_SE_ACT_FUSE', '1') != '0':
                    # the gate's two dense layers with their activation (ReLU, sigmoid) in the same pass over the output
                    call('fte_gemm_nn_act', sq, self.view(w1), self.view(b1), hid, n, hd, c, 1, self.ws, self.ws_bytes, st)
                    call('fte_gemm_nn_act', hid, self.view(w2), self.view(b2), gate, n, c, hd, 2, self.ws, self.ws_bytes, st)
                else:                                          # (A/B hook: the activations as launches of their own)
                    call('fte_gemm_nn', sq, self.view(w1), self.view(b1), hid, n, hd, c, self.ws, self.ws_bytes, st)
                    call('fte_act_fwd', hid, hid, hid.numel(), 0, st)
                    call('fte_gemm_nn', hid, self.view(w2), self.view(b2), gate, n, c, hd, self.ws, self.ws_bytes, st)
                    call('fte_act_fwd', gate, gate, gate.numel(), 1, st)
                call('fte_channel_scale_fwd_s16' if s16 else 'fte_channel_scale_fwd', T[inp], gate, T[out], n, ih * iw, c, st)
            elif kind == 'seblock':
                _, _, zin, pre, scn, seop, _, _ = op
                w1, b1, w2, b2, hd = self._se_names(seop)
                ih, iw, c = self.shapes[out]
                hw = ih * iw
                b = self.bn[out]
                fl = ((1 if zin in h16 else 0) | 2) if s16 else 0
                if j in stats_done:                            # the producing conv's epilogue left the batch statistics
                    pass
                elif is_training:
                    upd = self.update_moving_stats
                    args = (T[zin], self.view(pre + '/gamma'), self.view(pre + '/beta'), b['mean'], b['rstd'], b['scale'], b['shift'],
                            self.state[pre + '/moving_mean'] if upd else None, self.state[pre + '/moving_variance'] if upd else None,
                            n * hw, c, BN_EPS, BN_DECAY)
                    if s16:
                        call('fte_bn_train_stats_s16', *args, 1 if zin in h16 else 0, self.ws, self.ws_bytes, st)
                    else:
                        call('fte_bn_train_stats', *args, self.ws, self.ws_bytes, st)
                else:
                    call('fte_bn_infer_coef', self.view(pre + '/gamma'), self.view(pre + '/beta'),
                         self.state[pre + '/moving_mean'], self.state[pre + '/moving_variance'], b['scale'], b['shift'], c, BN_EPS, st)
                sq, hid, gate = T[out + '/sq'], T[out + '/hid'], T[out + '/gate']
                call('fte_se_squeeze', T[zin], b['scale'], b['shift'], b['mean'], b['rstd'], sq, T[out + '/xm'] if is_training else None,
                     n, hw, c, fl & 1, st)
                if self._se_small(c, hd):                      # the gate's dense layers, one launch each
                    call('fte_dense_small', sq, self.view(w1), self.view(b1), None, hid, n, hd, c, 0, 1, st)
                    call('fte_dense_small', hid, self.view(w2), self.view(b2), None, gate, n, c, hd, 0, 2, st)
                else:
                    call('fte_gemm_nn_act', sq, self.view(w1), self.view(b1), hid, n, hd, c, 1, self.ws, self.ws_bytes, st)
                    call('fte_gemm_nn_act', hid, self.view(w2), self.view(b2), gate, n, c, hd, 2, self.ws, self.ws_bytes, st)
                if scn in sc_ev:
                    main_s.wait_event(sc_ev.pop(scn))          # the shortcut branch (side stream) has written it
                call('fte_se_apply_fwd', T[zin], b['scale'], b['shift'], gate, T[scn], T[out], n, hw, c, fl, st)
            elif kind == 'addrelu':
                c = self.shapes[out][-1]
                one, zero = self.ident[c]
                for nm in (op[2], op[3]):
                    if nm in sc_ev:
                        main_s.wait_event(sc_ev.pop(nm))
                if s16:
                    assert op[2] in h16 and op[3] in h16
                    call('fte_bn_infer_fwd_s16', T[op[2]], one, zero, zero, one, T[op[3]], T[out], self._scr(c, 0), self._scr(c, 1),
                         T[out].numel() // c, c, 0.0, 1, 3, st)
                else:
                    call('fte_bn_infer_fwd', T[op[2]], one, zero, zero, one, T[op[3]], T[out], self._scr(c, 0), self._scr(c, 1),
                         T[out].numel() // c, c, 0.0, 1, st)
            elif kind == 'maxpool':
                ih, iw, c = self.shapes[op[2]]
                call('fte_maxpool3x3s2_fwd_s16' if s16 else 'fte_maxpool3x3s2_fwd', T[op[2]], T[out], T[out + '/idx'], n, ih, iw, c, st)
            elif kind == 'gap':
                ih, iw, c = self.shapes[op[2]]
                call('fte_gap_fwd_s16' if op[2] in h16 else 'fte_gap_fwd', T[op[2]], T[out], n, ih * iw, c, st)
            elif kind == 'dropout':
                if is_training:
                    seed = (self.dropout_seed * 1000003 + self.global_step) & 0x7FFFFFFFFFFFFFFF
                    call('fte_dropout_fwd', T[op[2]], T[out + '/mask'], T[out], T[out].numel(), op[3], seed, st)
                else:
                    T[out].copy_(T[op[2]])
            elif kind == 'fc':
                k = self.shapes[op[2]][0]
                call('fte_gemm_nn', T[op[2]], self.view(op[3]), None, T[out], n, self.cpad, k, self.ws, self.ws_bytes, st)
            else:
                raise RuntimeError('op %s must have been fused away' % kind)

    def _gconv_pack(self, op):
        """(forward, dgrad) packed bf16 filters of a grouped 3x3 that runs on the matrix cores -- bf16 MFMA mode,
        4 / 8 / 16 / 32 channels per group -- else None (fp32 vector kernels)."""
        _, _, inp, wname, stride, groups = op
        c = self.shapes[inp][-1]
        if c % 32 or (c // groups) not in (4, 8, 16, 32) or _lib.get_mfma_dtype() != 'bf16' \
                or os.environ.get('FTE_GCONV_MFMA', '1') == '0':
            return None
        pk = self._gpacks.get(wname)
        if pk is None:
            words = (c // 32) * 9 * 1024
            pk = self._gpacks[wname] = (torch.empty(words, dtype=torch.int16, device=self.device),
                                        torch.empty(words, dtype=torch.int16, device=self.device))
        return pk

    @staticmethod
    def _direct_stem(k, cin, cout):
        """3x3 first conv on 1 / 3 image channels with 32 or 64 stored filters: fte_conv3x3_first_* (K = 9*cin is too short for
        the GEMM path's im2col round trip through HBM)"""
        return k == 3 and cin in (1, 3) and cout in (32, 64) and os.environ.get('FTE_DIRECT_STEM', '1') != '0'

    def _se_names(self, op):
        """('se', out, inp, prefix[, scope1, scope2]) -> weight / bias names of the two FCs and the (padded) hidden width"""
        pre = op[3]
        s1, s2 = (op[4], op[5]) if len(op) > 4 else ('fc1', 'fc2')
        w1 = pre + '/%s/weights' % s1
        return w1, pre + '/%s/biases' % s1, pre + '/%s/weights' % s2, pre + '/%s/biases' % s2, self.ishape[w1][1]

    @staticmethod
    def _se_small(c, hd):
        """the SE gate's dense layers through fte_dense_small (one launch each)?  FTE_SE_DENSE=0: fte_gemm_* (A/B hook)"""
        return c % 128 == 0 and hd % 128 == 0 and os.environ.get('FTE_SE_DENSE', '1') != '0'

    def _scr(self, c, i):
        key = ('scr', c, i)
        if key not in self.ident:
            self.ident[key] = torch.empty(c, dtype=torch.float32, device=self.device)
        return self.ident[key]

    def _ensure_built(self, images, num_classes):
        if not self.built:
            n, h, w, ch = images.shape
            self.build(h, w, ch, num_classes, images.device)
        else:
            assert num_classes == self.num_classes, 'num_classes changed after the variables were created'

    def backbone(self, inputs, is_training=False, reuse=None):
        self._run_forward(inputs, is_training)
        return self.t[self.feature_name]

    def eval_features(self, images):
        """The extractor's output for these nets (evaluate.py): the pooled backbone features in inference mode (BN on the
        moving statistics, no dropout).  The reference's ResNet.forward asserts num_classes even for is_training=False
        (nets/resnet.py:147), so its evaluate.py:63 only ever worked for SphereNet; this is what that call was after."""
        assert self.built, 'build() / restore the variables first'
        return self.backbone(images, is_training=False)

    def forward(self, images, num_classes=None, is_training=True):
        assert num_classes is not None, 'num_classes must be given when is_training=True'   # nets/resnet.py:147
        self._ensure_built(images, num_classes)
        self._run_forward(images, is_training)
        out = {'features': self.t[self.feature_name]}
        if self.has_classifier:
            out['logits'] = self.t['logits'][:, :self.num_classes]
        return out

    # ---- loss -------------------------------------------------------------------------------------------
    def _centers(self):
        if 'centers' not in self.state:          # loss.py:34-35: zeros, non-trainable
            d = self.shapes[self.feature_name][0]
            self.state['centers'] = torch.zeros(self.num_classes, d, dtype=torch.float32, device=self.device)
        return self.state['centers']

    def _reconcile_centers(self, comm, labels, n, d):
        """Opt-in (DataParallel(sync_centers=True)): one `centers` table for all replicas.  The reference creates the table inside
        every tower's variable scope and each tower scatter_subs only ITS shard (loss.py:34-39), so its replicas drift apart
        silently (SURVEY.md 8e caveat).  Here every rank has evaluated loss and gradient against the OLD table without touching it;
        the ranks all-gather their (labels, f - c_y) rows -- n x (d + 1) words per rank, the sparse rows only, never the table --
        and each applies ALL rows in rank order with the deterministic scatter kernel: tables stay bit-identical across replicas and
        equal the single-tower update of the GLOBAL batch."""
        world = comm.world_size()
        diff = self.ws[:n * d]
        all_diff = torch.empty(world * n * d, dtype=torch.float32, device=self.device)
        all_lab = torch.empty(world * n, dtype=torch.int32, device=self.device)
        comm.all_gather(all_diff, diff)
        comm.all_gather(all_lab, labels)
        _lib.call('fte_center_scatter_update', all_diff, all_lab, self._centers(), world * n, d, self.num_classes,
                  self.center_alpha, _stream())

    def loss_function(self, scope, labels, **logits):
        """nets/resnet.py:163-176 + Network._regularize; the center / triplet terms are loss.py's functions wired to
        the pooled features (the reference leaves that wiring to the caller)."""
        if not (isinstance(labels, torch.Tensor) and labels.is_cuda and labels.dtype == torch.int32):
            raise TypeError('labels must be an int32 CUDA tensor (data.py:259)')
        labels = labels.contiguous()
        n = labels.shape[0]
        st = _stream()
        call = _lib.call
        slots = self.loss_slots
        feat = self.t[self.feature_name]
        d = feat.shape[1]
        losses, names = [], []
        self._dfeat = None
        if self.head == 'triplet':
            soft = self.triplet_margin is None                                  # loss.py:74-77: None -> softplus, any number -> hinge
            call('fte_batch_hard_triplet_fwd_bwd', feat, labels, 0.0 if soft else float(self.triplet_margin), int(soft), self.tower_scale / n, self.loss_rows, self.dfeat,
                 n, d, self.ws, self.ws_bytes, st)
            call('fte_sum', self.loss_rows, n, self.tower_scale / n, slots[0:1], self.ws, self.ws_bytes, st)
            self._dfeat = self.dfeat
            losses.append(slots[0]); names.append('triplet_loss')
        else:
            if self.head == 'focal':                         # loss.py:18-27 on the classifier logits
                call('fte_focal_loss_fwd_bwd', self.t['logits'], labels, self.loss_rows, self.G, n,
                     self.num_classes, self.cpad, self.focal_gamma, self.focal_alpha, self.tower_scale / n, st)
            else:
                call('fte_softmax_ce_fwd_bwd', self.t['logits'], labels, self.loss_rows, self.G, n,
                     self.num_classes, self.cpad, self.tower_scale / n, st)
            call('fte_sum', self.loss_rows, n, self.tower_scale / n, slots[0:1], self.ws, self.ws_bytes, st)
            losses.append(slots[0]); names.append('focal_entropy' if self.head == 'focal' else 'cross_entropy')
            if self.head == 'softmax+center':
                comm = self.center_comm if self.update_centers else None
                call('fte_center_loss_fwd_bwd_update', feat, labels, self._centers(), self.loss_rows, self.dfeat, n, d, self.num_classes,
                     self.center_alpha if (self.update_centers and comm is None) else 1.0,      # alpha = 1: loss + gradient only
                     self.center_weight * self.tower_scale / (n * d), self.ws, self.ws_bytes, st)
                if comm is not None:
                    self._reconcile_centers(comm, labels, n, d)
                call('fte_sum', self.loss_rows, n, self.tower_scale / (n * d), slots[2:3], self.ws, self.ws_bytes, st)
                self._dfeat = self.dfeat
                losses.append(slots[2]); names.append('center_loss')
        if getattr(self, '_reg_ev', None) is not None and self._reg_scale == 0.5 * self.weight_decay * self.tower_scale:
            torch.cuda.current_stream().wait_event(self._reg_ev)           # taken on the side stream at the start of this forward pass
            self._reg_ev = None
            call('fte_axpby', 1.0, self._reg_scratch[0:1], 0.0, self._reg_scratch[0:1], slots[1:2], 1, st)
        else:
            self._reg_ev = None
            nreg = self.arena_size - self.small_end
            call('fte_sumsq', self.params[self.small_end:], nreg, 0.5 * self.weight_decay * self.tower_scale,
                 slots[1:2], self.ws, self.ws_bytes, st)
        losses.append(slots[1]); names.append('reg_loss')
        return losses, names, OrderedDict()

    # ---- backward ---------------------------------------------------------------------------------------
    def backward(self):
        if self.has_classifier:
            self.backward_head(join=False)               # (backward_body joins the side stream when it returns)
        self.backward_body()

    def backward_stages(self):
        """One callable per all-reduce bucket of grad_buckets(), in the order the backward walk completes them: the classifier, then
        the body in segments from the last layers to the first (data_parallel.py:88-113 issues one nccl.all_sum per variable, which TF
        schedules as each gradient becomes ready; here every segment's filter gradients are final -- side stream joined -- when its
        callable returns, and DataParallel enqueues that bucket's all-reduce while the next segment still runs)."""
        segs = self._segments()
        body = [(lambda lo=lo, hi=hi: self.backward_body(lo, hi)) for lo, hi, _, _ in reversed(segs)]
        return ([self.backward_head] if self.has_classifier else []) + body

    def _op_weight_names(self, op):
        if op[0] in ('conv', 'gconv', 'dwconv'):
            return [op[3]]
        if op[0] in ('se', 'seblock'):
            w1, _, w2, _, _ = self._se_names(op[5] if op[0] == 'seblock' else op)
            return [w1, w2]
        return []

    def _segments(self):
        """[(plan lo, plan hi, arena a, arena b)] in forward order: the body's plan split into FTE_GRAD_BUCKETS (default 4) runs of
        about equal filter bytes.  The filters lie in the arena in the order the plan uses them, so a run of ops owns a contiguous
        arena range; the first segment's range starts at 0 and so carries gamma / beta / biases of the whole net (0.1 - 0.4 MB: final
        early, reduced last, at no cost).  A net whose filters are not in plan order keeps ONE body segment."""
        if getattr(self, '_segs', None) is not None:
            return self._segs
        nops = len(self.plan) - (1 if self.has_classifier else 0)
        one = [(0, nops, 0, self.cls_start)]
        want = int(os.environ.get('FTE_GRAD_BUCKETS', '4'))
        offs = []                                        # (plan index, first arena offset, end offset) of every op with filters
        owned = set()
        for j in range(nops):
            names = self._op_weight_names(self.plan[j])
            if names:
                vs = [self.variables[w] for w in names]
                owned.update(names)
                offs.append((j, min(v.offset for v in vs), max(v.offset + v.size for v in vs)))
        mono = all(offs[i][2] <= offs[i + 1][1] for i in range(len(offs) - 1)) and (not offs or offs[0][1] >= self.small_end)
        # every variable of the body range that SOME plan op names must belong to an op seen above: a bucket's all-reduce is issued when
        # the ops of its plan range have been walked, so a variable of another kind of op (none today; e.g. a mid-plan fc) could land in a
        # bucket reduced before its gradient is final.  (Variables no op names -- ShuffleNet-v2-large's dead convs -- have no gradient.)
        named = {x for op in self.plan[:nops] for x in op if isinstance(x, str) and x in self.variables}
        stray = [k for k in named if self.small_end <= self.variables[k].offset < self.cls_start and k not in owned]
        if want <= 1 or not mono or stray or len(offs) < want:
            self._segs = one
            return one
        total = offs[-1][2] - offs[0][1]
        cuts, acc, k = [], 0, 1                          # cut BEFORE the op at which the running size passes k / want of the total
        for i, (j, a, b) in enumerate(offs):
            if k < want and i > 0 and acc >= total * k / want:
                cuts.append((j, a))
                k += 1
            acc += b - a
        segs, lo, a0 = [], 0, 0
        for j, a in cuts:
            segs.append((lo, j, a0, a))
            lo, a0 = j, a
        segs.append((lo, nops, a0, self.cls_start))
        self._segs = segs
        return segs

    def backward_head(self, join=True):
        """Classifier gradient (first all-reduce bucket) and the gradient wrt its input.  The filter gradient goes to the side stream
        like every other one (nothing but the optimizer / the bucket's all-reduce reads it); `join`: the main stream waits for it before
        this returns (backward_stages: the bucket is reduced next)."""
        n = self._act_n
        st = _stream()
        op = self.plan[-1]
        k = self.shapes[op[2]][0]
        self._grad = {}
        gin = torch.empty(n, k, dtype=torch.float32, device=self.device)
        side = self.side if os.environ.get('FTE_HEAD_SIDE', '1') != '0' else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_event(main.record_event())         # G (the loss head's gradient) and the features are complete
            _lib.call('fte_gemm_tn', self.t[op[2]], self.G, self.view(op[3], self.grads), n, self.cpad, k, self.ws_side, self.ws_bytes, side.cuda_stream)
        else:
            _lib.call('fte_gemm_tn', self.t[op[2]], self.G, self.view(op[3], self.grads), n, self.cpad, k, self.ws, self.ws_bytes, st)
        _lib.call('fte_gemm_nt', self.G, self.view(op[3]), None, None, 0, None, gin, None, n, self.cpad, k, self.ws, self.ws_bytes, st)
        if side is not None and join:
            torch.cuda.current_stream().wait_stream(side)
        self._grad[op[2]] = gin

    def _new(self, name):
        """a gradient buffer for tensor `name`: bf16 where the tensor (or, for a BN output folded into a gather, its gradient) is"""
        if name in self.folded:                  # never stored: no tensor to take the layout from
            z = self.t[self.folded[name][0]]
            return torch.empty(z.shape, dtype=torch.int16 if name in self.g16 else torch.float32, device=self.device)
        return torch.empty_like(self.t[name])

    def backward_body(self, lo=0, hi=None):
        """The backward walk over plan ops [lo, hi) (default: the whole body), last op first.  Segments are walked from the end of
        the plan: the gradients in flight between two calls stay in self._grad; when a call returns its filter gradients are final."""
        n = self._act_n
        st = _stream()
        call = _lib.call
        T = self.t
        s16 = self._act_s16
        h16 = self.h16
        nops = len(self.plan) - (1 if self.has_classifier else 0)
        hi = nops if hi is None else hi
        if hi == nops and not self.has_classifier:
            self._grad = {self.feature_name: self._dfeat}
            self._dfeat = None
        G = self._grad
        ops = self.plan[lo:hi]
        # Filter gradients (conv / depthwise / grouped wgrad + their slab reductions) depend only on the layer's dz and its
        # stored input, and nothing but the optimizer reads them: they go to a second stream and overlap the dgrad -> BN
        # backward chain (these nets' kernels are 5-60 us long and leave CUs idle at their ramps and tails).  An event on
        # the main stream costs it a ~7 us bubble, so the launches are queued and released a few layers at a time.
        main, side = torch.cuda.current_stream(), self.side
        wst, wws = (side.cuda_stream, self.ws_side) if side is not None else (st, self.ws)
        pending = []
        if hi == nops:
            self._reduced = set()
        reduced = self._reduced  # BN outputs whose mask / reduction pass ran in the epilogue of the data gradient that produced G[name]

        def bn_below(name):
            """arguments of the BN layer whose output `name` a fused data gradient lands on, or None"""
            bj = self.fuse_bwd.get(name)
            if bj is None:
                return None
            _, bout, binp, pre, res, relu = self.plan[bj]
            if s16 and not (binp in h16 and bout in h16):
                return None
            b = self.bn[bout]
            zmask = relu and res is None
            return (T[binp], T[bout] if res is not None else None, self.view(pre + '/gamma'), b['mean'], b['rstd'],
                    b['scale'] if zmask else None, b['shift'] if zmask else None), \
                   (self.view(pre + '/gamma', self.grads), self.view(pre + '/beta', self.grads), b['coef'])

        def wgrad(name, dy, *args):
            if side is None:
                call(name, *args)
            else:
                pending.append((name, dy, args))

        def flush(limit=0):
            if len(pending) > limit:
                side.wait_event(main.record_event())
                for name, dy, args in pending:
                    dy.record_stream(side)
                    call(name, *args)
                del pending[:]
        for op in reversed(ops):
            kind, out = op[0], op[1]
            if out == self.feature_name and self._dfeat is not None and out in G:
                # the pooled features also feed the center loss: add its gradient to the classifier path's
                d = G[out].shape[1]
                call('fte_add_scaled_rows_cols', G[out], self._dfeat, self.ones_n, None, n, d, d, st)
                self._dfeat = None
            if kind == 'gather':
                ga, gb = op[2]['gouts']
                if ga not in G:
                    continue
                da = G.pop(ga)
                db = G.pop(gb) if gb else None
                gs = [(name, table, torch.empty((n,) + self.shapes[name], dtype=torch.int16 if s16 else torch.float32, device=self.device))
                      for name, table in op[2]['bwd']]
                (n0, t0, g0), (n1, t1, g1) = gs[0], (gs[1] if len(gs) > 1 else (None, None, None))
                co0 = self.shapes[n0][-1]
                if g1 is None:
                    call('fte_channel_gather_s16' if s16 else 'fte_channel_gather', da, db, g0, t0, g0.numel() // co0, da.shape[-1],
                         db.shape[-1] if db is not None else 0, co0, st)
                else:                                    # the gradients of both sources in one launch
                    call('fte_channel_gather_affine_s16' if s16 else 'fte_channel_gather_affine', da, db, g0, t0, co0, g1, t1, self.shapes[n1][-1], g0.numel() // co0,
                         da.shape[-1], db.shape[-1] if db is not None else 0, None, None, 0, None, None, 0, st)
                for name, _, g in gs:
                    self._put(name, g)
                continue
            if out not in G:
                continue
            dy = G.pop(out)
            if kind == 'dwconv':
                _, _, inp, wname, stride = op
                ih, iw, c = self.shapes[inp]
                wgrad('fte_dwconv3x3_wgrad_s16' if s16 else 'fte_dwconv3x3_wgrad', dy, T[inp], dy, self.view(wname, self.grads), n, ih, iw, c, stride, wws, self.ws_bytes, wst)
                dx = self._new(inp)
                call('fte_dwconv3x3_dgrad_s16' if s16 else 'fte_dwconv3x3_dgrad', dy, self.view(wname), dx, n, ih, iw, c, stride, st)
                self._put(inp, dx)
            elif kind == 'dropout':
                g = self._new(op[2])
                call('fte_dropout_bwd', dy, T[out + '/mask'], g, dy.numel(), op[3], st)
                self._put(op[2], g)
            elif kind == 'gap':
                ih, iw, c = self.shapes[op[2]]
                g = self._new(op[2])
                call('fte_gap_bwd_s16' if op[2] in h16 else 'fte_gap_bwd', dy, g, n, ih * iw, c, st)
                self._put(op[2], g)
            elif kind == 'maxpool':
                ih, iw, c = self.shapes[op[2]]
                g = self._new(op[2])
                call('fte_maxpool3x3s2_bwd_s16' if s16 else 'fte_maxpool3x3s2_bwd', dy, T[out + '/idx'], g, n, ih, iw, c, st)
                self._put(op[2], g)
            elif kind == 'addrelu':
                g = self._new(out)
                call('fte_relu_bwd_s16' if s16 else 'fte_relu_bwd', dy, T[out], g, dy.numel(), st)
                self._put(op[2], g)
                self._put(op[3], g)                      # both addends see the same (read-only) gradient
            elif kind == 'seblock':
                _, _, zin, pre, scn, seop, _, _ = op
                w1, b1, w2, b2, hd = self._se_names(seop)
                ih, iw, c = self.shapes[out]
                hw = ih * iw
                b = self.bn[out]
                fl = ((1 if zin in h16 else 0) | 2) if s16 else 0
                sq, hid, gate = T[out + '/sq'], T[out + '/hid'], T[out + '/gate']
                s1, s2, xm = T[out + '/s1'], T[out + '/s2'], T[out + '/xm']
                dgate, dhid, dsq = self.ident[('se', out, 'dgate')], self.ident[('se', out, 'dhid')], self.ident[('se', 'dsq', c)]
                gam, bet = self.view(pre + '/gamma'), self.view(pre + '/beta')
                g = self._new(out)                             # dy * (out > 0): the shortcut's gradient, and the gate path's input
                call('fte_se_bwd_gate', dy, T[out], T[zin], gam, bet, b['mean'], b['rstd'], gate, g, s1, s2, dgate, n, hw, c, fl, st)
                wgrad('fte_gemm_tn', dgate, hid, dgate, self.view(w2, self.grads), n, c, hd, wws, self.ws_bytes, wst)
                wgrad('fte_reduce_rows', dgate, dgate, self.view(b2, self.grads), None, 1, n, c, 1, 1.0, wst)
                small = self._se_small(c, hd)
                if small:                                      # d(pre-ReLU) = (dgate W2^T) * (hid > 0) in one launch
                    call('fte_dense_small', dgate, self.view(w2), None, hid, dhid, n, hd, c, 1, 0, st)
                else:
                    call('fte_gemm_nt', dgate, self.view(w2), None, None, 0, None, dhid, None, n, c, hd, self.ws, self.ws_bytes, st)
                    call('fte_act_bwd', dhid, hid, dhid, dhid.numel(), 0, st)                        # -> d(pre-ReLU)
                wgrad('fte_gemm_tn', dhid, sq, dhid, self.view(w1, self.grads), n, hd, c, wws, self.ws_bytes, wst)
                wgrad('fte_reduce_rows', dhid, dhid, self.view(b1, self.grads), None, 1, n, hd, 1, 1.0, wst)
                if small:
                    call('fte_dense_small', dhid, self.view(w1), None, None, dsq, n, c, hd, 1, 0, st)
                else:
                    call('fte_gemm_nt', dhid, self.view(w1), None, None, 0, None, dsq, None, n, hd, c, self.ws, self.ws_bytes, st)
                call('fte_se_bn_bwd_coef', s1, s2, gate, dsq, xm, gam, b['mean'], b['rstd'], self.view(pre + '/gamma', self.grads),
                     self.view(pre + '/beta', self.grads), b['coef'], n, hw, c, st)
                dz = torch.empty_like(T[zin])
                call('fte_se_bn_bwd_apply', g, T[zin], b['coef'], gate, dsq, dz, n, hw, c, fl, st)
                self._put(scn, g)
                self._put(zin, dz)
            elif kind == 'se':
                inp = op[2]
                w1, b1, w2, b2, hd = self._se_names(op)
                ih, iw, c = self.shapes[inp]
                hw = ih * iw
                sq, hid, gate = T[out + '/sq'], T[out + '/hid'], T[out + '/gate']
                f32 = dict(dtype=torch.float32, device=self.device)
                dx = self._new(inp)
                # scratch preallocated in _alloc (three allocator calls per SE block and step otherwise)
                dgate, dhid, dsq = self.ident[('se', out, 'dgate')], self.ident[('se', out, 'dhid')], self.ident[('se', 'dsq', c)]
                if s16:          # reduction only; dx is written once by the apply pass below
                    call('fte_channel_scale_bwd_s16', dy, T[inp], gate, dgate, n, hw, c, 1, st)
                else:
                    call('fte_channel_scale_bwd', dy, T[inp], gate, dx, dgate, n, hw, c, 1, st)         # dgate = d(pre-sigmoid)
                # the gate's four parameter gradients feed nothing but the optimizer: side stream, like every filter gradient
                wgrad('fte_gemm_tn', dgate, hid, dgate, self.view(w2, self.grads), n, c, hd, wws, self.ws_bytes, wst)
                wgrad('fte_reduce_rows', dgate, dgate, self.view(b2, self.grads), None, 1, n, c, 1, 1.0, wst)
                call('fte_gemm_nt', dgate, self.view(w2), None, None, 0, None, dhid, None, n, c, hd, self.ws, self.ws_bytes, st)
                call('fte_act_bwd', dhid, hid, dhid, dhid.numel(), 0, st)                            # -> d(pre-ReLU)
                wgrad('fte_gemm_tn', dhid, sq, dhid, self.view(w1, self.grads), n, hd, c, wws, self.ws_bytes, wst)
                wgrad('fte_reduce_rows', dhid, dhid, self.view(b1, self.grads), None, 1, n, hd, 1, 1.0, wst)
                call('fte_gemm_nt', dhid, self.view(w1), None, None, 0, None, dsq, None, n, hd, c, self.ws, self.ws_bytes, st)
                if s16:
                    call('fte_channel_scale_bwd_apply_s16', dy, gate, dsq, dx, n, hw, c, 1.0 / hw, st)
                else:
                    call('fte_bcast_add', dx, dsq, n, hw, c, 1.0 / hw, st)
                self._put(inp, dx)
            elif kind == 'gconv':
                _, _, inp, wname, stride, groups = op
                ih, iw, c = self.shapes[inp]
                if self._gconv_pack(op) is not None:     # bf16 MFMA mode
                    wgrad('fte_gconv3x3_wgrad_bf16_s16' if s16 else 'fte_gconv3x3_wgrad_bf16', dy, T[inp], dy, self.view(wname, self.grads), n, ih, iw, c, groups, stride, wws, self.ws_bytes, wst)
                else:
                    wgrad('fte_gconv3x3_wgrad', dy, T[inp], dy, self.view(wname, self.grads), n, ih, iw, c, groups, stride, wws, self.ws_bytes, wst)
                dx = self._new(inp)
                pk = self._gconv_pack(op)
                bnb = bn_below(inp) if (pk is not None and s16 and self.fuse_bwd_gconv) else None
                if bnb is not None:                      # ... with the mask / sums of the BN layer below in the epilogue
                    (zbn, _, gam, mean, rstd, sc, sh), outs = bnb
                    call('fte_gconv3x3_dgrad_bn_bf16_s16', dy, pk[1], zbn, gam, mean, rstd, sc, sh, dx, *outs, n, ih, iw, c, stride,
                         self.ws, self.ws_bytes, st)
                    reduced.add(inp)
                elif pk is not None:                     # packed by this step's forward pass (the weights have not changed since)
                    call('fte_gconv3x3_bf16_s16' if s16 else 'fte_gconv3x3_bf16', dy, pk[1], dx, n, ih, iw, c, stride, 1, st)
                else:
                    call('fte_gconv3x3_dgrad', dy, self.view(wname), dx, n, ih, iw, c, groups, stride, st)
                self._put(inp, dx)
            elif kind in ('bn', 'bnstats'):
                _, _, inp, pre, res, relu = op
                b = self.bn[out]
                c = self.shapes[out][-1]
                rows = dy.numel() // c
                dz = torch.empty_like(T[inp])
                if out in reduced:               # dy is the masked gradient already, dgamma / dbeta / coef are there: the apply pass alone
                    call('fte_bn_bwd_apply', dy, T[inp], b['coef'], dz, rows, c, 3 if s16 else 0, st)
                    if res is not None:
                        self._put(res, dy)               # the shortcut sees the same (read-only) masked gradient
                elif s16:
                    fl = (1 if inp in h16 else 0) | 2
                    gam, dgam, dbet = self.view(pre + '/gamma'), self.view(pre + '/gamma', self.grads), self.view(pre + '/beta', self.grads)
                    if res is not None:
                        g = self._new(out)
                        call('fte_bn_train_bwd_s16', dy, T[out], T[inp], gam, b['mean'], b['rstd'], None, None, g, dz, dgam, dbet,
                             rows, c, fl, self.ws, self.ws_bytes, st)
                        self._put(res, g)
                    elif relu:
                        call('fte_bn_train_bwd_s16', dy, None, T[inp], gam, b['mean'], b['rstd'], b['scale'], b['shift'], None, dz, dgam, dbet,
                             rows, c, fl, self.ws, self.ws_bytes, st)
                    else:
                        call('fte_bn_train_bwd_s16', dy, None, T[inp], gam, b['mean'], b['rstd'], None, None, None, dz, dgam, dbet,
                             rows, c, fl, self.ws, self.ws_bytes, st)
                elif res is not None:                      # the shortcut gets g = dy * (out > 0): a by-product of the reduce pass
                    g = self._new(out)
                    call('fte_bn_train_bwd_res', dy, T[out], T[inp], self.view(pre + '/gamma'), b['mean'], b['rstd'], g, dz,
                         self.view(pre + '/gamma', self.grads), self.view(pre + '/beta', self.grads), rows, c,
                         self.ws, self.ws_bytes, st)
                    self._put(res, g)
                elif relu:                                 # ReLU mask recomputed from z: the output is not read
                    call('fte_bn_train_bwd_zmask', dy, T[inp], self.view(pre + '/gamma'), b['mean'], b['rstd'], b['scale'], b['shift'],
                         dz, self.view(pre + '/gamma', self.grads), self.view(pre + '/beta', self.grads), rows, c,
                         self.ws, self.ws_bytes, st)
                else:
                    call('fte_bn_train_bwd', dy, None, T[inp], self.view(pre + '/gamma'), b['mean'], b['rstd'], dz,
                         self.view(pre + '/gamma', self.grads), self.view(pre + '/beta', self.grads), rows, c,
                         self.ws, self.ws_bytes, st)
                self._put(inp, dz)
            elif kind == 'conv':
                _, _, inp, wname, stride = op
                ih, iw, cin = self.shapes[inp]
                k = self.spec[wname][0][0]
                cout = self.shapes[out][-1]
                gw = self.view(wname, self.grads)
                if cin < 32 and self._direct_stem(k, cin, cout):
                    call('fte_conv3x3_first_wgrad_s16' if s16 else 'fte_conv3x3_first_wgrad', T[inp], dy, gw, n, ih, iw, cin, cout, stride, self.ws, self.ws_bytes, st)
                    continue
                if cin < 32:                             # stem: filter gradient only
                    oh, ow, _ = self.shapes[out]
                    if s16:
                        call('fte_conv2d_wgrad16', self.cols, dy, gw, n, oh, ow, stem_kpad(k, cin), cout, 1, 1, self.ws, self.ws_bytes, st)
                    else:
                        call('fte_gemm_tn', self.cols, dy, gw, n * oh * ow, cout, stem_kpad(k, cin), self.ws, self.ws_bytes, st)
                    continue
                wgrad('fte_conv2d_wgrad16' if s16 else 'fte_conv2d_wgrad', dy, T[inp], dy, gw, n, ih, iw, cin, cout, k, stride, wws, self.ws_bytes, wst)
                flush(self.side_batch)
                prev = G.pop(inp, None)                  # accumulate into an existing contribution through `addin`
                dx = self._new(inp)
                bnb = bn_below(inp) if self.fuse_bwd_conv else None
                if bnb is not None:          # the last contribution to the gradient of a BN output: mask + BN sums in the epilogue
                    ins, outs = bnb
                    call('fte_conv2d_dgrad_bn', dy, self.w16[wname] if s16 else self.view(wname), prev, *ins, dx, *outs,
                         n, ih, iw, cin, cout, k, stride, 1 if s16 else 0, self.ws, self.ws_bytes, st)
                    reduced.add(inp)
                elif s16:          # bf16 dz in, bf16 dx out (+ the bf16 contribution already there); the HWIO pack is this step's
                    call('fte_conv2d_dgrad_s16', dy, self.w16[wname], prev, None, None, None, dx, None, None,
                         n, ih, iw, cin, cout, k, stride, self.ws, self.ws_bytes, st)
                else:
                    call('fte_conv2d_dgrad', dy, self.view(wname), prev, None, None, None, dx, None, None,
                         n, ih, iw, cin, cout, k, stride, self.ws, self.ws_bytes, st)
                G[inp] = dx
            else:
                raise RuntimeError(kind)
        if side is not None:
            flush()
            main.wait_stream(side)
        if lo == 0:
            self._grad = {}

    def _put(self, name, g):
        if name in self._grad:
            raise RuntimeError('gradient of %s already has a contribution that cannot be accumulated in place' % name)
        self._grad[name] = g

    # ---- bookkeeping the wrappers use ---------------------------------------------------------------------
    def param_list(self, is_training, trainable, scope=None):
        """nets/resnet.py:178-184: trainable=True -> tf.trainable_variables(scope), trainable=False ->
        tf.global_variables(scope), which also holds the scope's BatchNorm moving statistics (what the fine-tune
        saver of train.py:191-193 restores through pretrained_param)."""
        bb = [v for k, v in self.variables.items() if k.startswith(self.name + '/')]
        if not trainable:
            bb = bb + [Variable(k, 'state', (self.state_ref.get(k, t.shape[0]),), -1, self.state_ref.get(k, t.shape[0]))
                       for k, t in self.state.items() if k.startswith(self.name + '/')]
        if is_training:
            return [bb, [v for k, v in self.variables.items() if k.startswith('classifier/')]]
        return [bb]

    def pretrained_param(self, scope=None):
        return [v for grp in self.param_list(is_training=False, trainable=False, scope=scope) for v in grp if self.name in v.name]

    def arena_groups(self):
        groups = [(0, self.small_end, False, 0), (self.small_end, self.cls_start, True, 0)]
        if self.has_classifier:
            groups.append((self.cls_start, self.arena_size, True, 1))
        return groups

    def grad_buckets(self):
        """arena ranges of backward_stages()'s callables, in the same order; the four loss slots behind the arena ride on the bucket
        that ends at the arena's end"""
        segs = self._segments()
        body = [(a, b) for _, _, a, b in reversed(segs)]
        if self.has_classifier:
            return [(self.cls_start, self.arena_size + 4)] + body
        body[0] = (body[0][0], self.arena_size + 4)          # (cls_start == arena_size: the last segment ends the arena)
        return body
