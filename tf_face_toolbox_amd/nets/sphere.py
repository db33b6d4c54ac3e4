"""SphereNet-20 ("SphereFaceNet-20") on libfte.so -- host-side mirror of nets/sphere.py.

Reference: nets/sphere.py:23-134 (class SphereNet: prelu :29-36, resBlock :38-45, backbone
:47-76, forward :78-101, loss_function :103-118, param_list :120-126, pretrained_param
:128-134).  Same constructor / method names and argument meaning; the bodies enqueue HIP
kernels instead of building TF graph nodes.

MI355X-first layout decisions (engine-internal; the boundary keeps the reference layouts):
  * activations are NHWC fp32 end to end -- the NHWC->NCHW transpose of nets/sphere.py:53-54
    disappears; with data_format='NCHW' only the ROW ORDER of the 25088x512 FC weight differs
    (flatten order C,H,W vs H,W,C, nets/sphere.py:72) and that is permuted on import/export;
  * all parameters live in ONE flat fp32 arena  [biases+alphas | conv W (HWIO) | FC W | classifier W],
    gradients and optimizer slots in arenas of the same layout: one fused optimizer launch per
    decay group and one (bucketed) RCCL all-reduce instead of 47 (data_parallel.py:179);
  * the classifier is padded to a multiple of 128 columns (zero weights, masked in the loss);
  * every conv keeps z (pre-activation) and y = PReLU(z) (+shortcut): backward needs sign(z)
    and min(z,0) exactly, for any alpha.
"""
import math
import os
from collections import OrderedDict

import torch

from .. import _lib
from .net_base import Network, side_stream

NUM_BLOCKS = (1, 2, 4, 1)     # nets/sphere.py:58,62,66,70
EMBED = 512                   # nets/sphere.py:73


def same_pads(in_size, k, stride):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2, total - total // 2


class _Conv(object):
    __slots__ = ('name', 'stage', 'stride', 'has_bias', 'second', 'cin', 'cout', 'hin', 'win', 'hout', 'wout')


class Variable(object):
    """A named view of the parameter arena, addressed by the reference's TF variable name."""

    def __init__(self, name, kind, ref_shape, offset, size):
        self.name, self.kind, self.ref_shape, self.offset, self.size = name, kind, tuple(ref_shape), offset, size

    def __repr__(self):
        return '<Variable %s %s>' % (self.name, self.ref_shape)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class SphereNet(Network):
    head = 'softmax'

    def __init__(self, weight_decay=0.0005, data_format='NCHW', name='SphereNet', seed=0):
        super(SphereNet, self).__init__(weight_decay, data_format, name)
        self.num_outputs = [64, 128, 256, 512]
        self.seed = seed
        self.built = False
        # in the bf16 MFMA mode the convs read bf16 COPIES of their operands (written by the producing epilogue / packed
        # once per step) through fte_conv2d_*16: bit-identical to the operand mode, half the bytes through the load path
        self.bf16_copies = os.environ.get('FTE_BF16_COPIES', '1') != '0'
        self.tower_scale = 1.0        # 1/num_gpus, set by the parallel wrapper (data_parallel.py:37)
        self.global_step = 0          # A-softmax lambda annealing reads it
        self._act_n = None

    # ------------------------------------------------------------------ construction
    def _conv_specs(self, h, w, cin):
        specs = []
        for si, nb in enumerate(NUM_BLOCKS):
            stage = '%s/conv%d' % (self.name, si + 1)
            lst = [(stage + '/Conv', 2, True, None)]
            for b in range(nb):
                blk = stage + ('/resBlock' if nb == 1 else '/Repeat/resBlock_%d' % (b + 1))
                lst += [(blk + '/Conv', 1, False, 0), (blk + '/Conv_1', 1, False, 1)]
            for nm, stride, has_bias, second in lst:
                c = _Conv()
                c.name, c.stage, c.stride, c.has_bias, c.second = nm, si, stride, has_bias, second
                c.cin, c.cout, c.hin, c.win = cin, self.num_outputs[si], h, w
                c.hout, c.wout = same_pads(h, 3, stride)[0], same_pads(w, 3, stride)[0]
                specs.append(c)
                h, w, cin = c.hout, c.wout, c.cout
        return specs

    def build(self, height, width, channels, num_classes, device='cuda'):
        """Create the variables (TF does this lazily inside the first forward())."""
        _lib.load()
        self.device = torch.device(device)
        self.in_hwc = (height, width, channels)
        self.num_classes = int(num_classes)
        self.cpad = (self.num_classes + 127) // 128 * 128
        self.convs = self._conv_specs(height, width, channels)
        last = self.convs[-1]
        self.feat_hwc = (last.hout, last.wout, last.cout)
        self.fin = last.hout * last.wout * last.cout
        # ---- arena layout: [small | conv W | FC W | classifier W(padded)] ----------------
        small, big = [], []
        for c in self.convs:
            if c.has_bias:
                small.append((c.name + '/biases', 'bias', (c.cout,), c.cout))
            small.append((c.name + '/alpha', 'alpha', (c.cout,), c.cout))
            big.append((c.name + '/weights', 'conv_w', (3, 3, c.cin, c.cout), 9 * c.cin * c.cout))
        small.append((self.name + '/fully_connected/biases', 'fc_b', (EMBED,), EMBED))
        big.append((self.name + '/fully_connected/weights', 'fc_w', (self.fin, EMBED), self.fin * EMBED))
        big.append(('classifier/fc_classifier/weights', 'cls_w', (EMBED, self.num_classes), EMBED * self.cpad))
        self.variables = OrderedDict()
        off = 0
        for nm, kind, shape, size in small + big:
            self.variables[nm] = Variable(nm, kind, shape, off, size)
            off += (size + 3) // 4 * 4
        self.small_end = self.variables[big[0][0]].offset
        self.cls_start = self.variables['classifier/fc_classifier/weights'].offset
        self.fc_start = self.variables[self.name + '/fully_connected/weights'].offset
        self._stage_first = [min(self.variables[c.name + '/weights'].offset for c in self.convs if c.stage == st_)
                             for st_ in sorted(set(c.stage for c in self.convs))]
        self.arena_size = off
        dev = self.device
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off + 4, dtype=torch.float32, device=dev)      # +4: [ce, reg, -, -] loss slots
        self.loss_slots = self.grads[off:off + 4]
        self._init_params()
        self.built = True
        return self

    def view(self, name, arena=None):
        v = self.variables[name]
        a = self.params if arena is None else arena
        return a[v.offset:v.offset + v.size]

    def _init_params(self):
        """Reference initialisers: nets/sphere.py:34 (alpha 0.25), :41-42 (N(0,0.01) resBlock convs),
        :87 (N(0,1e-3) classifier); layers.conv2d / fully_connected defaults = Xavier-uniform, zero bias."""
        g = torch.Generator().manual_seed(self.seed)
        for c in self.convs:
            if c.has_bias:
                lim = (6.0 / (9 * c.cin + 9 * c.cout)) ** 0.5
                w = (torch.rand(3, 3, c.cin, c.cout, generator=g) * 2 - 1) * lim
            else:
                w = torch.randn(3, 3, c.cin, c.cout, generator=g) * 0.01
            self.view(c.name + '/weights').copy_(w.reshape(-1))
            self.view(c.name + '/alpha').fill_(0.25)
        lim = (6.0 / (self.fin + EMBED)) ** 0.5
        w = (torch.rand(self.fin, EMBED, generator=g) * 2 - 1) * lim
        self.view(self.name + '/fully_connected/weights').copy_(w.reshape(-1))
        wc = torch.zeros(EMBED, self.cpad)
        wc[:, :self.num_classes] = torch.randn(EMBED, self.num_classes, generator=g) * 0.001
        self.view('classifier/fc_classifier/weights').copy_(wc.reshape(-1))

    # ---- reference-layout import / export ----------------------------------------------
    def _fc_perm(self, t, to_internal):
        """FC weight rows: reference order is the flatten order of data_format (nets/sphere.py:72)."""
        if self.data_format == 'NHWC':
            return t
        h, w, c = self.feat_hwc
        if to_internal:
            return t.reshape(c, h, w, EMBED).permute(1, 2, 0, 3).reshape(self.fin, EMBED)
        return t.reshape(h, w, c, EMBED).permute(2, 0, 1, 3).reshape(self.fin, EMBED)

    def get_variable(self, name, arena=None):
        """Value of a variable (or of its gradient / slot when `arena` is given) in the REFERENCE layout."""
        v = self.variables[name]
        t = self.view(name, arena)
        if v.kind == 'cls_w':
            return t.reshape(EMBED, self.cpad)[:, :self.num_classes].clone()
        if v.kind == 'fc_w':
            return self._fc_perm(t.reshape(self.fin, EMBED), False).contiguous()
        return t.reshape(v.ref_shape).clone()

    def set_variable(self, name, value, arena=None):
        """Write a reference-layout value into the variable (or into its slot of another arena)."""
        v = self.variables[name]
        t = torch.as_tensor(value, dtype=torch.float32).to(self.device)
        assert tuple(t.shape) == v.ref_shape, (name, tuple(t.shape), v.ref_shape)
        if v.kind == 'cls_w':
            buf = torch.zeros(EMBED, self.cpad, device=self.device)
            buf[:, :self.num_classes] = t
            t = buf
        elif v.kind == 'fc_w':
            t = self._fc_perm(t, True)
        self.view(name, arena).copy_(t.reshape(-1))

    def load_params(self, params):
        for k, val in params.items():
            self.set_variable(k, val)

    # ---- buffers that depend on the batch size -----------------------------------------
    def _storage16(self):
        """bf16 STORAGE ('bf16s', fte.h): z, y, dz and the skip-path gradient live in HBM as bf16 only; the last conv layer's z / y
        (the dense layer's operands, 25088 values per image) are kept in fp32 as well."""
        return _lib.bf16_storage()

    def _alloc_acts(self, n):
        s16 = self._storage16()
        key = (s16, _lib.get_mfma_dtype(), _lib.query('fte_get_conv_algo'))      # what the workspace / V-pack sizes depend on
        if self._act_n == n and getattr(self, '_act_key', None) == key:
            return
        self._act_key = key
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i16 = dict(dtype=torch.int16, device=dev)
        self._act_s16 = s16
        self.y16 = None
        self.z, self.y = [], []
        last = len(self.convs) - 1
        for l, c in enumerate(self.convs):
            keep32 = not s16 or l == last
            self.z.append(torch.empty(n, c.hout, c.wout, c.cout, **f32) if keep32 else None)
            self.y.append(torch.empty(n, c.hout, c.wout, c.cout, **f32) if keep32 else None)
        if s16:
            self.z16 = [torch.empty(n, c.hout, c.wout, c.cout, **i16) for c in self.convs]
        self.emb = torch.empty(n, EMBED, **f32)
        self.s_raw = torch.empty(n, self.cpad, **f32)
        self.G = torch.empty(n, self.cpad, **f32)
        self.logits_buf = torch.empty(n, self.cpad, **f32) if self.head == 'asoftmax' else self.s_raw
        self.loss_rows = torch.empty(n, **f32)
        self.xn = torch.empty(n, **f32)
        self.wn = torch.empty(self.cpad, **f32)
        self.rowcoef = torch.empty(n, **f32)
        self.colcoef = torch.empty(self.cpad, **f32)
        self.demb = torch.empty(n, EMBED, **f32)
        self.bwd = {}
        for si in range(4):
            c = [q for q in self.convs if q.stage == si][0]
            shp = (n, c.hout, c.wout, c.cout)
            if s16:      # bf16 only (dz16 / raw16); the last stage keeps one fp32 dz / raw for the dense layer's backward epilogue
                one = [torch.empty(shp, **f32)] if si == 3 else []
                self.bwd[si] = dict(dz=one, raw=list(one and [torch.empty(shp, **f32)]), dzi=0, rawi=0,
                                    dz16=[torch.empty(shp, **i16) for _ in range(self._dz_buffers())],
                                    raw16=[torch.empty(shp, **i16), torch.empty(shp, **i16)])
            else:
                self.bwd[si] = dict(dz=[torch.empty(shp, **f32) for _ in range(self._dz_buffers())],
                                    raw=[torch.empty(shp, **f32), torch.empty(shp, **f32)], dzi=0, rawi=0)
        need = 4096
        q = _lib.query
        # Winograd layers (fte.h FTE_CONV_*; the reference's train.py:260): the forward pass leaves V = B^T d B of its input for the
        # layer's filter gradient (fte_conv3x3_fwd_keep / fte_conv3x3_wgrad_kept) -- one tile transform instead of two
        self.vpack = [None] * len(self.convs)
        if not s16 and os.environ.get('FTE_WINO_KEEP', '1') != '0':
            for l, c in enumerate(self.convs):
                if l > 0 and c.stride == 1 and q('fte_conv3x3_algo', n, c.hin, c.win, c.cin, c.cout, 1, 0) == 1 \
                        and q('fte_conv3x3_algo', n, c.hin, c.win, c.cin, c.cout, 1, 2) == 1:
                    self.vpack[l] = torch.empty(q('fte_wino_pack_bytes', n, c.hin, c.win, c.cin) // 4, **f32)
        for c in self.convs[1:]:
            need = max(need, q('fte_conv3x3_fwd_ws_bytes', n, c.hin, c.win, c.cin, c.cout, c.stride),
                       q('fte_conv3x3_wgrad_ws_bytes', n, c.hin, c.win, c.cin, c.cout, c.stride),
                       q('fte_conv3x3_dgrad_ws_bytes', n, c.hin, c.win, c.cin, c.cout, c.stride))
        c0 = self.convs[0]
        need = max(need, q('fte_conv3x3_first_wgrad_ws_bytes', n, c0.hin, c0.win, c0.cin, c0.cout, c0.stride),
                   q('fte_gemm_ws_bytes', n, EMBED, self.fin), q('fte_gemm_ws_bytes', n, self.cpad, EMBED))
        self.ws = torch.empty((need + 3) // 4 + 1024, **f32)
        self.ws_bytes = self.ws.numel() * 4
        # filter gradients run on a second stream beside the data gradient of the same layer (_body_walk); their split-K slabs
        # need a workspace of their own
        self.side = side_stream(self.device) if os.environ.get('FTE_SIDE_STREAM', '1') != '0' else None
        self.ws_side = torch.empty_like(self.ws) if self.side is not None else self.ws
        self._act_n = n
        self.y16 = None                                   # allocated on first use (_alloc_copies)

    def _use_copies(self):
        return self.bf16_copies and _lib.get_mfma_dtype() == 'bf16'

    def _alloc_copies(self):
        """bf16 copies of what the conv MFMAs read: y16[l] (written by layer l's forward epilogue), dz16 per stage
        (written by the dgrad epilogue), the weight packs w16 (HWIO, dgrad) and w16t ([tap][cout][cin], forward)."""
        if self.y16 is not None and self.y16[0].shape[0] == self._act_n:
            return
        i16 = dict(dtype=torch.int16, device=self.device)
        n = self._act_n
        self.y16 = [torch.empty(n, c.hout, c.wout, c.cout, **i16) for c in self.convs]
        for si in range(4):
            b = self.bwd[si]
            if 'dz16' not in b:
                b['dz16'] = [torch.empty(b['dz'][0].shape, **i16) for _ in b['dz']]
        if getattr(self, 'packs', None) is None:
            from ._packs import FilterPacks
            self.packs = FilterPacks([(c.name, self.variables[c.name + '/weights'].offset, 3, c.cin, c.cout) for c in self.convs[1:]], self.device)
            self.w16, self.w16t = self.packs.w16, self.packs.w16t

    def _pack_weights(self, st):
        self.packs.refresh(self.params, st)          # all 19 filters, one launch per layout

    # ------------------------------------------------------------------ forward
    def prelu(self, x, name='prelu'):
        raise RuntimeError('PReLU is fused into the conv epilogue (fte_conv3x3_fwd); it is not a separate op here.')

    def _check_images(self, images):
        if not (isinstance(images, torch.Tensor) and images.is_cuda and images.dtype == torch.float32):
            raise TypeError('images must be a float32 CUDA tensor in NHWC (data.py:275-279 layout)')
        if not images.is_contiguous():
            images = images.contiguous()
        return images

    def _fwd_split(self, n, copies):
        """Forward walk as two part shards on two streams: the size of the first part, or 0 (one chain).  fp32 Winograd plan only (the
        direct kernels fill the chip by themselves and have no HBM-bound companion).  The first part must end on a whole 64-tile row
        block of every kept V pack (the second part's pack is the rest of the shard's): the split point is the multiple of that
        granule nearest below the middle (112 x 112: 64 images), and both parts must plan the same algorithm per layer as the whole
        shard.  FTE_FWD_HALVES=0 / 1 forces it off / on where it is possible; FTE_FWD_SPLIT=<images> moves the split point (exploration)."""
        env = os.environ.get('FTE_FWD_HALVES', 'auto')
        if env == '0' or copies or self.side is None or n < 2 or getattr(self, 'one_stream', False):      # (one_stream: bench.py's launch-record steps)
            return 0
        key = (n, getattr(self, '_act_key', None))
        if getattr(self, '_split_key', None) != key:
            q = _lib.query
            gran = 1
            for l, c in enumerate(self.convs):
                if self.vpack[l] is not None:
                    tpi = ((c.hin + 1) // 2) * ((c.win + 1) // 2)
                    gran = max(gran, 64 // math.gcd(64, tpi))      # (powers of two: the largest is their common multiple)
            a = int(os.environ.get('FTE_FWD_SPLIT', (n // 2) // gran * gran))
            ok, any_w = 0 < a < n, False
            for l, c in enumerate(self.convs):
                if l == 0 or c.stride != 1 or not ok:
                    continue
                algo = [q('fte_conv3x3_algo', m, c.hin, c.win, c.cin, c.cout, 1, 0) for m in (n, a, n - a)]
                any_w = any_w or algo[0] == 1
                if algo[1] != algo[0] or algo[2] != algo[0]:
                    ok = False
                if self.vpack[l] is not None and (a * ((c.hin + 1) // 2) * ((c.win + 1) // 2)) % 64:
                    ok = False
            self._split_key, self._split = key, (a if ok and any_w else 0)
        if not self._split:
            return 0
        return self._split if env == '1' or n >= int(os.environ.get('FTE_FWD_HALVES_MIN', '128')) else 0

    def _fwd_halves(self, n, copies):
        return self._fwd_split(n, copies) > 0

    def backbone(self, inputs, is_training=False, reuse=None):
        """nets/sphere.py:47-76: [N,H,W,C] NHWC -> embedding [N,512] (a view of an internal buffer)."""
        x = self._check_images(inputs)
        n, h, w, ch = x.shape
        if not self.built:
            raise RuntimeError('call build() or forward(..., num_classes=) with is_training=True first')
        assert (h, w, ch) == self.in_hwc, ((h, w, ch), self.in_hwc)
        self._alloc_acts(n)
        st = _stream()
        call = _lib.call
        keep = is_training
        self._images = x
        s16 = self._storage16()
        copies = self._use_copies() or s16
        self._copies_live = copies and keep              # backward of THIS forward may use the bf16 copies
        self._s16_live = s16 and keep
        self._vpack_live = keep and not copies           # ... or the V packs the Winograd layers keep
        if copies:
            self._alloc_copies()
            self._pack_weights(st)
        def layer(l, c, lo, hi, ws, st):
            """layer l over images [lo, hi) on stream st (every forward kernel is per image: no statistics, nets/sphere.py:38-45)"""
            m = hi - lo
            wv = self.view(c.name + '/weights')
            bv = self.view(c.name + '/biases') if c.has_bias else None
            av = self.view(c.name + '/alpha')
            zz = self.z[l][lo:hi] if keep and self.z[l] is not None else None      # (bf16 storage keeps fp32 z / y for the last layer only)
            if s16:
                # bf16 storage: every layer writes bf16 z / y only (+ the unrounded fp32 pair for the last layer: the dense layer reads it)
                z16 = self.z16[l][lo:hi] if keep else None
                if l == 0:
                    call('fte_conv3x3_first_fwd_s16', x[lo:hi], wv, bv, av, z16, self.y16[0][lo:hi], m, c.hin, c.win, c.cin, c.cout, c.stride, st)
                else:
                    call('fte_conv2d_fwd_s16', self.y16[l - 1][lo:hi], self.w16t[c.name], bv, av, self.y16[l - 2][lo:hi] if c.second == 1 else None,
                         z16, self.y16[l][lo:hi], zz, self.y[l][lo:hi] if self.y[l] is not None else None, m, c.hin, c.win, c.cin, c.cout, 3, c.stride,
                         ws, self.ws_bytes, st)
            elif l == 0:
                call('fte_conv3x3_first_fwd', x[lo:hi], wv, bv, av, zz, self.y[0][lo:hi], m, c.hin, c.win, c.cin, c.cout, c.stride, st)
                if copies:
                    call('fte_to_bf16', self.y[0][lo:hi], self.y16[0][lo:hi], self.y[0][lo:hi].numel(), st)
            else:
                res = self.y[l - 2][lo:hi] if c.second == 1 else None
                if copies:
                    call('fte_conv2d_fwd16', self.y16[l - 1][lo:hi], self.w16t[c.name], bv, av, res, zz, self.y[l][lo:hi], self.y16[l][lo:hi],
                         m, c.hin, c.win, c.cin, c.cout, 3, c.stride, ws, self.ws_bytes, st)
                elif keep and self.vpack[l] is not None:
                    vp = self.vpack[l]
                    if m != n:                               # a part shard's V pack is its part of the shard's (whole row blocks: _fwd_split)
                        per = ((c.hin + 1) // 2) * ((c.win + 1) // 2) * c.cin * 16      # floats per image
                        vp = vp[lo * per:hi * per] if hi < n else vp[lo * per:]        # (the last part keeps the shard's padding rows)
                    call('fte_conv3x3_fwd_keep', self.y[l - 1][lo:hi], wv, bv, av, res, zz, self.y[l][lo:hi],
                         m, c.hin, c.win, c.cin, c.cout, c.stride, vp, ws, self.ws_bytes, st)
                else:
                    call('fte_conv3x3_fwd', self.y[l - 1][lo:hi], wv, bv, av, res, zz, self.y[l][lo:hi],
                         m, c.hin, c.win, c.cin, c.cout, c.stride, ws, self.ws_bytes, st)

        split = self._fwd_split(n, copies)
        if split:
            # Two part shards (halves), one per stream: layer l of one part runs beside the tile transform of the other (an HBM-bound
            # kernel of 68 registers under an MFMA-bound resident one), and the CUs a launch's last, partly filled round leaves idle go
            # to the other part's launch.  No forward kernel of this net looks across images: each part comes out bit for bit as the
            # net computes it for those images as a shard of their own (against ONE launch over the whole shard the direct kernels of
            # the stride-2 layers may split their K sums differently: ~1e-7 relative; tests/test_gpu_stress.py).
            main, side = torch.cuda.current_stream(), self.side
            side.wait_stream(main)
            for l, c in enumerate(self.convs):
                layer(l, c, 0, split, self.ws, st)
                layer(l, c, split, n, self.ws_side, side.cuda_stream)
            main.wait_stream(side)
        else:
            for l, c in enumerate(self.convs):
                layer(l, c, 0, n, self.ws, st)
        call('fte_gemm_nn', self.y[-1], self.view(self.name + '/fully_connected/weights'),
             self.view(self.name + '/fully_connected/biases'), self.emb, n, EMBED, self.fin, self.ws, self.ws_bytes, st)
        return self.emb

    def _classifier_raw(self, n):
        _lib.call('fte_gemm_nn', self.emb, self.view('classifier/fc_classifier/weights'), None, self.s_raw,
                  n, self.cpad, EMBED, self.ws, self.ws_bytes, _stream())

    def _ensure_built(self, images, num_classes):
        if not self.built:
            n, h, w, ch = images.shape
            self.build(h, w, ch, num_classes, images.device)
        else:
            assert num_classes == self.num_classes, 'num_classes changed after the variables were created'

    def forward(self, images, num_classes=None, is_training=True):
        """nets/sphere.py:78-101."""
        if is_training:
            assert num_classes is not None, 'num_classes must be given when is_training=True'
            self._ensure_built(images, num_classes)
            self.backbone(images, is_training=True)
            self._classifier_raw(images.shape[0])
            return {'logits': self.s_raw[:, :self.num_classes]}
        return self._eval_features(images)

    def _eval_features(self, images):
        # nets/sphere.py:97-101: mean of the embeddings of x and of its horizontal flip (axis 2 of NHWC)
        # (flip and mean run in libfte.so like everything else of the path: fte_flip_width, fte_axpby)
        images = self._check_images(images)
        n, h, w, ch = images.shape
        st = _stream()
        out = torch.empty(n, EMBED, dtype=torch.float32, device=images.device)
        f1 = self.backbone(images, is_training=False)
        _lib.call('fte_axpby', 0.5, f1, 0.0, f1, out, n * EMBED, st)
        flipped = torch.empty_like(images)
        _lib.call('fte_flip_width', images, flipped, n, h, w, ch, st)
        f2 = self.backbone(flipped, is_training=False)
        _lib.call('fte_axpby', 1.0, out, 0.5, f2, out, n * EMBED, st)
        return out

    # ------------------------------------------------------------------ loss
    def _grad_scale(self, n):
        # mean over the shard (tf.losses.sparse_softmax_cross_entropy) times the 1/num_gpus of
        # data_parallel.py:37 -- applied at the head so that it propagates through backward for free.
        return self.tower_scale / n

    def _finish_losses(self, n):
        st = _stream()
        # slots: [ce * tower_scale, reg * tower_scale]  (sum over towers == data_parallel.py:248 mean)
        _lib.call('fte_sum', self.loss_rows, n, self.tower_scale / n, self.loss_slots[0:1], self.ws, self.ws_bytes, st)
        nreg = self.arena_size - self.small_end
        _lib.call('fte_sumsq', self.params[self.small_end:], nreg, 0.5 * self.weight_decay * self.tower_scale,
                  self.loss_slots[1:2], self.ws, self.ws_bytes, st)

    def loss_function(self, scope, labels, **logits):
        """nets/sphere.py:103-118 + Network._regularize (nets/net_base.py:103-116).
        Returns ([cross_entropy, reg_loss], names, others): the losses are 0-d device tensors
        (views of the loss slots that ride on the gradient arena); read them after the step."""
        n = labels.shape[0]
        labels = self._check_labels(labels)
        self._labels = labels
        _lib.call('fte_softmax_ce_fwd_bwd', self.s_raw, labels, self.loss_rows, self.G, n, self.num_classes, self.cpad,
                  self._grad_scale(n), _stream())
        self._finish_losses(n)
        return [self.loss_slots[0], self.loss_slots[1]], ['cross_entropy', 'reg_loss'], OrderedDict()

    @staticmethod
    def _check_labels(labels):
        if not (isinstance(labels, torch.Tensor) and labels.is_cuda and labels.dtype == torch.int32):
            raise TypeError('labels must be an int32 CUDA tensor (data.py:259)')
        return labels.contiguous()

    # ------------------------------------------------------------------ backward (replaces tf.gradients)
    def _head_backward(self, n, st):
        call = _lib.call
        wc = self.view('classifier/fc_classifier/weights')
        gwc = self.view('classifier/fc_classifier/weights', self.grads)
        call('fte_gemm_tn', self.emb, self.G, gwc, n, self.cpad, EMBED, self.ws, self.ws_bytes, st)
        call('fte_gemm_nt', self.G, wc, None, None, 0, None, self.demb, None, n, self.cpad, EMBED, self.ws, self.ws_bytes, st)

    def backward_head(self):
        """Classifier + FC gradients: the first (and largest) all-reduce bucket."""
        n = self._act_n
        st = _stream()
        call = _lib.call
        g = self.grads
        self._head_backward(n, st)
        fcw = self.name + '/fully_connected/weights'
        call('fte_reduce_rows', self.demb, self.view(self.name + '/fully_connected/biases', g), None, 1, n, EMBED, 1, 1.0, st)
        call('fte_gemm_tn', self.y[-1], self.demb, self.view(fcw, g), n, EMBED, self.fin, self.ws, self.ws_bytes, st)

    def backward(self):
        for stage in self.backward_stages():
            stage()

    def backward_stages(self):
        """One callable per all-reduce bucket of grad_buckets(), in the order backward completes them: the head, then the
        conv stack stage by stage from the last one (its filter gradients are final as soon as the walk leaves the
        stage -- 23.6 MB for stage 4, 21 MB for stage 3 -- so they cross xGMI under the remaining backward; only the
        stage-1 bucket with the biases / alphas is reduced after the last kernel)."""
        it = self._body_walk()
        return [self.backward_head] + [lambda it=it: next(it) for _ in range(len(self._stage_first))]

    @staticmethod
    def _dz_buffers():
        # three dz buffers per stage: wgrad(l) may still be reading its dz while dgrad(l - 1) writes the next one (_body_walk)
        return max(2, int(os.environ.get('FTE_DZ_BUFFERS', '3')))

    def _side_stream(self, n):
        """The stream of the filter gradients, or None for the one-stream walk.  Measured on MI355X (fp32, images/s, one stream ->
        two): 64 per GPU 7.95 k -> 8.30 k, 128: 9.08 k -> 9.44 k, 256: 9.79 k -> 9.96 k, 512: 10.44 k -> 10.30 k -- at 512 every
        kernel is many rounds of blocks, nothing is left to cover and two MFMA-bound kernels sharing the chip cost 1.3 %; in the
        bf16 mode (launches 3x shorter) 512 gains too: 16.63 -> 16.13 ms.  FTE_SIDE_STREAM=0 / 1 forces one / two streams."""
        if self.side is None or getattr(self, 'one_stream', False):      # one_stream: bench.py's launch-record steps (a launch's duration is its own)
            return None
        if os.environ.get('FTE_SIDE_STREAM') == '1':
            return self.side
        # ... and with the Winograd layers (round 6) at every size: the data gradient's tile transform (HBM-bound, 68 registers) shares the
        # CUs with the resident filter-gradient kernel of the other stream (MFMA-bound): 34.5 -> 33.4 ms per step at 512 images
        return self.side if (n <= 256 or _lib.get_mfma_dtype() == 'bf16' or _lib.query('fte_get_conv_algo') != 0) else None

    def backward_body(self):
        for _ in self._body_walk():
            pass

    def _body_walk(self):
        """Generator over the conv stack's backward pass; yields each time every filter gradient of one stage is final."""
        n = self._act_n
        st = _stream()
        call = _lib.call
        g = self.grads
        fcw = self.name + '/fully_connected/weights'
        L = self.convs
        last = len(L) - 1
        b4 = self.bwd[L[last].stage]
        d_out, dz_cur = b4['raw'][0], b4['dz'][0]
        b4['rawi'], b4['dzi'] = 0, 0
        call('fte_gemm_nt', self.demb, self.view(fcw), self.z[last], self.view(L[last].name + '/alpha'), L[last].cout,
             d_out, dz_cur, self.view(L[last].name + '/alpha', g), n, EMBED, self.fin, self.ws, self.ws_bytes, st)
        s16 = getattr(self, '_s16_live', False) and self._storage16()
        copies = s16 or (getattr(self, '_copies_live', False) and self._use_copies())
        dz16_cur = None
        d16_out = None                                  # bf16 storage: the skip-path gradient (what `d_out` is in fp32)
        if copies:
            dz16_cur = b4['dz16'][0]
            call('fte_to_bf16', dz_cur, dz16_cur, dz_cur.numel(), st)
        if s16:
            d16_out = b4['raw16'][0]
            call('fte_to_bf16', d_out, d16_out, d_out.numel(), st)
        trace = getattr(self, '_trace_dz', None)
        # Second stream: wgrad(l) and dgrad(l) both consume dz(l) and are independent of each other (and of every other layer's
        # wgrad), so the filter gradients are queued on `side` and share the chip with the dgrad chain -- at small per-GPU shards a
        # launch is one round of blocks whose prologue, epilogue and stragglers (~25 us of 130) are covered by the other stream's
        # MFMAs.  dz rotates through the stage's three buffers: before dgrad(l) overwrites one, the main stream waits for the
        # wgrad that read it (two layers earlier).
        # Same kernels, same operands, same order of every reduction: bit-identical to the one-stream walk.
        main, side = torch.cuda.current_stream(), self._side_stream(n)
        wst, wws = (side.cuda_stream, self.ws_side) if side is not None else (st, self.ws)
        readers = {}                                    # dz buffer -> event: the filter gradient that reads it has finished
        last_w = None
        for l in range(last, -1, -1):
            c = L[l]
            if trace is not None:
                trace[c.name] = (dz16_cur if s16 else dz_cur).clone()
            gw = self.view(c.name + '/weights', g)
            if l == 0:
                if last_w is not None:
                    main.wait_event(last_w)
                if s16:
                    call('fte_conv3x3_first_wgrad_s16', self._images, dz16_cur, gw, n, c.hin, c.win, c.cin, c.cout, c.stride,
                         self.ws, self.ws_bytes, st)
                else:
                    call('fte_conv3x3_first_wgrad', self._images, dz_cur, gw, n, c.hin, c.win, c.cin, c.cout, c.stride,
                         self.ws, self.ws_bytes, st)
                break
            if side is not None:
                side.wait_event(main.record_event())    # dz(l) is complete
            if copies:
                call('fte_conv2d_wgrad16', self.y16[l - 1], dz16_cur, gw, n, c.hin, c.win, c.cin, c.cout, 3, c.stride,
                     wws, self.ws_bytes, wst)
            elif getattr(self, '_vpack_live', False) and self.vpack[l] is not None:
                call('fte_conv3x3_wgrad_kept', self.y[l - 1], dz_cur, gw, n, c.hin, c.win, c.cin, c.cout, c.stride,
                     self.vpack[l], wws, self.ws_bytes, wst)
            else:
                call('fte_conv3x3_wgrad', self.y[l - 1], dz_cur, gw, n, c.hin, c.win, c.cin, c.cout, c.stride,
                     wws, self.ws_bytes, wst)
            if side is not None:
                last_w = readers[(dz16_cur if s16 else dz_cur).data_ptr()] = side.record_event()
            p = L[l - 1]
            bp = self.bwd[p.stage]
            if s16:
                # bf16 storage: dz and the skip-path gradient exist as bf16 only (same buffer rotation as below)
                if p.stage != c.stage:
                    bp['dzi'], bp['rawi'] = 0, 0
                    raw16_t = bp['raw16'][0]
                else:
                    bp['dzi'] = (bp['dzi'] + 1) % len(bp['dz16'])
                    raw16_t = bp['raw16'][bp['rawi'] ^ 1]
                dz16_prev = bp['dz16'][bp['dzi']]
                raw16 = raw16_t if p.second == 1 else None
                ev = readers.pop(dz16_prev.data_ptr(), None)
                if ev is not None:
                    main.wait_event(ev)
                call('fte_conv2d_dgrad_s16', dz16_cur, self.w16[c.name], d16_out if c.second == 0 else None, self.z16[l - 1],
                     self.view(p.name + '/alpha'), raw16, dz16_prev, self.view(p.name + '/alpha', g),
                     self.view(p.name + '/biases', g) if p.has_bias else None,
                     n, c.hin, c.win, c.cin, c.cout, 3, c.stride, self.ws, self.ws_bytes, st)
                if raw16 is not None:
                    d16_out = raw16
                    if p.stage == c.stage:
                        bp['rawi'] ^= 1
                dz16_cur = dz16_prev
                if p.stage != c.stage:
                    if last_w is not None:
                        main.wait_event(last_w)
                        last_w = None
                        readers.clear()
                    yield
                continue
            addin = d_out if c.second == 0 else None
            if p.stage != c.stage:
                bp['dzi'], bp['rawi'] = 0, 0
                dz_prev, raw_t = bp['dz'][0], bp['raw'][0]
            else:
                bp['dzi'] = (bp['dzi'] + 1) % len(bp['dz'])
                dz_prev = bp['dz'][bp['dzi']]
                raw_t = bp['raw'][bp['rawi'] ^ 1]
            raw = raw_t if p.second == 1 else None
            ev = readers.pop(dz_prev.data_ptr(), None)
            if ev is not None:
                main.wait_event(ev)                     # an earlier layer's wgrad still reads the buffer dgrad(l) is about to overwrite
            if copies:
                dz16_prev = bp['dz16'][bp['dzi']]
                call('fte_conv2d_dgrad16', dz16_cur, self.w16[c.name], addin, self.z[l - 1],
                     self.view(p.name + '/alpha'), raw, dz_prev, dz16_prev, self.view(p.name + '/alpha', g),
                     self.view(p.name + '/biases', g) if p.has_bias else None,
                     n, c.hin, c.win, c.cin, c.cout, 3, c.stride, self.ws, self.ws_bytes, st)
                dz16_cur = dz16_prev
            else:
                call('fte_conv3x3_dgrad', dz_cur, self.view(c.name + '/weights'), addin, self.z[l - 1],
                     self.view(p.name + '/alpha'), raw, dz_prev, self.view(p.name + '/alpha', g),
                     self.view(p.name + '/biases', g) if p.has_bias else None,
                     n, c.hin, c.win, c.cin, c.cout, c.stride, self.ws, self.ws_bytes, st)
            if raw is not None:
                d_out = raw
                if p.stage == c.stage:
                    bp['rawi'] ^= 1
            dz_cur = dz_prev
            if p.stage != c.stage:
                if last_w is not None:
                    main.wait_event(last_w)             # the stage's filter gradients are final for whoever follows on this stream
                    last_w = None
                    readers.clear()
                yield                                   # stage c.stage is done: its filter gradients are final
        yield

    # ------------------------------------------------------------------ bookkeeping the wrappers use
    def param_list(self, is_training, trainable, scope=None):
        """nets/sphere.py:120-126: [backbone vars, classifier vars] (lists of Variable)."""
        bb = [v for k, v in self.variables.items() if k.startswith(self.name + '/')]
        if is_training:
            return [bb, [self.variables['classifier/fc_classifier/weights']]]
        return [bb]

    def pretrained_param(self, scope=None):
        """nets/sphere.py:128-134: backbone variables only."""
        return [v for grp in self.param_list(is_training=False, trainable=False, scope=scope) for v in grp
                if self.name in v.name]

    def arena_groups(self):
        """(start, end, decayed, param_group_index) ranges of the flat arena, in arena order."""
        return [(0, self.small_end, False, 0), (self.small_end, self.cls_start, True, 0),
                (self.cls_start, self.arena_size, True, 1)]

    def grad_buckets(self):
        """All-reduce buckets in the order backward completes them: head (classifier + FC, produced first, with the 4 loss
        slots that follow the arena riding along), then the conv filters stage by stage from the last; the first stage's
        bucket also carries the biases and alphas at the front of the arena (final only after the last dgrad)."""
        firsts = self._stage_first                       # arena offset of each stage's first conv filter, ascending
        ends = firsts[1:] + [self.fc_start]
        buckets = [(self.fc_start, self.arena_size + 4)]
        for i in range(len(firsts) - 1, 0, -1):
            buckets.append((firsts[i], ends[i]))
        buckets.append((0, ends[0]))
        return buckets


class SphereNetMargin(SphereNet):
    """SphereNet-20 + A-softmax (SphereFace, m = 4): the margin net DataParallel_margin drives
    (data_parallel.py:220: forward(images, labels, num_classes=..., is_training=True)).  The head's
    code is absent from the reference snapshot; spec = SURVEY.md Appendix A.9."""
    head = 'asoftmax'
    needs_labels = True

    def __init__(self, weight_decay=0.0005, data_format='NCHW', name='SphereNet', seed=0,
                 lambda_base=1000.0, gamma=0.12, power=1.0, lambda_min=5.0):
        super(SphereNetMargin, self).__init__(weight_decay, data_format, name, seed)
        self.lambda_base, self.gamma, self.power, self.lambda_min = lambda_base, gamma, power, lambda_min

    def current_lambda(self):
        return max(self.lambda_min, self.lambda_base * (1.0 + self.gamma * self.global_step) ** (-self.power))

    def forward(self, images, labels=None, num_classes=None, is_training=True):
        if not is_training:
            return self._eval_features(images)
        assert num_classes is not None, 'num_classes must be given when is_training=True'
        assert labels is not None, 'margin nets take labels in forward (data_parallel.py:220)'
        self._ensure_built(images, num_classes)
        n = images.shape[0]
        labels = self._check_labels(labels)
        self._labels = labels
        st = _stream()
        self.backbone(images, is_training=True)
        self._classifier_raw(n)
        wc = self.view('classifier/fc_classifier/weights')
        _lib.call('fte_row_norms', self.emb, self.xn, n, EMBED, EMBED, st)
        _lib.call('fte_col_norms', wc, self.wn, EMBED, self.num_classes, self.cpad, st)
        self.lam = self.current_lambda()
        _lib.call('fte_asoftmax_fwd_bwd', self.s_raw, self.xn, self.wn, labels, self.lam, self.logits_buf, self.loss_rows,
                  self.G, self.rowcoef, n, self.num_classes, self.cpad, self._grad_scale(n), st)
        _lib.call('fte_asoftmax_colcoef', self.G, self.s_raw, self.wn, self.colcoef, n, self.num_classes, self.cpad, st)
        return {'logits': self.logits_buf[:, :self.num_classes]}

    def loss_function(self, scope, labels, **logits):
        n = labels.shape[0]
        self._finish_losses(n)
        others = OrderedDict()
        others['lambda'] = self.lam
        return [self.loss_slots[0], self.loss_slots[1]], ['cross_entropy', 'reg_loss'], others

    def _head_backward(self, n, st):
        super(SphereNetMargin, self)._head_backward(n, st)
        wc = self.view('classifier/fc_classifier/weights')
        gwc = self.view('classifier/fc_classifier/weights', self.grads)
        _lib.call('fte_add_scaled_rows_cols', gwc, wc, None, self.colcoef, EMBED, self.cpad, self.cpad, st)
        _lib.call('fte_add_scaled_rows_cols', self.demb, self.emb, self.rowcoef, None, n, EMBED, EMBED, st)
