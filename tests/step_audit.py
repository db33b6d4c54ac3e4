"""Audit of a REAL training step at the full per-GPU shard: every conv / grouped-conv / depthwise / batch-norm C-ABI call the step
makes is intercepted, and its outputs are compared with the float64 oracle evaluated on the call's OWN inputs (teacher forcing:
batch statistics make the layers of a BN net dependent on the whole shard, so the chain to the oracle goes call by call).  The
launch records name the kernel symbol every MFMA launch ran on, so the audit also yields {symbol -> checked against the oracle}
for exactly the dispatches of the profiled run (same shapes, same planner).

Used by tests/test_gpu_fullshard.py.  One check per distinct (entry point, shape) -- the nets repeat their blocks.
Reference semantics: nets/resnet.py:47-61,97-99 (conv / batch_norm eps 1e-3, biased batch variance), nets/resnext.py:41-51 (grouped
conv), nets/shufflenet_v2.py:87-115 (depthwise), TF-SAME padding."""
import numpy as np
import torch

from oracle import ops
from tf_face_toolbox_amd import _lib

TOL = 2e-5
IMG_BLOCK = 16


def _h(t, bf16=False):
    """device tensor -> float64 host array (int16 tensors hold bf16 bits)"""
    if t is None:
        return None
    if t.dtype == torch.int16:
        return t.view(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    return t.detach().float().cpu().numpy().astype(np.float64)


def _bf(a):
    return ops.bf16_round(np.asarray(a, np.float64))


class Audit(object):
    def __init__(self, net, bf16_operands):
        self.net = net
        self.bf = bf16_operands              # MFMA operands are rounded to bf16 (the 'bf16' / 'bf16s' modes)
        self.seen = set()
        self.checked = {}                    # entry point -> number of distinct shapes checked
        self.worst = {}                      # entry point -> worst error / limit
        self.symbols = {}                    # kernel symbol -> entry points whose checked calls ran on it
        self.real_call = _lib.call

    # ---- plumbing ------------------------------------------------------------------------------------------------------------
    def __enter__(self):
        _lib.call = self._call
        return self

    def __exit__(self, *exc):
        _lib.call = self.real_call

    def _call(self, fn, *args):
        chk = getattr(self, '_chk_' + fn, None)
        key = (fn,) + tuple(a for a in args if isinstance(a, (int, float)) and not isinstance(a, bool)) + \
            tuple(i for i, a in enumerate(args) if a is None)
        if chk is None or key in self.seen:
            return self.real_call(fn, *args)
        self.seen.add(key)
        torch.cuda.synchronize()
        _lib.query('fte_prof_enable', 1)
        r = self.real_call(fn, *args)
        torch.cuda.synchronize()
        _lib.query('fte_prof_enable', 0)
        syms = sorted({rec[5] for rec in _lib.prof_records(shapes=True) if rec[5]})
        chk(*args)
        self.checked[fn] = self.checked.get(fn, 0) + 1
        for s in syms:
            self.symbols.setdefault(s, set()).add(fn)
        return r

    def _ok(self, fn, what, got, ref, stored16=False, tol=TOL, rell2=False):
        got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        assert got.shape == ref.shape, (fn, what, got.shape, ref.shape)
        assert np.isfinite(got).all(), (fn, what)
        scale = max(float(np.abs(ref).max()), 1e-30)
        if rell2:
            ratio = float(np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref * ref).sum()), 1e-30)) / tol
        elif stored16:           # a bf16-stored value: within half a bf16 step of the float64 value (+ fp32 noise)
            ratio = float((np.abs(got - ref) / (np.abs(ref) * 2.0 ** -8 + tol * scale)).max())
        else:
            ratio = float(np.abs(got - ref).max() / (tol * scale))
        self.worst[fn] = max(self.worst.get(fn, 0.0), ratio)
        assert ratio <= 1.0, '%s: %s off by %.2f x its limit' % (fn, what, ratio)

    def _w(self, a):
        return _bf(a) if self.bf else a

    # ---- oracle pieces ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv_blocks(x, w, stride):
        return np.concatenate([ops.conv2d_fwd(x[i:i + IMG_BLOCK], w, stride) for i in range(0, x.shape[0], IMG_BLOCK)], axis=0)

    def _stats(self, fn, z, gamma, beta, mean, rstd, scale, shift):
        c = z.shape[-1]
        zz = z.reshape(-1, c)
        m = zz.mean(0)
        v = zz.var(0)
        r = 1.0 / np.sqrt(v + 1e-3)
        lim = 2e-6 * (float(np.abs(m).max()) + float(np.sqrt(v).max()))      # a mean is as good as the spread of what it averages
        assert float(np.abs(_h(mean) - m).max()) <= lim, '%s: batch mean off by %.2e (limit %.2e)' % (fn, float(np.abs(_h(mean) - m).max()), lim)
        self._ok(fn, 'rstd', _h(rstd), r, tol=1e-5)
        g, b = _h(gamma), _h(beta)
        self._ok(fn, 'scale', _h(scale), g * r, tol=1e-5)
        sref = b - m * g * r                 # shift = beta - mean * scale: as good as the mean it is made of
        slim = 2e-5 * float(np.abs(sref).max()) + 2 * lim * float(np.abs(g * r).max())
        assert float(np.abs(_h(shift) - sref).max()) <= slim, '%s: shift off by %.2e (limit %.2e)' % (fn, float(np.abs(_h(shift) - sref).max()), slim)

    @staticmethod
    def _unpack_w16t(w16t, k, cin, cout):
        """[tap][cout][cin] bf16 pack -> HWIO float64"""
        return _h(w16t[:k * k * cout * cin]).reshape(k, k, cout, cin).transpose(0, 1, 3, 2)

    @staticmethod
    def _unpack_w16(w16, k, cin, cout):
        return _h(w16[:k * k * cin * cout]).reshape(k, k, cin, cout)

    @staticmethod
    def _gpack_dense(wpk, c):
        """grouped-conv pack [slice][tap][col 32][k 32] -> per slice HWIO [3][3][k][col] float64 (block-diagonal)"""
        return _h(wpk).reshape(c // 32, 3, 3, 32, 32).transpose(0, 1, 2, 4, 3)

    def _gconv(self, x, wd, stride):
        c = x.shape[-1]
        return np.concatenate([self._conv_blocks(x[..., s * 32:(s + 1) * 32], wd[s], stride) for s in range(c // 32)], axis=-1)

    # ---- forward -----------------------------------------------------------------------------------------------------------------
    def _chk_fte_conv2d_bn_fwd(self, x, w, z, gamma, beta, mean, rstd, scale, shift, mm, mv, eps, decay, isc, ish, yside,
                               n, h, wd, cin, cout, k, stride, s16, ws, wsb, st):
        fn = 'fte_conv2d_bn_fwd'
        xin = _h(x).reshape(n, h, wd, cin)
        if isc is not None:                 # the loader's normalise pass: y = relu(isc * x + ish), rounded where it is written back
            y = np.maximum(xin * _h(isc) + _h(ish), 0)
            self._ok(fn, 'side-stored y', _h(yside).reshape(xin.shape), y, stored16=True)
            xin = _h(yside).reshape(xin.shape)
        wt = self._unpack_w16t(w, k, cin, cout) if s16 else self._w(_h(w).reshape(k, k, cin, cout))
        ref = self._conv_blocks(self._w(xin), wt, stride)
        zz = _h(z).reshape(ref.shape)
        self._ok(fn, 'z', zz, ref, stored16=bool(s16))
        self._stats(fn, zz, gamma, beta, mean, rstd, scale, shift)

    def _chk_fte_conv2d_fwd_s16(self, x, w16t, bias, alpha, res, z16, y16, z32, y32, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        assert bias is None and alpha is None and res is None
        ref = self._conv_blocks(_h(x).reshape(n, h, wd, cin), self._unpack_w16t(w16t, k, cin, cout), stride)
        self._ok('fte_conv2d_fwd_s16', 'y', _h(y16).reshape(ref.shape), ref, stored16=True)

    def _chk_fte_conv2d_fwd(self, x, w, bias, alpha, res, z, y, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        assert bias is None and alpha is None and res is None
        ref = self._conv_blocks(self._w(_h(x).reshape(n, h, wd, cin)), self._w(_h(w).reshape(k, k, cin, cout)), stride)
        self._ok('fte_conv2d_fwd', 'y', _h(y).reshape(ref.shape), ref)

    def _chk_fte_gconv3x3_bn_fwd_bf16_s16(self, x, wpk, z, gamma, beta, mean, rstd, scale, shift, mm, mv, eps, decay, isc, ish, yside,
                                          n, h, wd, c, stride, ws, wsb, st):
        fn = 'fte_gconv3x3_bn_fwd_bf16_s16'
        xin = _h(x).reshape(n, h, wd, c)
        if isc is not None:
            y = np.maximum(xin * _h(isc) + _h(ish), 0)
            self._ok(fn, 'side-stored y', _h(yside).reshape(xin.shape), y, stored16=True)
            xin = _h(yside).reshape(xin.shape)
        ref = self._gconv(xin, self._gpack_dense(wpk, c), stride)
        zz = _h(z).reshape(ref.shape)
        self._ok(fn, 'z', zz, ref, stored16=True)
        self._stats(fn, zz, gamma, beta, mean, rstd, scale, shift)

    def _chk_fte_gconv3x3_bf16_s16(self, x, wpk, y, n, h, wd, c, stride, dgrad, st):
        fn = 'fte_gconv3x3_bf16_s16'
        wdn = self._gpack_dense(wpk, c)
        if not dgrad:
            ref = self._gconv(_h(x).reshape(n, h, wd, c), wdn, stride)
            self._ok(fn, 'y', _h(y).reshape(ref.shape), ref, stored16=True)
            return
        # data gradient: the pack holds the mirrored, transposed filter -- dx = correlation of dz with it; as a gradient of the forward
        # conv with filter wf[r][q][ic][oc] = pack[2 - r][2 - q][oc][ic]
        ho, wo = ops.same_pads(h, 3, stride)[0], ops.same_pads(wd, 3, stride)[0]
        dz = _h(x).reshape(n, ho, wo, c)
        wf = wdn[:, ::-1, ::-1].transpose(0, 1, 2, 4, 3)
        ref = np.concatenate([np.concatenate([ops.conv2d_bwd(np.zeros((min(IMG_BLOCK, n - i), h, wd, 32)), wf[s], dz[i:i + IMG_BLOCK, ..., s * 32:(s + 1) * 32],
                                                              stride, need_dw=False)[0] for i in range(0, n, IMG_BLOCK)], axis=0)
                              for s in range(c // 32)], axis=-1)
        self._ok(fn, 'dx', _h(y).reshape(ref.shape), ref, stored16=True)

    def _chk_fte_bn_apply(self, z, scale, shift, res, y, rows, c, relu, flags, st):
        v = _h(z).reshape(rows, c) * _h(scale) + _h(shift)
        if res is not None:
            v = v + _h(res).reshape(rows, c)
        if relu:
            v = np.maximum(v, 0)
        self._ok('fte_bn_apply', 'y', _h(y).reshape(rows, c), v, stored16=bool(flags & 2))

    def _bn_fwd(self, fn, z, gamma, beta, res, y, mean, rstd, scale, shift, rows, c, relu, out16):
        zz = _h(z).reshape(rows, c)
        self._stats(fn, zz, gamma, beta, mean, rstd, scale, shift)
        v = zz * _h(scale) + _h(shift)
        if res is not None:
            v = v + _h(res).reshape(rows, c)
        if relu:
            v = np.maximum(v, 0)
        self._ok(fn, 'y', _h(y).reshape(rows, c), v, stored16=out16)

    def _chk_fte_bn_train_fwd_s16(self, z, gamma, beta, res, y, mean, rstd, scale, shift, mm, mv, rows, c, eps, decay, relu, flags, ws, wsb, st):
        self._bn_fwd('fte_bn_train_fwd_s16', z, gamma, beta, res, y, mean, rstd, scale, shift, rows, c, relu, bool(flags & 2))

    def _chk_fte_bn_train_fwd(self, z, gamma, beta, res, y, mean, rstd, scale, shift, mm, mv, rows, c, eps, decay, relu, ws, wsb, st):
        self._bn_fwd('fte_bn_train_fwd', z, gamma, beta, res, y, mean, rstd, scale, shift, rows, c, relu, False)

    def _chk_fte_bn_train_stats(self, z, gamma, beta, mean, rstd, scale, shift, mm, mv, rows, c, eps, decay, ws, wsb, st):
        self._stats('fte_bn_train_stats', _h(z).reshape(rows, c), gamma, beta, mean, rstd, scale, shift)

    def _chk_fte_dwconv3x3_fwd(self, x, w, y, n, h, wd, c, stride, st):
        ref = np.concatenate([ops.dwconv3x3_fwd(_h(x).reshape(n, h, wd, c)[i:i + IMG_BLOCK], _h(w).reshape(3, 3, c, 1), stride) for i in range(0, n, IMG_BLOCK)], axis=0)
        self._ok('fte_dwconv3x3_fwd', 'y', _h(y).reshape(ref.shape), ref)

    # ---- backward ----------------------------------------------------------------------------------------------------------------
    def _dgrad(self, fn, dz, wt, addin, dx, n, h, wd, cin, cout, k, stride, out16):
        ho, wo = ops.same_pads(h, k, stride)[0], ops.same_pads(wd, k, stride)[0]
        d = _h(dz).reshape(n, ho, wo, cout)
        ref = np.concatenate([ops.conv2d_bwd(np.zeros((min(IMG_BLOCK, n - i), h, wd, cin)), wt, self._w(d[i:i + IMG_BLOCK]), stride, need_dw=False)[0]
                              for i in range(0, n, IMG_BLOCK)], axis=0)
        if addin is not None:
            ref = ref + _h(addin).reshape(ref.shape)
        self._ok(fn, 'dx', _h(dx).reshape(ref.shape), ref, stored16=out16)

    def _chk_fte_conv2d_dgrad_s16(self, dz, w16, addin, zprev, alpha, raw, dzprev, dalpha, dbias, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        assert zprev is None and raw is None
        self._dgrad('fte_conv2d_dgrad_s16', dz, self._unpack_w16(w16, k, cin, cout), addin, dzprev, n, h, wd, cin, cout, k, stride, True)

    def _chk_fte_conv2d_dgrad(self, dz, w, addin, zprev, alpha, raw, dzprev, dalpha, dbias, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        assert zprev is None and raw is None
        self._dgrad('fte_conv2d_dgrad', dz, self._w(_h(w).reshape(k, k, cin, cout)), addin, dzprev, n, h, wd, cin, cout, k, stride, False)

    def _wgrad(self, fn, x, dz, dw, n, h, wd, cin, cout, k, stride):
        ho, wo = ops.same_pads(h, k, stride)[0], ops.same_pads(wd, k, stride)[0]
        xx, d = self._w(_h(x).reshape(n, h, wd, cin)), self._w(_h(dz).reshape(n, ho, wo, cout))
        ref = np.zeros((k, k, cin, cout))
        wz = np.zeros((k, k, cin, cout))
        for i in range(0, n, IMG_BLOCK):
            ref += ops.conv2d_bwd(xx[i:i + IMG_BLOCK], wz, d[i:i + IMG_BLOCK], stride, need_dx=False)[1]
        self._ok(fn, 'dw', _h(dw).reshape(ref.shape), ref)

    def _chk_fte_conv2d_wgrad16(self, x16, dz16, dw, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        self._wgrad('fte_conv2d_wgrad16', x16, dz16, dw, n, h, wd, cin, cout, k, stride)

    def _chk_fte_conv2d_wgrad(self, x, dz, dw, n, h, wd, cin, cout, k, stride, ws, wsb, st):
        self._wgrad('fte_conv2d_wgrad', x, dz, dw, n, h, wd, cin, cout, k, stride)

    def _chk_fte_gconv3x3_wgrad_bf16_s16(self, x16, dz16, dw, n, h, wd, c, groups, stride, ws, wsb, st):
        gw = c // groups
        ho, wo = ops.same_pads(h, 3, stride)[0], ops.same_pads(wd, 3, stride)[0]
        xx, d = _h(x16).reshape(n, h, wd, c), _h(dz16).reshape(n, ho, wo, c)
        ref = np.zeros((groups, 3, 3, gw, gw))
        wz = np.zeros((3, 3, gw, gw))
        for g in range(groups):
            for i in range(0, n, 4 * IMG_BLOCK):
                ref[g] += ops.conv2d_bwd(xx[i:i + 4 * IMG_BLOCK, ..., g * gw:(g + 1) * gw], wz, d[i:i + 4 * IMG_BLOCK, ..., g * gw:(g + 1) * gw], stride, need_dx=False)[1]
        self._ok('fte_gconv3x3_wgrad_bf16_s16', 'dw', _h(dw).reshape(ref.shape), ref)

    def _bn_bwd(self, fn, dy, ymask, z, gamma, mean, rstd, scale, shift, gout, dz, dgamma, dbeta, rows, c, g16, z16):
        g = _h(dy).reshape(rows, c)
        zz = _h(z).reshape(rows, c)
        if scale is not None:            # the mask recomputed from z with the forward pass's own expression (fp32 fma: sign of the exact value)
            g = g * ((zz * _h(scale) + _h(shift)) > 0)
        elif ymask is not None:
            g = g * (_h(ymask).reshape(rows, c) > 0)
        if gout is not None:
            self._ok(fn, 'masked gradient', _h(gout).reshape(rows, c), g, stored16=g16, tol=1e-7 if not g16 else TOL)
        xhat = (zz - _h(mean)) * _h(rstd)
        db, dg = g.sum(0), (g * xhat).sum(0)
        # a sum is as good as the magnitude of its terms: the gradient reaching a layer that feeds another batch norm sums to ~0 per
        # channel (1e-11 here), which no relative figure of the SUM can be held to
        for what, got, ref, terms in (('dbeta', _h(dbeta), db, np.abs(g).sum(0)), ('dgamma', _h(dgamma), dg, np.abs(g * xhat).sum(0))):
            lim = 2e-6 * float(terms.max()) + 1e-30
            err = float(np.abs(got - ref).max())
            self.worst[fn] = max(self.worst.get(fn, 0.0), err / lim)
            assert err <= lim, '%s: %s off by %.2e (limit %.2e = 2e-6 x the largest sum of |terms|)' % (fn, what, err, lim)
        ref = _h(gamma) * _h(rstd) * (g - db / rows - xhat * (dg / rows))
        self._ok(fn, 'dz', _h(dz).reshape(rows, c), ref, stored16=z16)

    def _chk_fte_bn_train_bwd_s16(self, dy, y, z, gamma, mean, rstd, scale, shift, gout, dz, dgamma, dbeta, rows, c, flags, ws, wsb, st):
        self._bn_bwd('fte_bn_train_bwd_s16', dy, y, z, gamma, mean, rstd, scale, shift, gout, dz, dgamma, dbeta, rows, c, bool(flags & 2), bool(flags & 1))

    def _chk_fte_bn_train_bwd(self, dy, ymask, z, gamma, mean, rstd, dz, dgamma, dbeta, rows, c, ws, wsb, st):
        self._bn_bwd('fte_bn_train_bwd', dy, ymask, z, gamma, mean, rstd, None, None, None, dz, dgamma, dbeta, rows, c, False, False)

    def _chk_fte_bn_train_bwd_zmask(self, dy, z, gamma, mean, rstd, scale, shift, dz, dgamma, dbeta, rows, c, ws, wsb, st):
        self._bn_bwd('fte_bn_train_bwd_zmask', dy, None, z, gamma, mean, rstd, scale, shift, None, dz, dgamma, dbeta, rows, c, False, False)

    def _chk_fte_bn_train_bwd_res(self, dy, y, z, gamma, mean, rstd, gout, dz, dgamma, dbeta, rows, c, ws, wsb, st):
        self._bn_bwd('fte_bn_train_bwd_res', dy, y, z, gamma, mean, rstd, None, None, gout, dz, dgamma, dbeta, rows, c, False, False)

    def _chk_fte_dwconv3x3_dgrad(self, dy, w, dx, n, h, wd, c, stride, st):
        ho, wo = ops.same_pads(h, 3, stride)[0], ops.same_pads(wd, 3, stride)[0]
        d = _h(dy).reshape(n, ho, wo, c)
        ww = _h(w).reshape(3, 3, c, 1)
        ref = np.concatenate([ops.dwconv3x3_bwd(np.zeros((min(IMG_BLOCK, n - i), h, wd, c)), ww, d[i:i + IMG_BLOCK], stride)[0] for i in range(0, n, IMG_BLOCK)], axis=0)
        self._ok('fte_dwconv3x3_dgrad', 'dx', _h(dx).reshape(ref.shape), ref)

    def _chk_fte_dwconv3x3_wgrad(self, x, dy, dw, n, h, wd, c, stride, ws, wsb, st):
        ho, wo = ops.same_pads(h, 3, stride)[0], ops.same_pads(wd, 3, stride)[0]
        xx, d = _h(x).reshape(n, h, wd, c), _h(dy).reshape(n, ho, wo, c)
        ref = np.zeros((3, 3, c, 1))
        for i in range(0, n, IMG_BLOCK):
            ref += ops.dwconv3x3_bwd(xx[i:i + IMG_BLOCK], np.zeros((3, 3, c, 1)), d[i:i + IMG_BLOCK], stride)[1]
        self._ok('fte_dwconv3x3_wgrad', 'dw', _h(dw).reshape(ref.shape), ref)

    # ---- dense products (7x7 stem through im2col, classifier, SE gate) ------------------------------------------------------------------
    def _chk_fte_gemm_nn(self, x, w, bias, y, m, n, k, ws, wsb, st):
        ref = self._w(_h(x).reshape(m, k)) @ self._w(_h(w).reshape(k, n))
        if bias is not None:
            ref = ref + _h(bias)
        self._ok('fte_gemm_nn', 'y', _h(y).reshape(m, n), ref)

    def _chk_fte_gemm_nn_act(self, x, w, bias, y, m, n, k, act, ws, wsb, st):
        ref = self._w(_h(x).reshape(m, k)) @ self._w(_h(w).reshape(k, n))
        if bias is not None:
            ref = ref + _h(bias)
        if act == 1:
            ref = np.maximum(ref, 0.0)
        elif act == 2:
            ref = 1.0 / (1.0 + np.exp(-ref))
        self._ok('fte_gemm_nn_act', 'y', _h(y).reshape(m, n), ref)

    def _chk_fte_gemm_nt(self, dy, w, zprev, alpha, amod, raw, dx, dalpha, m, n, k, ws, wsb, st):
        assert zprev is None and raw is None
        ref = self._w(_h(dy).reshape(m, n)) @ self._w(_h(w).reshape(k, n)).T
        self._ok('fte_gemm_nt', 'dx', _h(dx).reshape(m, k), ref)

    def _chk_fte_gemm_tn(self, x, dy, dw, m, n, k, ws, wsb, st):
        ref = self._w(_h(x).reshape(m, k)).T @ self._w(_h(dy).reshape(m, n))
        self._ok('fte_gemm_tn', 'dw', _h(dw).reshape(k, n), ref)
