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

    def _chk_fte_dense_small(self, a, w, bias, mask, out, m, n, k, trans_w, act, st):
        ww = self._w(_h(w).reshape(n, k)).T if trans_w else self._w(_h(w).reshape(k, n))
        ref = self._w(_h(a).reshape(m, k)) @ ww
        if bias is not None:
            ref = ref + _h(bias)
        if act == 1:
            ref = np.maximum(ref, 0.0)
        elif act == 2:
            ref = 1.0 / (1.0 + np.exp(-ref))
        if mask is not None:
            ref = ref * (_h(mask).reshape(m, n) > 0)
        self._ok('fte_dense_small', 'out', _h(out).reshape(m, n), ref)

    def _chk_fte_gemm_nt(self, dy, w, zprev, alpha, amod, raw, dx, dalpha, m, n, k, ws, wsb, st):
        assert zprev is None and raw is None
        ref = self._w(_h(dy).reshape(m, n)) @ self._w(_h(w).reshape(k, n)).T
        self._ok('fte_gemm_nt', 'dx', _h(dx).reshape(m, k), ref)

    def _chk_fte_gemm_tn(self, x, dy, dw, m, n, k, ws, wsb, st):
        ref = self._w(_h(x).reshape(m, k)).T @ self._w(_h(dy).reshape(m, n))
        self._ok('fte_gemm_tn', 'dw', _h(dw).reshape(k, n), ref)

    # ---- streaming kernels (round 5: the audit covers every entry point of the step, not only the conv / BN / dense families) ------------
    @staticmethod
    def _nhwc(t, n, hw, c):
        return _h(t).reshape(n, hw, c)

    def _chk_fte_se_squeeze(self, z, scale, shift, mean, rstd, sq, xm, n, hw, c, flags, st):
        zm = self._nhwc(z, n, hw, c).mean(1)
        self._ok('fte_se_squeeze', 'sq', _h(sq).reshape(n, c), zm * _h(scale) + _h(shift))
        if xm is not None:      # (mean_hw z - mean) * rstd: as good as the mean it subtracts (error ~ eps * |mean| * rstd)
            ref = (zm - _h(mean)) * _h(rstd)
            lim = 2e-6 * (1.0 + float((np.abs(_h(mean)) * _h(rstd)).max()))
            err = float(np.abs(_h(xm).reshape(n, c) - ref).max())
            self.worst['fte_se_squeeze'] = max(self.worst.get('fte_se_squeeze', 0.0), err / lim)
            assert err <= lim, 'fte_se_squeeze: xm off by %.2e (limit %.2e)' % (err, lim)

    def _chk_fte_se_apply_fwd(self, z, scale, shift, gate, sc, out, n, hw, c, flags, st):
        y = self._nhwc(z, n, hw, c) * _h(scale) + _h(shift)
        ref = np.maximum(y * _h(gate).reshape(n, 1, c) + self._nhwc(sc, n, hw, c), 0)
        self._ok('fte_se_apply_fwd', 'out', self._nhwc(out, n, hw, c), ref, stored16=bool(flags & 2))

    def _chk_fte_se_bwd_gate(self, dy, out, z, gamma, beta, mean, rstd, gate, g, s1, s2, dgate, n, hw, c, flags, st):
        fn = 'fte_se_bwd_gate'
        gr = self._nhwc(dy, n, hw, c) * (self._nhwc(out, n, hw, c) > 0)
        gg = self._nhwc(g, n, hw, c)
        self._ok(fn, 'g', gg, gr, stored16=bool(flags & 2), tol=1e-7 if not flags & 2 else TOL)
        xhat = (self._nhwc(z, n, hw, c) - _h(mean)) * _h(rstd)
        r1, r2 = gg.sum(1), (gg * xhat).sum(1)              # the STORED g is what the kernel sums (fte.h)
        for what, got, ref, terms in (('s1', _h(s1), r1, np.abs(gg).sum(1)), ('s2', _h(s2), r2, np.abs(gg * xhat).sum(1))):
            lim = 2e-6 * float(terms.max()) + 1e-30
            err = float(np.abs(got.reshape(n, c) - ref).max())
            self.worst[fn] = max(self.worst.get(fn, 0.0), err / lim)
            assert err <= lim, '%s: %s off by %.2e (limit %.2e = 2e-6 x the largest sum of |terms|)' % (fn, what, err, lim)
        gt = _h(gate).reshape(n, c)
        ref = (_h(gamma) * r2 + _h(beta) * r1) * gt * (1 - gt)
        lim = 2e-6 * float(((np.abs(_h(gamma)) * np.abs(gg * xhat).sum(1) + np.abs(_h(beta)) * np.abs(gg).sum(1)) * gt * (1 - gt)).max()) + 1e-30
        err = float(np.abs(_h(dgate).reshape(n, c) - ref).max())
        self.worst[fn] = max(self.worst.get(fn, 0.0), err / lim)
        assert err <= lim, '%s: dgate off by %.2e (limit %.2e)' % (fn, err, lim)

    def _chk_fte_se_bn_bwd_coef(self, s1, s2, gate, dsq, xm, gamma, mean, rstd, dgamma, dbeta, coef, n, hw, c, st):
        fn = 'fte_se_bn_bwd_coef'
        a1, a2, gt, dq, x = (_h(t).reshape(n, c) for t in (s1, s2, gate, dsq, xm))
        db, dg = (gt * a1 + dq).sum(0), (gt * a2 + dq * x).sum(0)
        for what, got, ref, terms in (('dbeta', _h(dbeta), db, (np.abs(gt * a1) + np.abs(dq)).sum(0)), ('dgamma', _h(dgamma), dg, (np.abs(gt * a2) + np.abs(dq * x)).sum(0))):
            lim = 2e-6 * float(terms.max()) + 1e-30
            err = float(np.abs(got - ref).max())
            self.worst[fn] = max(self.worst.get(fn, 0.0), err / lim)
            assert err <= lim, '%s: %s off by %.2e (limit %.2e)' % (fn, what, err, lim)
        cnt = float(n * hw)
        gr = _h(gamma) * _h(rstd)
        bb = -gr * _h(rstd) * dg / cnt
        cf = _h(coef).reshape(3, c)
        self._ok(fn, 'A', cf[0], gr, tol=1e-6)
        self._ok(fn, 'B', cf[1], bb, tol=1e-5 * max(1.0, float(np.abs(dg).max()) and 1.0))
        c0 = -gr * db / cnt - bb * _h(mean)
        lim = 1e-5 * float((np.abs(gr * db / cnt) + np.abs(bb * _h(mean))).max()) + 1e-30
        assert float(np.abs(cf[2] - c0).max()) <= lim, '%s: C0 off by %.2e (limit %.2e)' % (fn, float(np.abs(cf[2] - c0).max()), lim)

    def _chk_fte_se_bn_bwd_apply(self, g, z, coef, gate, dsq, dz, n, hw, c, flags, st):
        cf = _h(coef).reshape(3, c)
        dyb = self._nhwc(g, n, hw, c) * _h(gate).reshape(n, 1, c) + _h(dsq).reshape(n, 1, c) / hw
        ref = cf[0] * dyb + cf[1] * self._nhwc(z, n, hw, c) + cf[2]
        self._ok('fte_se_bn_bwd_apply', 'dz', self._nhwc(dz, n, hw, c), ref, stored16=bool(flags & 1))

    def _chscale_fwd(self, fn, x, gate, y, n, hw, c, s16):
        self._ok(fn, 'y', self._nhwc(y, n, hw, c), self._nhwc(x, n, hw, c) * _h(gate).reshape(n, 1, c), stored16=s16)

    def _chk_fte_channel_scale_fwd(self, x, gate, y, n, hw, c, st):
        self._chscale_fwd('fte_channel_scale_fwd', x, gate, y, n, hw, c, False)

    def _chk_fte_channel_scale_fwd_s16(self, x, gate, y, n, hw, c, st):
        self._chscale_fwd('fte_channel_scale_fwd_s16', x, gate, y, n, hw, c, True)

    def _dgate(self, fn, dy, x, gate, dgate, n, hw, c, pre_sigmoid):
        d, xx, gt = self._nhwc(dy, n, hw, c), self._nhwc(x, n, hw, c), _h(gate).reshape(n, c)
        ref = (d * xx).sum(1)
        terms = np.abs(d * xx).sum(1)
        if pre_sigmoid:
            ref, terms = ref * gt * (1 - gt), terms * gt * (1 - gt)
        lim = 2e-6 * float(terms.max()) + 1e-30
        err = float(np.abs(_h(dgate).reshape(n, c) - ref).max())
        self.worst[fn] = max(self.worst.get(fn, 0.0), err / lim)
        assert err <= lim, '%s: dgate off by %.2e (limit %.2e)' % (fn, err, lim)

    def _chk_fte_channel_scale_bwd(self, dy, x, gate, dx, dgate, n, hw, c, pre_sigmoid, st):
        self._dgate('fte_channel_scale_bwd', dy, x, gate, dgate, n, hw, c, pre_sigmoid)
        if dx is not None:
            self._ok('fte_channel_scale_bwd', 'dx', self._nhwc(dx, n, hw, c), self._nhwc(dy, n, hw, c) * _h(gate).reshape(n, 1, c))

    def _chk_fte_channel_scale_bwd_s16(self, dy, x, gate, dgate, n, hw, c, pre_sigmoid, st):
        self._dgate('fte_channel_scale_bwd_s16', dy, x, gate, dgate, n, hw, c, pre_sigmoid)

    def _chk_fte_channel_scale_bwd_apply_s16(self, dy, gate, dsq, dx, n, hw, c, scale, st):
        ref = self._nhwc(dy, n, hw, c) * _h(gate).reshape(n, 1, c) + _h(dsq).reshape(n, 1, c) * scale
        self._ok('fte_channel_scale_bwd_apply_s16', 'dx', self._nhwc(dx, n, hw, c), ref, stored16=True)

    def _chk_fte_bcast_add(self, dx, v, n, hw, c, scale, st):
        pass        # in place: checked through fte_channel_scale_bwd's dx and the block's BN backward (the input is gone)

    def _relu_bwd(self, fn, dy, y, g):
        ref = _h(dy) * (_h(y) > 0)
        assert np.array_equal(_h(g).reshape(ref.shape), ref), '%s: g != dy * (y > 0) exactly' % fn      # a product with 0 / 1: no rounding
        self.worst[fn] = max(self.worst.get(fn, 0.0), 0.0)

    def _chk_fte_relu_bwd(self, dy, y, g, n, st):
        self._relu_bwd('fte_relu_bwd', dy, y, g)

    def _chk_fte_relu_bwd_s16(self, dy, y, g, n, st):
        self._relu_bwd('fte_relu_bwd_s16', dy, y, g)

    def _bn_infer(self, fn, z, gamma, beta, mm, mv, res, y, scale, shift, rows, c, eps, relu, out16):
        sc = _h(gamma) / np.sqrt(_h(mv) + eps)
        sh = _h(beta) - _h(mm) * sc
        self._ok(fn, 'scale', _h(scale), sc, tol=1e-6)
        self._ok(fn, 'shift', _h(shift), sh, tol=1e-6) if float(np.abs(sh).max()) > 0 else None
        v = _h(z).reshape(rows, c) * sc + sh
        if res is not None:
            v = v + _h(res).reshape(rows, c)
        if relu:
            v = np.maximum(v, 0)
        self._ok(fn, 'y', _h(y).reshape(rows, c), v, stored16=out16)

    def _chk_fte_bn_infer_fwd(self, z, gamma, beta, mm, mv, res, y, scale, shift, rows, c, eps, relu, st):
        self._bn_infer('fte_bn_infer_fwd', z, gamma, beta, mm, mv, res, y, scale, shift, rows, c, eps, relu, False)

    def _chk_fte_bn_infer_fwd_s16(self, z, gamma, beta, mm, mv, res, y, scale, shift, rows, c, eps, relu, flags, st):
        self._bn_infer('fte_bn_infer_fwd_s16', z, gamma, beta, mm, mv, res, y, scale, shift, rows, c, eps, relu, bool(flags & 2))

    @staticmethod
    def _gather(a, b, table, rows, ca, cb, sa=None, sb=None):
        """out[row, k] = table[k] < 0 ? 0 : (table[k] >> 16 ? b : a)[row, table[k] & 0xffff], a source with (scale, shift, relu) normalised on the way"""
        t = table.cpu().numpy().astype(np.int64)
        srcs = []
        for src, cs, aff in ((a, ca, sa), (b, cb, sb)):
            if src is None:
                srcs.append(None)
                continue
            v = _h(src).reshape(rows, cs)
            if aff is not None and aff[0] is not None:
                v = v * _h(aff[0]) + _h(aff[1])
                if aff[2]:
                    v = np.maximum(v, 0)
            srcs.append(v)
        out = np.zeros((rows, len(t)))
        for k, e in enumerate(t):
            if e >= 0:
                out[:, k] = srcs[e >> 16][:, e & 0xffff]
        return out

    def _chk_fte_channel_gather(self, a, b, out, table, rows, ca, cb, co, st):
        ref = self._gather(a, b, table, rows, ca, cb)
        assert np.array_equal(_h(out).reshape(rows, co), ref), 'fte_channel_gather: not the table\'s permutation'
        self.worst['fte_channel_gather'] = 0.0

    def _chk_fte_channel_gather_s16(self, a, b, out, table, rows, ca, cb, co, st):
        ref = self._gather(a, b, table, rows, ca, cb)
        assert np.array_equal(_h(out).reshape(rows, co), ref), 'fte_channel_gather_s16: not the table\'s permutation'
        self.worst['fte_channel_gather_s16'] = 0.0

    def _gather_affine(self, fn, a, b, out, table, co, out1, table1, co1, rows, ca, cb, sca, sha, ra, scb, shb, rb, s16):
        for o, t, cw in ((out, table, co), (out1, table1, co1)):
            if o is None:
                continue
            ref = self._gather(a, b, t, rows, ca, cb, (sca, sha, ra), (scb, shb, rb))
            self._ok(fn, 'out', _h(o).reshape(rows, cw), ref, stored16=s16)

    def _chk_fte_channel_gather_affine(self, a, b, out, table, co, out1, table1, co1, rows, ca, cb, sca, sha, ra, scb, shb, rb, st):
        self._gather_affine('fte_channel_gather_affine', a, b, out, table, co, out1, table1, co1, rows, ca, cb, sca, sha, ra, scb, shb, rb, False)

    def _chk_fte_channel_gather_affine_s16(self, a, b, out, table, co, out1, table1, co1, rows, ca, cb, sca, sha, ra, scb, shb, rb, st):
        self._gather_affine('fte_channel_gather_affine_s16', a, b, out, table, co, out1, table1, co1, rows, ca, cb, sca, sha, ra, scb, shb, rb, True)

    def _maxpool_fwd(self, fn, x, y, idx, n, h, wd, c):
        xx = _h(x).reshape(n, h, wd, c)
        for i in range(0, n, IMG_BLOCK):
            ref, cache = ops.maxpool3x3s2_fwd(xx[i:i + IMG_BLOCK])
            assert np.array_equal(_h(y).reshape((n,) + ref.shape[1:])[i:i + IMG_BLOCK], ref), '%s: y is not the window maximum' % fn      # a selection: exact
            assert np.array_equal(idx.reshape((n,) + ref.shape[1:])[i:i + IMG_BLOCK].cpu().numpy(), cache['arg']), '%s: idx is not the FIRST maximum' % fn
        self.worst[fn] = 0.0

    def _chk_fte_maxpool3x3s2_fwd(self, x, y, idx, n, h, wd, c, st):
        self._maxpool_fwd('fte_maxpool3x3s2_fwd', x, y, idx, n, h, wd, c)

    def _chk_fte_maxpool3x3s2_fwd_s16(self, x, y, idx, n, h, wd, c, st):
        self._maxpool_fwd('fte_maxpool3x3s2_fwd_s16', x, y, idx, n, h, wd, c)

    def _maxpool_bwd(self, fn, dy, idx, dx, n, h, wd, c, s16):
        ho, wo = ops.same_pads(h, 3, 2)[0], ops.same_pads(wd, 3, 2)[0]
        d, ii = _h(dy).reshape(n, ho, wo, c), idx.reshape(n, ho, wo, c).cpu().numpy().astype(np.int64)
        pt, pl = ops.same_pads(h, 3, 2)[1], ops.same_pads(wd, 3, 2)[1]
        for i in range(0, n, IMG_BLOCK):
            m = min(IMG_BLOCK, n - i)
            ref = ops.maxpool3x3s2_bwd(d[i:i + m], dict(arg=ii[i:i + m], shape=(m, h, wd, c), pads=(pt, pl)))
            self._ok(fn, 'dx', _h(dx).reshape(n, h, wd, c)[i:i + m], ref, stored16=s16)

    def _chk_fte_maxpool3x3s2_bwd(self, dy, idx, dx, n, h, wd, c, st):
        self._maxpool_bwd('fte_maxpool3x3s2_bwd', dy, idx, dx, n, h, wd, c, False)

    def _chk_fte_maxpool3x3s2_bwd_s16(self, dy, idx, dx, n, h, wd, c, st):
        self._maxpool_bwd('fte_maxpool3x3s2_bwd_s16', dy, idx, dx, n, h, wd, c, True)

    def _chk_fte_gap_fwd(self, x, y, n, hw, c, st):
        self._ok('fte_gap_fwd', 'y', _h(y).reshape(n, c), self._nhwc(x, n, hw, c).mean(1))

    def _chk_fte_gap_fwd_s16(self, x, y, n, hw, c, st):
        self._ok('fte_gap_fwd_s16', 'y', _h(y).reshape(n, c), self._nhwc(x, n, hw, c).mean(1))

    def _chk_fte_gap_bwd(self, dy, dx, n, hw, c, st):
        self._ok('fte_gap_bwd', 'dx', self._nhwc(dx, n, hw, c), np.broadcast_to(_h(dy).reshape(n, 1, c) / hw, (n, hw, c)))

    def _chk_fte_gap_bwd_s16(self, dy, dx, n, hw, c, st):
        self._ok('fte_gap_bwd_s16', 'dx', self._nhwc(dx, n, hw, c), np.broadcast_to(_h(dy).reshape(n, 1, c) / hw, (n, hw, c)), stored16=True)

    def _im2col(self, fn, x, cols, n, h, wd, cin, k, stride, kpad, s16):
        """cols[n*ho*wo, kpad]: k ordered (r, s, c) like the HWIO weight rows, zero columns behind k*k*cin (fte.h)"""
        xx = _h(x).reshape(n, h, wd, cin)
        ho, pt, _ = ops.same_pads(h, k, stride)
        wo, pl, _ = ops.same_pads(wd, k, stride)
        got = _h(cols).reshape(n, ho, wo, kpad)
        assert not got[..., k * k * cin:].any(), '%s: padding columns are not zero' % fn
        xp = np.zeros((n, h + k, wd + k, cin))
        xp[:, pt:pt + h, pl:pl + wd] = xx
        for r in range(k):
            for s in range(k):
                ref = xp[:, r:r + (ho - 1) * stride + 1:stride, s:s + (wo - 1) * stride + 1:stride, :]
                self._ok(fn, 'tap (%d, %d)' % (r, s), got[..., (r * k + s) * cin:(r * k + s + 1) * cin], ref, stored16=s16, tol=1e-7 if not s16 else TOL)

    def _chk_fte_im2col_first(self, x, cols, n, h, wd, cin, k, stride, kpad, st):
        self._im2col('fte_im2col_first', x, cols, n, h, wd, cin, k, stride, kpad, False)

    def _chk_fte_im2col_first_s16(self, x, cols, n, h, wd, cin, k, stride, kpad, st):
        self._im2col('fte_im2col_first_s16', x, cols, n, h, wd, cin, k, stride, kpad, True)

    def _chk_fte_dropout_fwd(self, x, mask, y, n, keep, seed, st):
        m = _h(mask)
        assert np.isin(m, (0.0, 1.0)).all() and abs(float(m.mean()) - keep) < 0.02, 'fte_dropout_fwd: the mask is not a Bernoulli(keep) 0 / 1 field'
        self._ok('fte_dropout_fwd', 'y', _h(y), ops.dropout_fwd(_h(x), m, keep), tol=1e-6)

    def _chk_fte_dropout_bwd(self, dy, mask, dx, n, keep, st):
        self._ok('fte_dropout_bwd', 'dx', _h(dx), _h(dy) * _h(mask) / keep, tol=1e-6)

    def _chk_fte_pack_weights_bf16_table(self, params, dst, table, nconv, total, transposed, st):
        """rows {source offset (floats), destination offset (bf16 elements), taps, cin, cout, ...}: HWIO packs, or [tap][cout][cin]"""
        t = table.cpu().numpy().reshape(nconv, -1)
        with np.errstate(invalid='ignore'):      # the pack arena behind the table's ranges holds whatever bits were there
            p, d = _h(params), _h(dst)
        for src, off, taps, cin, cout in t[:, :5]:
            w = _bf(p[src:src + taps * cin * cout]).reshape(taps, cin, cout)
            ref = w.transpose(0, 2, 1) if transposed else w
            assert np.array_equal(d[off:off + taps * cin * cout].reshape(ref.shape), ref), 'fte_pack_weights_bf16_table: pack at %d is not bf16(w)' % off
        self.worst['fte_pack_weights_bf16_table'] = 0.0

    def _chk_fte_gconv3x3_pack_bf16(self, w, wf, wd, c, groups, st):
        """block-diagonal 32-channel slices [slice][tap][col][k] of the grouped filter [group][tap][ic][oc]: forward col = oc, k = ic;
        data gradient col = ic, k = oc, taps mirrored (fte.h:220)"""
        gw, nsl = c // groups, c // 32
        W = _bf(_h(w).ravel()[:groups * 9 * gw * gw]).reshape(groups, 9, gw, gw)
        f = np.zeros((nsl, 9, 32, 32)); d = np.zeros_like(f)
        for sl in range(nsl):
            for j in range(32 // gw):
                q = slice(j * gw, (j + 1) * gw)
                f[sl, :, q, q] = W[sl * (32 // gw) + j].transpose(0, 2, 1)
                d[sl, :, q, q] = W[sl * (32 // gw) + j][::-1]
        assert np.array_equal(_h(wf).ravel()[:f.size].reshape(f.shape), f), 'fte_gconv3x3_pack_bf16: forward pack'
        assert np.array_equal(_h(wd).ravel()[:d.size].reshape(d.shape), d), 'fte_gconv3x3_pack_bf16: data-gradient pack'
        self.worst['fte_gconv3x3_pack_bf16'] = 0.0

    def _chk_fte_act_fwd(self, x, y, n, kind, st):
        pass        # in place on the gate's [n, c] vectors: covered by fte_gemm_nn_act where fused; the unfused form is an A/B hook

    def _chk_fte_act_bwd(self, dy, y, dx, n, kind, st):
        pass        # in place (dx is dy): checked through the dense product that follows it

    def _chk_fte_batch_hard_triplet_fwd_bwd(self, feat, labels, margin, soft, lw, rows, dfeat, n, d, ws, wsb, st):
        per, df = ops.batch_hard_triplet(_h(feat).reshape(n, d), labels.cpu().numpy().astype(np.int64), None if soft else margin)
        self._ok('fte_batch_hard_triplet_fwd_bwd', 'loss rows', _h(rows), per, tol=1e-5)
        self._ok('fte_batch_hard_triplet_fwd_bwd', 'dfeat', _h(dfeat).reshape(n, d), df * lw, tol=2e-5, rell2=True)

    def _chk_fte_center_loss_fwd_bwd_update(self, feat, labels, centers, rows, dfeat, n, d, ncls, alpha, lw, ws, wsb, st):
        pass        # updates `centers` in place: the input table is gone after the call (tests/test_gpu_configs34.py checks it at this shard)

    def _chk_fte_softmax_ce_fwd_bwd(self, logits, labels, rows, dlogits, n, c, ld, gs, st):
        lg = _h(logits).reshape(n, ld)[:, :c]
        y = labels.cpu().numpy().astype(np.int64)
        m = lg.max(1, keepdims=True)
        lse = m[:, 0] + np.log(np.exp(lg - m).sum(1))
        self._ok('fte_softmax_ce_fwd_bwd', 'loss rows', _h(rows), lse - lg[np.arange(n), y], tol=1e-5)
        p = np.exp(lg - lse[:, None])
        p[np.arange(n), y] -= 1
        self._ok('fte_softmax_ce_fwd_bwd', 'dlogits', _h(dlogits).reshape(n, ld)[:, :c], p * gs, tol=2e-5, rell2=True)


# kernel FAMILIES (symbol names without template arguments, as rocprofv3 prints them) that each entry point's checked call ran, for
# the kernels that leave no launch record (everything outside the MFMA families): read off the launchers in csrc/layers.hip /
# kernels.hip.  tests/test_gpu_fullshard.py requires every symbol with >= 1 % of a profiled step's kernel time to be covered by a
# launch record of a checked call or by this table.
ENTRY_KERNELS = {
    'fte_conv2d_bn_fwd': ['bn_finalize_kernel', 'bn_finalize_wide_kernel'],
    'fte_gconv3x3_bn_fwd_bf16_s16': ['gconv3x3_mfma16_win_kernel', 'gconv3x3_mfma16_kernel', 'bn_finalize_kernel', 'bn_finalize_wide_kernel'],
    'fte_gconv3x3_bf16_s16': ['gconv3x3_mfma16_win_kernel', 'gconv3x3_mfma16_kernel'],
    'fte_gconv3x3_wgrad_bf16_s16': ['gconv3x3_wgrad_mfma16_kernel', 'gconv_wgrad16_reduce_kernel'],
    'fte_bn_apply': ['bn_apply_kernel'],
    'fte_bn_train_fwd': ['bn_stats_v4_kernel', 'bn_stats_kernel', 'bn_finalize_kernel', 'bn_apply_kernel'],
    'fte_bn_train_fwd_s16': ['bn_stats_v4_kernel', 'bn_finalize_kernel', 'bn_apply_kernel'],
    'fte_bn_train_stats': ['bn_stats_v4_kernel', 'bn_stats_kernel', 'bn_finalize_kernel'],
    'fte_bn_train_bwd': ['bn_bwd_reduce_v4_kernel', 'bn_bwd_reduce_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_finalize_wide_kernel', 'bn_bwd_apply_kernel'],
    'fte_bn_train_bwd_s16': ['bn_bwd_reduce_v4_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_finalize_wide_kernel', 'bn_bwd_apply_kernel'],
    'fte_bn_train_bwd_res': ['bn_bwd_reduce_v4_kernel', 'bn_bwd_reduce_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_apply_kernel'],
    'fte_bn_train_bwd_zmask': ['bn_bwd_reduce_v4_kernel', 'bn_bwd_reduce_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_apply_kernel'],
    'fte_bn_infer_fwd': ['bn_infer_coef_kernel', 'bn_apply_kernel'], 'fte_bn_infer_fwd_s16': ['bn_infer_coef_kernel', 'bn_apply_kernel'],
    'fte_conv2d_wgrad': ['reduce_slabs_kernel', 'reduce_rows_kernel', 'reduce_rows_q_kernel'],
    'fte_conv2d_wgrad16': ['reduce_slabs_kernel', 'reduce_rows_kernel', 'reduce_rows_q_kernel'],
    'fte_conv2d_dgrad': ['reduce_rows_kernel', 'reduce_rows_q_kernel'], 'fte_conv2d_dgrad_s16': ['reduce_rows_kernel', 'reduce_rows_q_kernel'],
    'fte_gemm_nn': ['reduce_slabs_kernel', 'reduce_rows_kernel'], 'fte_gemm_nn_act': ['reduce_slabs_kernel', 'reduce_rows_kernel'],
    'fte_gemm_nt': ['reduce_slabs_kernel', 'reduce_rows_kernel'], 'fte_gemm_tn': ['reduce_slabs_kernel', 'reduce_rows_kernel'],
    'fte_dwconv3x3_fwd': ['dwconv3x3_win_kernel', 'dwconv3x3_kernel'],
    'fte_dwconv3x3_dgrad': ['dwconv3x3_win_kernel', 'dwconv3x3_kernel', 'dwconv3x3_dgrad_s2_kernel'],
    'fte_dwconv3x3_wgrad': ['dwconv3x3_wgrad_kernel', 'reduce_rows_kernel', 'reduce_rows_q_kernel', 'reduce_slabs_kernel'],
    'fte_dense_small': ['dense_small_kernel'],
    'fte_se_squeeze': ['se_squeeze_kernel'], 'fte_se_apply_fwd': ['se_apply_kernel'], 'fte_se_bwd_gate': ['se_bwd_gate_kernel'],
    'fte_se_bn_bwd_coef': ['se_bn_coef_kernel'], 'fte_se_bn_bwd_apply': ['se_bn_apply_kernel'],
    'fte_channel_scale_fwd': ['chscale_fwd_kernel'], 'fte_channel_scale_fwd_s16': ['chscale_fwd_kernel'],
    'fte_channel_scale_bwd': ['chscale_bwd_kernel'], 'fte_channel_scale_bwd_s16': ['chscale_bwd_kernel'],
    'fte_channel_scale_bwd_apply_s16': ['chscale_bwd_apply_kernel'],
    'fte_relu_bwd': ['relu_bwd_kernel'], 'fte_relu_bwd_s16': ['relu_bwd_kernel'],
    'fte_channel_gather': ['channel_gather_kernel', 'channel_gather_lds_kernel'], 'fte_channel_gather_s16': ['channel_gather_kernel', 'channel_gather_lds_kernel'],
    'fte_channel_gather_affine': ['channel_gather_affine_kernel', 'channel_gather_lds_kernel'],
    'fte_channel_gather_affine_s16': ['channel_gather_affine_kernel', 'channel_gather_lds_kernel'],
    'fte_maxpool3x3s2_fwd': ['maxpool_fwd_kernel'], 'fte_maxpool3x3s2_fwd_s16': ['maxpool_fwd_kernel'],
    'fte_maxpool3x3s2_bwd': ['maxpool_bwd_kernel', 'maxpool_bwd_even_kernel'], 'fte_maxpool3x3s2_bwd_s16': ['maxpool_bwd_kernel', 'maxpool_bwd_even_kernel'],
    'fte_gap_fwd': ['gap_fwd_kernel'], 'fte_gap_fwd_s16': ['gap_fwd_kernel'], 'fte_gap_bwd': ['gap_bwd_kernel'], 'fte_gap_bwd_s16': ['gap_bwd_kernel'],
    'fte_im2col_first': ['im2col_first_kernel', 'im2col_first_rows_kernel'], 'fte_im2col_first_s16': ['im2col_first_kernel', 'im2col_first_rows_kernel'],
    'fte_dropout_fwd': ['dropout_fwd_kernel'], 'fte_dropout_bwd': ['scale_mask_kernel'],
    'fte_pack_weights_bf16_table': ['pack_weights_table_kernel', 'pack_weights_tiles_kernel'], 'fte_gconv3x3_pack_bf16': ['gconv_pack16_kernel'],
    'fte_batch_hard_triplet_fwd_bwd': ['triplet_dist_kernel', 'triplet_mine_kernel', 'triplet_grad_kernel'],
    'fte_softmax_ce_fwd_bwd': ['softmax_ce_reg_kernel', 'softmax_ce_kernel'],
}
