"""Child process of tests/test_gpu_tiles.py: runs fte_conv3x3_{fwd,dgrad,wgrad} through the C ABI on one case list and compares
every result with the float64 oracle (oracle/ops.py), image block by image block so that the production sizes of the headline run
(hundreds of 56x56 / 28x28 images) fit the host.  The tile / split-K hooks of csrc/api.hip are read ONCE per process
(FTE_WGRAD_TILE, FTE_WGRAD_SPLIT_MAJOR, ...), hence a process per environment.

    python tests/tile_worker.py '[["wgrad", n, h, w, cin, cout, stride], ["fwd", ...], ["dgrad", ...]]'

Prints one JSON line: {"cases": [{"case": [...], "symbols": [...kernel symbols the launch records name...], "splits": [...],
"errors": {...}}], "ok": true|false}.  Tolerances are tests/util_gpu.py's (2e-5 max-abs / rel-L2 against float64)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ops                                     # noqa: E402
from tf_face_toolbox_amd import _lib                       # noqa: E402
from util_gpu import call, query, stream, ws, TOL_MAXABS, TOL_RELL2      # noqa: E402

BLOCK = 8           # images per oracle block


def _maxabs(got, ref):
    return float(np.abs(got - ref).max()), float(np.abs(ref).max())


def _records():
    torch.cuda.synchronize()
    recs = _lib.prof_records(shapes=True)
    return sorted({r[5] for r in recs}), [r[0][4] for r in recs]


def run_fwd(n, h, w, cin, cout, stride, r):
    x = r.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (r.standard_normal((3, 3, cin, cout), dtype=np.float32) * 0.05).astype(np.float32)
    b = r.standard_normal(cout, dtype=np.float32)
    al = (0.25 + 0.1 * r.standard_normal(cout, dtype=np.float32)).astype(np.float32)
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    res = r.standard_normal((n, ho, wo, cout), dtype=np.float32)
    z = torch.empty(n, ho, wo, cout, device='cuda'); y = torch.empty_like(z)
    wsb, nb = ws(query('fte_conv3x3_fwd_ws_bytes', n, h, w, cin, cout, stride))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv3x3_fwd', torch.from_numpy(x).cuda(), torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(al).cuda(),
         torch.from_numpy(res).cuda(), z, y, n, h, w, cin, cout, stride, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    zg, yg = z.cpu().numpy(), y.cpu().numpy()
    ez = ey = sz = sy = 0.0
    w64, b64, al64 = wt.astype(np.float64), b.astype(np.float64), al.astype(np.float64)
    for i in range(0, n, BLOCK):
        zr = ops.conv2d_fwd(x[i:i + BLOCK].astype(np.float64), w64, stride, b64)
        yr = ops.prelu_fwd(zr, al64) + res[i:i + BLOCK]
        e, s = _maxabs(zg[i:i + BLOCK], zr); ez = max(ez, e); sz = max(sz, s)
        e, s = _maxabs(yg[i:i + BLOCK], yr); ey = max(ey, e); sy = max(sy, s)
    errs = {'z_maxabs_rel': ez / sz, 'y_maxabs_rel': ey / sy}
    return syms, splits, errs, all(v <= TOL_MAXABS for v in errs.values())


def run_dgrad(n, h, w, cin, cout, stride, r):
    wt = (r.standard_normal((3, 3, cin, cout), dtype=np.float32) * 0.05).astype(np.float32)
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    dz = r.standard_normal((n, ho, wo, cout), dtype=np.float32)
    addin = r.standard_normal((n, h, w, cin), dtype=np.float32)
    zprev = r.standard_normal((n, h, w, cin), dtype=np.float32)
    zprev[0, 0, 0, :4] = 0.0                                  # the z == 0 sub-gradient (slope alpha / 2)
    alp = (0.25 + 0.1 * r.standard_normal(cin, dtype=np.float32)).astype(np.float32)
    raw = torch.empty(n, h, w, cin, device='cuda'); dzp = torch.empty_like(raw)
    da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_dgrad_ws_bytes', n, h, w, cin, cout, stride))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv3x3_dgrad', torch.from_numpy(dz).cuda(), torch.from_numpy(wt).cuda(), torch.from_numpy(addin).cuda(),
         torch.from_numpy(zprev).cuda(), torch.from_numpy(alp).cuda(), raw, dzp, da, db, n, h, w, cin, cout, stride, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    rawg, dzpg = raw.cpu().numpy(), dzp.cpu().numpy()
    w64, al64 = wt.astype(np.float64), alp.astype(np.float64)
    er = ed = sr = sd = 0.0
    da_ref = np.zeros(cin); db_ref = np.zeros(cin)
    xshape = np.zeros((1, h, w, cin))
    for i in range(0, n, BLOCK):
        m = min(BLOCK, n - i)
        dx, _ = ops.conv2d_bwd(np.broadcast_to(xshape, (m, h, w, cin)), w64, dz[i:i + m].astype(np.float64), stride, need_dw=False)
        g = dx + addin[i:i + m]
        dzr, dar = ops.prelu_bwd(zprev[i:i + m].astype(np.float64), al64, g)
        da_ref += dar; db_ref += dzr.sum(axis=(0, 1, 2))
        e, s = _maxabs(rawg[i:i + m], g); er = max(er, e); sr = max(sr, s)
        e, s = _maxabs(dzpg[i:i + m], dzr); ed = max(ed, e); sd = max(sd, s)

    def rl2(got, ref):
        return float(np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
    errs = {'raw_maxabs_rel': er / sr, 'dzprev_maxabs_rel': ed / sd,
            'dalpha_rell2': rl2(da.cpu().numpy().astype(np.float64), da_ref), 'dbias_rell2': rl2(db.cpu().numpy().astype(np.float64), db_ref)}
    ok = errs['raw_maxabs_rel'] <= TOL_MAXABS and errs['dzprev_maxabs_rel'] <= TOL_MAXABS and \
        errs['dalpha_rell2'] <= TOL_RELL2 and errs['dbias_rell2'] <= TOL_RELL2
    return syms, splits, errs, ok


def run_wgrad(n, h, w, cin, cout, stride, r):
    x = r.standard_normal((n, h, w, cin), dtype=np.float32)
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    dz = r.standard_normal((n, ho, wo, cout), dtype=np.float32)
    dw = torch.empty(3, 3, cin, cout, device='cuda')
    wsb, nb = ws(query('fte_conv3x3_wgrad_ws_bytes', n, h, w, cin, cout, stride))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv3x3_wgrad', torch.from_numpy(x).cuda(), torch.from_numpy(dz).cuda(), dw, n, h, w, cin, cout, stride, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    ref = np.zeros((3, 3, cin, cout))
    wz = np.zeros((3, 3, cin, cout))
    for i in range(0, n, BLOCK):
        _, d = ops.conv2d_bwd(x[i:i + BLOCK].astype(np.float64), wz, dz[i:i + BLOCK].astype(np.float64), stride, need_dx=False)
        ref += d
    e, s = _maxabs(dw.cpu().numpy().astype(np.float64), ref)
    errs = {'dw_maxabs_rel': e / s}
    return syms, splits, errs, e / s <= TOL_MAXABS


def _bf(a):
    """float32 array -> (bf16-exact float32 values, their int16 bit patterns on the GPU)"""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda().bfloat16()
    return t.float().cpu().numpy(), t.view(torch.int16)


def _stored_ok(got16, ref64, scale):
    """a bf16-STORED result against the float64 value it rounds: within one bf16 step of the reference (fp32 accumulation may tip a
    value across a rounding boundary) -- |got - ref| <= 2^-8 |ref| + 2e-5 scale; returns (worst excess ratio, ok)"""
    got = got16.view(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref64)
    lim = np.abs(ref64) * 2.0 ** -8 + TOL_MAXABS * scale
    return float((err / lim).max()), bool((err <= lim).all())


def run_s16fwd(n, h, w, cin, cout, stride, r):
    """fte_conv2d_fwd_s16 (bf16 x / shortcut in, bf16 z / y out): the persistent bf16 kernels at the sizes that select them"""
    assert stride == 1
    x, x16 = _bf(r.standard_normal((n, h, w, cin), dtype=np.float32))
    wt = (r.standard_normal((3, 3, cin, cout), dtype=np.float32) * 0.05).astype(np.float32)
    wd = torch.from_numpy(wt).cuda()
    w16 = torch.empty(3, 3, cin, cout, dtype=torch.int16, device='cuda'); w16t = torch.empty(3, 3, cout, cin, dtype=torch.int16, device='cuda')
    call('fte_pack_weights_bf16', wd, w16, w16t, 3, cin, cout, stream())
    wb = wd.bfloat16().float().cpu().numpy().astype(np.float64)
    b = r.standard_normal(cout, dtype=np.float32)
    al = (0.25 + 0.1 * r.standard_normal(cout, dtype=np.float32)).astype(np.float32)
    res, res16 = _bf(r.standard_normal((n, h, w, cout), dtype=np.float32))
    z16 = torch.empty(n, h, w, cout, dtype=torch.int16, device='cuda'); y16 = torch.empty_like(z16)
    wsb, nb = ws(query('fte_conv2d_fwd_ws_bytes', n, h, w, cin, cout, 3, 1))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv2d_fwd_s16', x16, w16t, torch.from_numpy(b).cuda(), torch.from_numpy(al).cuda(), res16, z16, y16, None, None,
         n, h, w, cin, cout, 3, 1, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    b64, al64 = b.astype(np.float64), al.astype(np.float64)
    wz = wy = 0.0
    ok = True
    for i in range(0, n, BLOCK):
        zr = ops.conv2d_fwd(x[i:i + BLOCK].astype(np.float64), wb, 1, b64)
        yr = ops.prelu_fwd(zr, al64) + res[i:i + BLOCK]
        e, o = _stored_ok(z16[i:i + BLOCK], zr, np.abs(zr).max()); wz = max(wz, e); ok = ok and o
        e, o = _stored_ok(y16[i:i + BLOCK], yr, np.abs(yr).max()); wy = max(wy, e); ok = ok and o
    return syms, splits, {'z_worst_over_limit': wz, 'y_worst_over_limit': wy}, ok


def run_s16dgrad(n, h, w, cin, cout, stride, r):
    """fte_conv2d_dgrad_s16 (bf16 dz / skip gradient / previous z in, bf16 raw / dz out, fp32 dalpha / dbias sums)"""
    assert stride == 1
    wt = (r.standard_normal((3, 3, cin, cout), dtype=np.float32) * 0.05).astype(np.float32)
    wd = torch.from_numpy(wt).cuda()
    w16 = torch.empty(3, 3, cin, cout, dtype=torch.int16, device='cuda'); w16t = torch.empty(3, 3, cout, cin, dtype=torch.int16, device='cuda')
    call('fte_pack_weights_bf16', wd, w16, w16t, 3, cin, cout, stream())
    wb = wd.bfloat16().float().cpu().numpy().astype(np.float64)
    dz, dz16 = _bf(r.standard_normal((n, h, w, cout), dtype=np.float32))
    addin, add16 = _bf(r.standard_normal((n, h, w, cin), dtype=np.float32))
    zp = r.standard_normal((n, h, w, cin), dtype=np.float32)
    zp[0, 0, 0, :4] = 0.0
    zprev, zp16 = _bf(zp)
    alp = (0.25 + 0.1 * r.standard_normal(cin, dtype=np.float32)).astype(np.float32)
    raw16 = torch.empty(n, h, w, cin, dtype=torch.int16, device='cuda'); dzp16 = torch.empty_like(raw16)
    da = torch.empty(cin, device='cuda'); db = torch.empty(cin, device='cuda')
    wsb, nb = ws(query('fte_conv2d_dgrad_ws_bytes', n, h, w, cin, cout, 3, 1))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv2d_dgrad_s16', dz16, w16, add16, zp16, torch.from_numpy(alp).cuda(), raw16, dzp16, da, db, n, h, w, cin, cout, 3, 1, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    al64 = alp.astype(np.float64)
    da_ref = np.zeros(cin); db_ref = np.zeros(cin)
    xshape = np.zeros((1, h, w, cin))
    wr = wd_ = 0.0
    ok = True
    for i in range(0, n, BLOCK):
        m = min(BLOCK, n - i)
        dx, _ = ops.conv2d_bwd(np.broadcast_to(xshape, (m, h, w, cin)), wb, dz[i:i + m].astype(np.float64), 1, need_dw=False)
        g = dx + addin[i:i + m]
        dzr, dar = ops.prelu_bwd(zprev[i:i + m].astype(np.float64), al64, g)
        da_ref += dar; db_ref += dzr.sum(axis=(0, 1, 2))
        e, o = _stored_ok(raw16[i:i + m], g, np.abs(g).max()); wr = max(wr, e); ok = ok and o
        e, o = _stored_ok(dzp16[i:i + m], dzr, np.abs(dzr).max()); wd_ = max(wd_, e); ok = ok and o

    def rl2(got, ref):
        return float(np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
    errs = {'raw_worst_over_limit': wr, 'dz_worst_over_limit': wd_,
            'dalpha_rell2': rl2(da.cpu().numpy().astype(np.float64), da_ref), 'dbias_rell2': rl2(db.cpu().numpy().astype(np.float64), db_ref)}
    ok = ok and errs['dalpha_rell2'] <= TOL_RELL2 and errs['dbias_rell2'] <= TOL_RELL2
    return syms, splits, errs, ok


def run_s16wgrad1(n, h, w, cin, cout, stride, r):
    """fte_conv2d_wgrad16 of a 1x1 conv: the pointwise resident kernel (wgrad16p_kernel) against the float64 product of the same bf16 inputs"""
    x, x16 = _bf(r.standard_normal((n, h, w, cin), dtype=np.float32))
    dz, dz16 = _bf(r.standard_normal((n, h, w, cout), dtype=np.float32))
    dw = torch.empty(1, 1, cin, cout, device='cuda')
    wsb, nb = ws(query('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, 1, stride))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv2d_wgrad16', x16, dz16, dw, n, h, w, cin, cout, 1, stride, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    ref = x.reshape(-1, cin).astype(np.float64).T @ dz.reshape(-1, cout).astype(np.float64)
    e, s = _maxabs(dw.cpu().numpy().astype(np.float64).reshape(cin, cout), ref)
    return syms, splits, {'dw_maxabs_rel': e / s}, e / s <= TOL_MAXABS


def run_s16wgrad(n, h, w, cin, cout, stride, r):
    """fte_conv2d_wgrad16 (bf16 x and dz in, fp32 dw out): the resident kernel of wgrad16.hip at the sizes that select it"""
    x, x16 = _bf(r.standard_normal((n, h, w, cin), dtype=np.float32))
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    dz, dz16 = _bf(r.standard_normal((n, ho, wo, cout), dtype=np.float32))
    dw = torch.empty(3, 3, cin, cout, device='cuda')
    wsb, nb = ws(query('fte_conv2d_wgrad_ws_bytes', n, h, w, cin, cout, 3, stride))
    _lib.query('fte_prof_enable', 1)
    call('fte_conv2d_wgrad16', x16, dz16, dw, n, h, w, cin, cout, 3, stride, wsb, nb, stream())
    _lib.query('fte_prof_enable', 0)
    syms, splits = _records()
    ref = np.zeros((3, 3, cin, cout))
    wz = np.zeros((3, 3, cin, cout))
    for i in range(0, n, BLOCK):
        _, d = ops.conv2d_bwd(x[i:i + BLOCK].astype(np.float64), wz, dz[i:i + BLOCK].astype(np.float64), stride, need_dx=False)
        ref += d
    e, s = _maxabs(dw.cpu().numpy().astype(np.float64), ref)
    return syms, splits, {'dw_maxabs_rel': e / s}, e / s <= TOL_MAXABS


def main():
    cases = json.loads(sys.argv[1])
    out, ok_all = [], True
    for ci, c in enumerate(cases):
        op, dims = c[0], [int(v) for v in c[1:]]
        r = np.random.default_rng(100 + ci)
        syms, splits, errs, ok = {'fwd': run_fwd, 'dgrad': run_dgrad, 'wgrad': run_wgrad, 's16fwd': run_s16fwd, 's16dgrad': run_s16dgrad, 's16wgrad': run_s16wgrad, 's16wgrad1': run_s16wgrad1}[op](*dims, r)
        out.append({'case': c, 'symbols': syms, 'splits': splits, 'errors': errs, 'ok': bool(ok)})
        ok_all = ok_all and ok
        torch.cuda.empty_cache()
    print(json.dumps({'cases': out, 'ok': bool(ok_all)}))
    sys.exit(0 if ok_all else 1)


if __name__ == '__main__':
    main()
