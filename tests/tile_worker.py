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


def main():
    cases = json.loads(sys.argv[1])
    out, ok_all = [], True
    for ci, c in enumerate(cases):
        op, dims = c[0], [int(v) for v in c[1:]]
        r = np.random.default_rng(100 + ci)
        syms, splits, errs, ok = {'fwd': run_fwd, 'dgrad': run_dgrad, 'wgrad': run_wgrad}[op](*dims, r)
        out.append({'case': c, 'symbols': syms, 'splits': splits, 'errors': errs, 'ok': bool(ok)})
        ok_all = ok_all and ok
        torch.cuda.empty_cache()
    print(json.dumps({'cases': out, 'ok': bool(ok_all)}))
    sys.exit(0 if ok_all else 1)


if __name__ == '__main__':
    main()
