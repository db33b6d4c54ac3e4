"""Debug: bf16-storage forward / data gradient with the row-coalesced epilogue (FTE_IGEMM16_STG=1) against the register epilogue (=0)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from tf_face_toolbox_amd import _lib
    hw, c, B, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    _lib.set_mfma_dtype('bf16s')
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(64 << 20, dtype=torch.float32, device='cuda'); wsb = ws.numel() * 4
    i16 = dict(dtype=torch.int16, device='cuda')
    g = torch.Generator(device='cuda'); g.manual_seed(hw)
    w = torch.randn(3, 3, c, c, device='cuda', generator=g) * 0.05
    w16 = torch.empty(w.shape, **i16); w16t = torch.empty(3, 3, c, c, **i16)
    _lib.call('fte_pack_weights_bf16', w, w16, w16t, 3, c, c, st)
    al = torch.rand(c, device='cuda', generator=g) * 0.5
    bias = torch.randn(c, device='cuda', generator=g)
    x16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    r16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    z16 = torch.zeros(B, hw, hw, c, **i16); y16 = torch.zeros_like(z16)
    raw16 = torch.zeros_like(z16); dzp16 = torch.zeros_like(z16)
    da = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    _lib.call('fte_conv2d_fwd_s16', x16, w16t, bias, al, r16, z16, y16, None, None, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    _lib.call('fte_conv2d_dgrad_s16', x16, w16, r16, z16, al, raw16, dzp16, da, db, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    torch.cuda.synchronize()
    f = lambda t: t.view(torch.bfloat16).float().cpu().reshape(-1, c)
    torch.save(dict(z=f(z16), y=f(y16), raw=f(raw16), dz=f(dzp16), da=da.cpu(), db=db.cpu()), out)
    sys.exit(0)
import torch
for spec in sys.argv[1:]:
    hw, c, B = [int(v) for v in spec.split(',')]
    res = []
    for mode in ('0', '1'):
        out = '/tmp/cmp_stg_%s.pt' % mode
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'child', str(hw), str(c), str(B), out], env=dict(os.environ, FTE_IGEMM16_STG=mode))
        res.append(torch.load(out))
    a, b = res
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        msg = '%s %s: max abs diff %.3e of %.3e' % (spec, k, float(d.max()), float(a[k].abs().max()))
        if d.dim() == 2 and float(d.max()) > 0:
            bad = d > 0
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            msg += ' | bad rows %d (first %s) bad cols %d (first %s)' % (len(rows), rows[:12].tolist(), len(cols), cols[:12].tolist())
            r0 = int(rows[0])
            msg += ' | row %d: old %s new %s' % (r0, a[k][r0, :6].tolist(), b[k][r0, :6].tolist())
        print(msg)
