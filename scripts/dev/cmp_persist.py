"""Debug: conv forward / dgrad s16 outputs of the persistent kernel against the per-tile kernel (two child processes, one per mode)."""
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
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    w = torch.randn(3, 3, c, c, device='cuda', generator=g) * 0.05
    w16 = torch.empty(w.shape, **i16); w16t = torch.empty(3, 3, c, c, **i16)
    _lib.call('fte_pack_weights_bf16', w, w16, w16t, 3, c, c, st)
    al = torch.rand(c, device='cuda', generator=g)
    bias = torch.randn(c, device='cuda', generator=g)
    x16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    r16 = torch.randn(B, hw, hw, c, device='cuda', generator=g).bfloat16().view(torch.int16)
    z16 = torch.zeros(B, hw, hw, c, **i16); y16 = torch.zeros_like(z16)
    raw16 = torch.zeros_like(z16); dzp16 = torch.zeros_like(z16)
    da = torch.zeros(c, device='cuda'); db = torch.zeros(c, device='cuda')
    _lib.call('fte_prof_enable', 1)
    _lib.call('fte_conv2d_fwd_s16', x16, w16t, bias, al, r16, z16, y16, None, None, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    _lib.call('fte_conv2d_dgrad_s16', x16, w16, r16, z16, al, raw16, dzp16, da, db, B, hw, hw, c, c, 3, 1, ws, wsb, st)
    torch.cuda.synchronize()
    print('   kernels:', [(r[5], r[3]) for r in _lib.prof_records(shapes=True)])
    torch.save({k: v.cpu() for k, v in dict(z=z16, y=y16, raw=raw16, dz=dzp16, da=da, db=db).items()}, out)
    sys.exit(0)
import torch
hw, c, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
outs = []
for mode in ('0', '1'):
    out = '/tmp/cmp_persist_%s.pt' % mode
    env = dict(os.environ, FTE_IGEMM16_PERSIST=mode)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), 'child', str(hw), str(c), str(B), out], env=env)
    outs.append(torch.load(out))
a, b = outs
for k in ('z', 'y', 'raw', 'dz'):
    x = a[k].view(torch.bfloat16).float().reshape(-1, c); y = b[k].view(torch.bfloat16).float().reshape(-1, c)
    bad = (a[k].reshape(-1, c) != b[k].reshape(-1, c))
    print(k, 'mismatching elements', int(bad.sum()), 'of', bad.numel(), 'max abs diff', float((x - y).abs().max()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('   rows', rows[:12].tolist(), '... n', len(rows), ' cols', cols[:40].tolist(), '... n', len(cols))
        r0 = int(rows[0])
        print('   row', r0, 'ref ', x[r0, :16].tolist()); print('   row', r0, 'got ', y[r0, :16].tolist())
for k in ('da', 'db'):
    d = (a[k] - b[k]).abs().max(); print(k, 'max abs diff', float(d), 'max ref', float(a[k].abs().max()))
