"""Debug / timing: fte_conv2d_wgrad16 with the resident kernels (wgrad16.hip; 3x3 and, spec "HW,CIN,COUT,B,1", 1x1) against the per-tile
plan (FTE_WGRAD16_RESIDENT=0 / FTE_WGRAD16_POINTWISE=0)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from tf_face_toolbox_amd import _lib
    hw, cin, cout, B, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    ks = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    x16 = torch.randn(B, hw, hw, cin, device='cuda', generator=g).bfloat16().view(torch.int16)
    dz16 = torch.randn(B, hw, hw, cout, device='cuda', generator=g).bfloat16().view(torch.int16)
    nb = _lib.query('fte_conv2d_wgrad_ws_bytes', B, hw, hw, cin, cout, ks, 1)
    ws = torch.empty(max(nb, 4) // 4 + 16, dtype=torch.float32, device='cuda')
    dw = torch.zeros(ks, ks, cin, cout, device='cuda')
    f = lambda: _lib.call('fte_conv2d_wgrad16', x16, dz16, dw, B, hw, hw, cin, cout, ks, 1, ws, ws.numel() * 4, st)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print('   %dx%d %d->%d k%d B=%d: %.4f ms  %.1f TF' % (hw, hw, cin, cout, ks, B, t, 2.0 * B * hw * hw * ks * ks * cin * cout / t / 1e9))
    torch.save(dw.cpu(), out)
    sys.exit(0)
import torch
for spec in sys.argv[1:]:
    f_ = [int(v) for v in spec.split(',')]
    hw, cin, cout, B = f_[:4]
    ks = f_[4] if len(f_) > 4 else 3
    res = []
    for mode in ('0', '1'):
        out = '/tmp/cmp_wg_%s.pt' % mode
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'child', str(hw), str(cin), str(cout), str(B), out, str(ks)],
                              env=dict(os.environ, FTE_WGRAD16_RESIDENT=mode, FTE_WGRAD16_POINTWISE=mode))
        res.append(torch.load(out).double())
    a, b = res
    print('%s: rel-L2 resident vs per-tile %.3e, max abs %.3e of %.3e' % (spec, float((a - b).norm() / a.norm()), float((a - b).abs().max()), float(a.abs().max())))
    if float((a - b).norm() / a.norm()) > 1e-4:
        d = (a - b).abs()
        bad = (d > 1e-3 * a.abs().max())
        print('   bad taps', bad.any(3).any(2).flatten().tolist(), ' bad cin count', int(bad.any(3).any(0).any(0).sum()), ' bad cout count', int(bad.any(2).any(0).any(0).sum()))
