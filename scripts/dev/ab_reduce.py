"""Timing / equality: the split-K slab reduction (reduce_slabs_kernel, FTE_REDUCE_SLABS=1 default) against reduce_rows_kernel (=0) on the
filter gradients of the SphereNet layers, bf16-storage (resident kernel) and fp32 entry points.  Usage: ab_reduce.py [batch]"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LAYERS = [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from tf_face_toolbox_amd import _lib
    B, out = int(sys.argv[2]), sys.argv[3]
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for hw, cin, cout in LAYERS:
        g = torch.Generator(device='cuda'); g.manual_seed(3)
        x = torch.randn(B, hw, hw, cin, device='cuda', generator=g)
        dz = torch.randn(B, hw, hw, cout, device='cuda', generator=g)
        x16, dz16 = x.bfloat16().view(torch.int16), dz.bfloat16().view(torch.int16)
        nb = _lib.query('fte_conv2d_wgrad_ws_bytes', B, hw, hw, cin, cout, 3, 1)
        ws = torch.empty(max(nb, 4) // 4 + 16, dtype=torch.float32, device='cuda')
        for name, f in (('bf16 storage', lambda dw: _lib.call('fte_conv2d_wgrad16', x16, dz16, dw, B, hw, hw, cin, cout, 3, 1, ws, ws.numel() * 4, st)),
                        ('fp32', lambda dw: _lib.call('fte_conv2d_wgrad', x, dz, dw, B, hw, hw, cin, cout, 3, 1, ws, ws.numel() * 4, st))):
            dw = torch.zeros(3, 3, cin, cout, device='cuda')
            f(dw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f(dw)
            e1.record(); torch.cuda.synchronize()
            print('   %-12s %dx%d %d->%d B=%d: %.4f ms' % (name, hw, hw, cin, cout, B, e0.elapsed_time(e1) / 10))
            res[(name, hw)] = dw.cpu()
    torch.save(res, out)
    sys.exit(0)
import torch
B = sys.argv[1] if len(sys.argv) > 1 else '512'
outs = []
for mode in ('0', '1'):
    print('FTE_REDUCE_SLABS=%s' % mode)
    out = '/tmp/ab_reduce_%s.pt' % mode
    subprocess.check_call([sys.executable, os.path.abspath(__file__), 'child', B, out], env=dict(os.environ, FTE_REDUCE_SLABS=mode))
    outs.append(torch.load(out))
for k in outs[0]:
    a, b = outs[0][k].double(), outs[1][k].double()
    print(k, 'rel-L2 %.2e' % float((a - b).norm() / a.norm()))
