"""Prints, per tensor, the HIP path's error vs the float64 oracle next to the error of the SAME
oracle evaluated in float32 (fp32's own noise floor).  Used to choose the stated tolerances."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ops, spherenet as osn
from tf_face_toolbox_amd import net_select

def rell2(a, b):
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))

def run(name, fmt, n, h, w, ch, ncls, seed=21):
    p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
    head = 'asoftmax' if 'ASoftmax' in name else 'softmax'
    lam = ops.asoftmax_lambda(0)
    l64, g64, ex64 = osn.loss_and_grads(p, x, y, 0.0, fmt, head, lam)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    l32, g32, ex32 = osn.loss_and_grads(p32, x.astype(np.float32), y, 0.0, fmt, head, np.float32(lam))
    net = net_select(name, fmt, 5e-4); net.build(h, w, ch, ncls, 'cuda'); net.load_params(p)
    xd = torch.tensor(x, dtype=torch.float32, device='cuda'); yd = torch.tensor(y, dtype=torch.int32, device='cuda')
    lg = net.forward(xd, yd, num_classes=ncls) if net.needs_labels else net.forward(xd, num_classes=ncls)
    net.loss_function('T', yd, **lg); net.backward(); torch.cuda.synchronize()
    print('== %s %s n=%d %dx%dx%d C=%d' % (name, fmt, n, h, w, ch, ncls))
    print('  embedding: hip %.2e  f32 %.2e | logits: hip %.2e f32 %.2e' % (
        rell2(net.emb.cpu().numpy().astype(np.float64), ex64['embedding']), rell2(ex32['embedding'].astype(np.float64), ex64['embedding']),
        rell2(lg['logits'].cpu().numpy().astype(np.float64), ex64['logits']), rell2(ex32['logits'].astype(np.float64), ex64['logits'])))
    worst = (0, None)
    for k in p:
        eh = rell2(net.get_variable(k, net.grads).cpu().numpy().astype(np.float64), g64[k])
        e3 = rell2(g32[k].astype(np.float64), g64[k])
        if eh > worst[0]: worst = (eh, k)
        if eh > 3e-5 or e3 > 3e-5:
            print('  %-55s hip %.2e   f32-oracle %.2e' % (k, eh, e3))
    print('  worst hip:', worst)

run('SphereNet', 'NCHW', 2, 112, 112, 3, 1000)
run('SphereNet-ASoftmax', 'NCHW', 4, 32, 32, 3, 10, seed=11)
run('SphereNet', 'NCHW', 8, 112, 112, 3, 1000)
