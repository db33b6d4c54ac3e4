import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ops, spherenet as osn
from tf_face_toolbox_amd import net_select
def rell2(a, b): return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
n,h,w,ch,ncls,seed = 4,32,32,3,10,11
p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
rng = np.random.default_rng(seed + 2)
x = rng.uniform(-1, 1, (n, h, w, ch)); y = rng.integers(0, ncls, n)
tr = {}
l64, g64, ex64 = osn.loss_and_grads(p, x, y, 0.0, 'NCHW', 'softmax', None, trace=tr)
net = net_select('SphereNet', 'NCHW', 5e-4); net.build(h, w, ch, ncls, 'cuda'); net.load_params(p)
net._trace_dz = {}
xd = torch.tensor(x, dtype=torch.float32, device='cuda'); yd = torch.tensor(y, dtype=torch.int32, device='cuda')
lg = net.forward(xd, num_classes=ncls); net.loss_function('T', yd, **lg); net.backward(); torch.cuda.synchronize()
for c in reversed(net.convs):
    a = net._trace_dz[c.name].cpu().numpy().astype(np.float64); b = tr[c.name]
    d = np.abs(a-b); i = np.unravel_index(d.argmax(), d.shape)
    print('%-45s dz rell2 %.2e maxabs %.2e at %s (ref %.3e got %.3e) nbad(>1e-5*max)=%d  gradW %.2e' % (c.name, rell2(a,b), d.max(), i, b[i], a[i], (d > 1e-5*np.abs(b).max()).sum(),
          rell2(net.get_variable(c.name+'/weights', net.grads).cpu().numpy().astype(np.float64), g64[c.name+'/weights'])))
