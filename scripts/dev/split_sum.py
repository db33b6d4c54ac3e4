"""tower-split additivity at the full size, per variable: gradient(512) vs gradient(256) + gradient(256) (tests/test_gpu_fullsize.py)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tf_face_toolbox_amd import net_select
B, H, W, CH, NCLS = 512, 112, 112, 3, 10575
def step(net, x, y, scale):
    net.tower_scale = scale; net.global_step = 0
    out = net.forward(x, y, num_classes=NCLS, is_training=True)
    net.loss_function('T', y, **out); net.backward(); torch.cuda.synchronize()
    return net.grads[:net.arena_size].clone()
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, H, W, CH, generator=g) * 2 - 1).cuda(); y = torch.randint(0, NCLS, (B,), generator=g, dtype=torch.int32).cuda()
net = net_select('SphereNet-ASoftmax', 'NCHW', 5e-4); net.seed = 2; net.build(H, W, CH, NCLS, 'cuda')
gf = step(net, x, y, 1.0).cpu().numpy().astype(np.float64)
gs = (step(net, x[:256], y[:256], 0.5) + step(net, x[256:], y[256:], 0.5)).cpu().numpy().astype(np.float64)
worst = []
for name, v in net.variables.items():
    a, b = gs[v.offset:v.offset + v.size], gf[v.offset:v.offset + v.size]
    worst.append((float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())), name, float(np.abs(b).max())))
for e, n, m in sorted(worst, reverse=True)[:12]:
    print('%.3e %-50s max|g| %.3e' % (e, n, m))
