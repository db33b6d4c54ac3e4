import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import graphnet as og
from tf_face_toolbox_amd.nets.resnet import ResNet
def rell2(a, b): return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30))
nl, n, h, w, ncls = 50, 6, 64, 48, 300
graph, spec = og.resnet_train_graph(nl, 3, ncls)
p, state = og.init_params(spec, 71); p = og.perturb(p, 72)
rng = np.random.default_rng(73)
x = rng.uniform(-1, 1, (n, h, w, 3)); y = rng.integers(0, ncls, n)
net = ResNet(nl); net.build(h, w, 3, ncls, 'cuda'); net.load_params(p)
xd = torch.tensor(x, dtype=torch.float32, device='cuda')
net.forward(xd, num_classes=ncls, is_training=True); torch.cuda.synchronize()
mask = net.t['features_drop/mask'].cpu().numpy().astype(np.float64)
env, cache, _ = og.forward(graph, p, x, train=True, masks={'features_drop': mask}, state=state)
p32 = {k: v.astype(np.float32) for k, v in p.items()}; s32 = {k: v.astype(np.float32) for k, v in state.items()}
env32, _, _ = og.forward(graph, p32, x.astype(np.float32), train=True, masks={'features_drop': mask.astype(np.float32)}, state=s32)
for op in net.plan:
    out = op[1]
    if out in env and out in net.t:
        a = net.t[out].cpu().numpy().astype(np.float64)
        if op[0] in ('conv',) or out.endswith('b0') or out.endswith('b2') or out in ('conv1', 'pool1', 'features', 'logits') or op[0] == 'bn' and '/c3' in out:
            print('%-14s %-5s hip %.2e  f32-oracle %.2e  shape %s' % (out, op[0], rell2(a, env[out]) if a.shape == env[out].shape else rell2(a[:, :ncls], env[out]), rell2(env32[out].astype(np.float64), env[out]), a.shape))
