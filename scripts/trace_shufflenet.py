"""Per-variable gradient error of the ShuffleNet engine vs the float64 oracle (debug aid; run on the GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from oracle import graphnet as og
from util_gpu import dev, host
from tf_face_toolbox_amd.nets.shufflenet_v2 import ShuffleNet_v2_small

n, h, w, ncls = 8, 64, 64, 10
blocks = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [4, 8, 4]
graph, spec = og.shufflenet_train_graph('small', 3, ncls, 'NCHW', blocks_override=blocks)
p, state = og.init_params(spec, 31); p = og.perturb(p, 32)
rng = np.random.default_rng(33)
x = rng.uniform(-1, 1, (n, h, w, 3)); y = rng.integers(0, ncls, n)
net = ShuffleNet_v2_small(alpha=2.0); net.num_block = blocks
net.build(h, w, 3, ncls, 'cuda'); net.load_params(p); net.dropout_seed = 5
out = net.forward(dev(x), num_classes=ncls, is_training=True)
net.loss_function('T', dev(y, torch.int32), **out); net.backward(); torch.cuda.synchronize()
mask = host(net.t['features_drop/mask'])
kink = {}
for op in net.graph:
    if op[0] == 'relu':
        kink[op[1]] = host(net.t[op[1]])[..., :net.real_c[op[1]]]
    elif op[0] == 'maxpool':
        kink[op[1] + '/idx'] = net.t[op[1] + '/idx'].cpu().numpy()[..., :net.real_c[op[1]]]
        kink[op[1]] = host(net.t[op[1]])[..., :net.real_c[op[1]]]
l, g, env, ns = og.loss_and_grads(graph, p, x, y, 5e-4, masks={'features_drop': mask}, state=state, kink=kink, bands=og.noise_bands(graph, p, x, {'features_drop': mask}, state))
for k in p:
    got = host(net.get_variable(k, net.grads)) + (5e-4 * p[k] if k.endswith('weights') else 0)
    ref = g[k]
    e = np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref * ref).sum()), 1e-30)
    if e > 1e-4:
        d = np.abs(got - ref)
        ax = tuple(i for i in range(d.ndim) if i != d.ndim - 2) if d.ndim >= 2 else ()
        rows = d.max(axis=ax) if d.ndim >= 2 else d
        bad = np.nonzero(rows > 1e-3 * np.abs(ref).max())[0]
        print('%-80s %.2e shape %s bad-cin-rows %s..%s (%d)' % (k, e, ref.shape, bad[:1], bad[-1:], len(bad)))
nm = 'ShuffleNet_v2_small_x2/conv2/resBlock_0/'
for k in (nm + 'separable_conv2_3x3/depthwise_weights', nm + 'conv_shortcut_1x1/weights', nm + 'separable_conv2_3x3/pointwise_weights'):
    raw = host(net.get_variable(k, net.grads)); ref = g[k] - 5e-4 * p[k]
    print(k.split('/')[-2:], 'raw', raw.reshape(-1, raw.shape[-1] if raw.shape[-1] > 1 else raw.shape[-2])[:2, :6], 'ref', ref.reshape(-1, ref.shape[-1] if ref.shape[-1] > 1 else ref.shape[-2])[:2, :6])
v = net.variables
for k in list(v)[:0]:
    pass
names = [k for k in v if k.startswith(nm)]
print([(k[len(nm):], v[k].offset, v[k].size) for k in names if v[k].kind in ('conv_w', 'dw_w')])
print('done')
