"""-m gpu: the HIP path against the committed golden fixtures (known answers that do not need the
oracle at run time, only its seeded weight generator)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import spherenet as osn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

if torch.cuda.is_available():
    from util_gpu import dev, host, check_maxabs, call, stream, ws
    from tf_face_toolbox_amd import net_select, Singular


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLD, 'sphere_*.npz'))))
def test_spherenet_golden(path):
    g = np.load(path)
    seed, n, h, w, ch, ncls = [int(v) for v in g['meta']]
    head, fmt = str(g['head']), str(g['data_format'])
    p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
    net = net_select('SphereNet-ASoftmax' if head == 'asoftmax' else 'SphereNet', fmt, 5e-4)
    net.build(h, w, ch, ncls, 'cuda')
    net.load_params(p)
    inputs = {'images': dev(g['images']), 'labels': dev(g['labels'], torch.int32), 'num_classes': ncls, 'num_examples': n}
    step, losses, names, others = Singular(net, 0.1, 'Momentum')(inputs)
    torch.cuda.synchronize()
    check_maxabs(host(net.emb), g['embedding'], what='embedding')
    logits = net.logits_buf[:, :ncls] if head == 'asoftmax' else net.s_raw[:, :ncls]
    check_maxabs(host(logits), g['logits'], what='logits')
    assert abs(float(losses[0]) - g['losses'][0]) <= 1e-5 * max(1, g['losses'][0])
    assert abs(float(losses[1]) - g['losses'][1]) <= 1e-5 * max(1, g['losses'][1])
    net.backward()
    torch.cuda.synchronize()
    # Gradients: the fixtures were produced WITHOUT kink resolution (the oracle alone cannot know which
    # side the fp32 path takes), so a case that has elements inside the kink band (|z| < 1e-5 rms) is
    # held to the looser bound a few flipped PReLU slopes allow; kink-free cases to the tight one.
    tight = float(g['min_z_over_rms']) > 1e-5
    tol = 2e-5 if tight else 5e-3
    for k in p:
        got = host(net.get_variable(k, net.grads))
        wd_term = 5e-4 * p[k] if k.endswith('/weights') else 0 * p[k]
        full = got + wd_term
        assert abs(np.sqrt((full ** 2).sum()) - g['gl2/' + k]) <= tol * g['gl2/' + k], k
        samp = full.reshape(-1)[g['gidx/' + k]]
        assert np.abs(samp - g['gval/' + k]).max() <= tol * max(np.abs(full).max(), 1e-30), k
    if tight:
        step()
        torch.cuda.synchronize()
        for k in p:
            got = host(net.get_variable(k)).reshape(-1)[g['gidx/' + k]]
            scale = max(np.abs(p[k]).max(), 0.1 * np.abs(g['gval/' + k]).max())
            assert np.abs(got - g['w1/' + k]).max() <= 2e-5 * scale, k


def test_head_goldens():
    g = np.load(os.path.join(GOLD, 'heads.npz'))
    x, w, y = g['x'], g['w'], g['y']
    n, d = x.shape
    c = w.shape[1]
    ld = 128
    wp = np.zeros((d, ld)); wp[:, :c] = w
    lrows = torch.empty(n, device='cuda'); dl = torch.empty(n, ld, device='cuda')
    call('fte_softmax_ce_fwd_bwd', dev(x @ wp), dev(y, torch.int32), lrows, dl, n, c, ld, 1.0 / n, stream())
    assert abs(host(lrows).mean() - g['ce_loss']) <= 1e-5 * g['ce_loss']
    check_maxabs(host(dl)[:, :c], g['ce_dlogits'], what='ce dlogits')
    cen = dev(g['centers']); df = torch.empty(n, d, device='cuda')
    wsb, nb = ws(n * d * 4)
    call('fte_center_loss_fwd_bwd_update', dev(x), dev(y, torch.int32), cen, lrows, df, n, d, c, 0.99, 1.0 / (n * d), wsb, nb, stream())
    check_maxabs(host(cen), g['center_new'], what='centers'); check_maxabs(host(df), g['center_df'], what='center df')
    for m in (None, 0.3):
        tl = torch.empty(n, device='cuda'); tg = torch.empty(n, d, device='cuda')
        wsb, nb = ws(3 * n * n * 4)
        call('fte_batch_hard_triplet_fwd_bwd', dev(x), dev(g['tri_labels'], torch.int32), 0.0 if m is None else m, int(m is None), 1.0,
             tl, tg, n, d, wsb, nb, stream())
        check_maxabs(host(tl), g['tri_loss_%s' % m], what='triplet loss'); check_maxabs(host(tg), g['tri_grad_%s' % m], what='triplet grad')


@pytest.mark.parametrize('tag', ['resnet26', 'resnext26_center', 'senet26_triplet', 'shufflenet_small_focal'])
def test_graphnet_goldens(tag):
    """The BN nets against the committed known-answer cases (tests/golden/graphnets.npz): losses, pooled features, the
    L2 norm of every gradient tensor (the per-element comparison with kink resolution lives in test_gpu_resnet /
    test_gpu_shufflenet; a golden file cannot know which side of a ReLU kink the fp32 run takes) and a moving variance."""
    from test_golden import graph_case_setup
    from tf_face_toolbox_amd.nets.resnet import ResNet, ResNeXt, SENet
    from tf_face_toolbox_amd.nets.shufflenet_v2 import ShuffleNet_v2_small
    g = np.load(os.path.join(GOLD, 'graphnets.npz'))
    graph, spec, p, state, x, y, masks, kw = graph_case_setup(g, tag)
    seed, n, h, w, ncls = [int(v) for v in g[tag + '/meta']]
    if tag == 'resnet26':
        net = ResNet(26)
    elif tag == 'resnext26_center':
        net = ResNeXt(26, head='softmax+center', center_weight=0.05)
    elif tag == 'senet26_triplet':
        net = SENet(26, head='triplet')
    else:
        net = ShuffleNet_v2_small(alpha=2.0); net.num_block = [1, 1, 1]; net.head = 'focal'
    net.build(h, w, 3, ncls, 'cuda')
    assert net.graph == graph
    net.load_params(p)
    if 'center' in kw:
        net._centers().copy_(torch.tensor(kw['center']['centers'], dtype=torch.float32))
    out = net.forward(dev(x), num_classes=ncls, is_training=True)
    if masks is not None:                                   # replay the fixture's dropout mask instead of the RNG's
        net.t['features_drop/mask'].copy_(dev(masks['features_drop']))
        net.t['features_drop'].copy_(net.t['features'] * net.t['features_drop/mask'] / 0.5)
        plan_fc = net.plan[-1]
        call('fte_gemm_nn', net.t[plan_fc[2]], net.view(plan_fc[3]), None, net.t['logits'], n, net.cpad, 2048, net.ws, net.ws_bytes, stream())
    losses, names, _ = net.loss_function('T', dev(y, torch.int32), **out)
    net.backward()
    torch.cuda.synchronize()
    check_maxabs(host(net.t['features'])[..., :2048], g[tag + '/features'], 5e-5, 'pooled features')
    got = [float(v) for v in losses]
    for a, b in zip(got, g[tag + '/losses']):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (names, got, g[tag + '/losses'])
    gl2 = [float(torch.linalg.vector_norm(net.get_variable(k, net.grads).double() + (5e-4 * torch.tensor(p[k], device='cuda') if k.endswith('weights') else 0)))
           for k in sorted(p) if k in net.variables]
    ref = g[tag + '/gl2']
    assert len(gl2) == len(ref)
    for k, a, b in zip(sorted(p), gl2, ref):
        # 1 %: at 8 x 32 x 32 a handful of ReLU / max-pool decisions inside fp32's noise band move a norm by a few 1e-3
        assert abs(a - b) <= 1e-2 * max(b, 1e-5 * ref.max()), (k, a, b)
