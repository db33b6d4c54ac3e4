"""Generates tests/golden/*.npz from the float64 oracle (run in the build container):

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot be executed here, so these fixtures
freeze the ORACLE's outputs on small seeded inputs: they guard the oracle against accidental edits
(tests/test_golden.py, CPU) and give the HIP path committed known-answer cases (tests/test_gpu_golden.py).
Weights are NOT stored (24.5 M floats); they are regenerated from the seed by oracle.spherenet.init_params
(numpy PCG64 streams are stable across numpy versions for these distributions).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ops, spherenet as osn, graphnet as og      # noqa: E402


def sample_idx(shape, k=24, seed=0):
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    return np.sort(rng.choice(n, size=min(k, n), replace=False))


def spherenet_case(tag, seed, n, h, w, ch, ncls, data_format, head):
    p = osn.perturb_params(osn.init_params(seed, ch, ncls, h, w), seed + 1)
    rng = np.random.default_rng(seed + 2)
    x = rng.uniform(-1, 1, (n, h, w, ch))
    y = rng.integers(0, ncls, n)
    lam = ops.asoftmax_lambda(0)
    losses, g, ex = osn.loss_and_grads(p, x, y, 5e-4, data_format, head, lam)
    # min |z| / rms per layer: the test refuses to compare gradients tightly if a kink-band element exists
    emb, cache = osn.backbone_fwd(p, x, data_format)
    kink = min(float(np.abs(z).min() / np.sqrt((z * z).mean())) for _, _, z in cache['layers'])
    out = dict(meta=np.array([seed, n, h, w, ch, ncls]), data_format=data_format, head=head, lam=lam,
               images=x.astype(np.float32), labels=y.astype(np.int32),
               embedding=ex['embedding'], logits=ex['logits'], losses=np.array(losses), min_z_over_rms=kink)
    p2, s2, l2 = osn.train_step(p, osn.zero_slots(p), x, y, 0.1, head=head, lam=lam, data_format=data_format)
    for k in p:
        idx = sample_idx(p[k].shape, seed=len(k))
        out['gidx/' + k] = idx
        out['gval/' + k] = g[k].reshape(-1)[idx]
        out['gl2/' + k] = np.sqrt((g[k] ** 2).sum())
        out['w1/' + k] = p2[k].reshape(-1)[idx]          # weights after one momentum step, lr 0.1
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)
    print(tag, 'losses', losses, 'min|z|/rms %.2e' % kink)


def heads_case():
    rng = np.random.default_rng(77)
    n, d, c = 16, 512, 40
    x = rng.standard_normal((n, d)); w = rng.standard_normal((d, c)) * 0.05; y = rng.integers(0, c, n)
    out = dict(x=x, w=w, y=y.astype(np.int32))
    for lam in (5.0, 1000.0):
        loss, f, dx, dw = ops.asoftmax_fwd_bwd(x, w, y, lam)
        out['asm_loss_%g' % lam] = loss; out['asm_f_%g' % lam] = f; out['asm_dx_%g' % lam] = dx; out['asm_dw_%g' % lam] = dw
    loss, d_ = ops.softmax_ce(x @ w, y)
    out['ce_loss'] = loss; out['ce_dlogits'] = d_
    cen = rng.standard_normal((c, d)) * 0.1
    cl, cdf, cnew = ops.center_loss(x, y, cen, 0.99)
    out['centers'] = cen; out['center_loss'] = cl; out['center_df'] = cdf; out['center_new'] = cnew
    yk = np.repeat(np.arange(4), 4).astype(np.int32)
    for m in (None, 0.3):
        tl, tg = ops.batch_hard_triplet(x, yk, m)
        out['tri_labels'] = yk; out['tri_loss_%s' % m] = tl; out['tri_grad_%s' % m] = tg
    steps = np.arange(0, 1201, 50)
    out['lr_steps'] = steps
    out['lr_step'] = np.array([ops.lr_step(s, 0.1, 0.1, ['3', '5', '9'], 100) for s in steps])
    out['lr_exp'] = np.array([ops.lr_exp(s, 0.1, 2, 12, 100) for s in steps])
    out['lr_cos'] = np.array([ops.lr_cosine(s, 0.1, 12, 100) for s in steps])
    np.savez_compressed(os.path.join(HERE, 'heads.npz'), **out)
    print('heads ok')


def graphnet_cases():
    """Small graph-net steps (BN nets): ResNet-26, ResNeXt-26 + center loss, SE-ResNet-26 + triplet, ShuffleNet-v2 x2 with one
    block per stage and the focal head.  Parameters are regenerated from the seed (oracle.graphnet.init_params)."""
    out = {}
    cases = [('resnet26', lambda ncls: og.resnet_train_graph(26, 3, ncls), dict()),
             ('resnext26_center', lambda ncls: og.resnet_train_graph(26, 3, ncls, 'resnext'), dict(center=True)),
             ('senet26_triplet', lambda ncls: og.resnet_train_graph(26, 3, ncls, 'senet', classifier=False), dict(triplet=True)),
             ('shufflenet_small_focal', lambda ncls: og.shufflenet_train_graph('small', 3, ncls, 'NCHW', blocks_override=[1, 1, 1]), dict(focal=(1.0, 2.0)))]
    for ci, (tag, build, opt) in enumerate(cases):
        n, h, w, ncls = 8, 32, 32, 6
        seed = 500 + 10 * ci
        graph, spec = build(ncls)
        p, state = og.init_params(spec, seed)
        p = og.perturb(p, seed + 1)
        rng = np.random.default_rng(seed + 2)
        x = rng.uniform(-1, 1, (n, h, w, 3)).astype(np.float32).astype(np.float64)      # stored (and fed to the GPU) as float32
        y = np.repeat(np.arange(4), 2) if opt.get('triplet') else rng.integers(0, ncls, n)
        fdim = 2048
        masks = None if opt.get('triplet') else {'features_drop': (rng.random((n, fdim)) < 0.5).astype(np.float64)}
        kw = {}
        if opt.get('center'):
            kw['center'] = dict(centers=rng.standard_normal((ncls, fdim)) * 0.1, alpha=0.99, weight=0.05)
            out[tag + '/centers'] = kw['center']['centers']
        if opt.get('triplet'):
            kw['triplet_margin'] = None
        if opt.get('focal'):
            kw['focal'] = opt['focal']
        res = og.loss_and_grads(graph, p, x, y, 5e-4, masks=masks, state=state, **kw)
        losses, g, env, new_state = res[0], res[1], res[2], res[3]
        out[tag + '/meta'] = np.array([seed, n, h, w, ncls])
        out[tag + '/images'] = x.astype(np.float32); out[tag + '/labels'] = y.astype(np.int32)
        if masks:
            out[tag + '/mask'] = masks['features_drop'].astype(np.uint8)
        out[tag + '/losses'] = np.array(losses); out[tag + '/features'] = env['features']
        names = sorted(g)
        out[tag + '/gl2'] = np.array([np.sqrt((g[k] ** 2).sum()) for k in names])
        k0 = [k for k in sorted(new_state) if k.endswith('moving_variance')][0]
        out[tag + '/mv0'] = new_state[k0]
        print(tag, 'losses', losses, len(names), 'gradients')
    np.savez_compressed(os.path.join(HERE, 'graphnets.npz'), **out)


if __name__ == '__main__':
    spherenet_case('sphere_softmax_nchw_32', 101, 4, 32, 32, 3, 10, 'NCHW', 'softmax')
    spherenet_case('sphere_asoftmax_nhwc_gray_48x16', 202, 3, 48, 16, 1, 33, 'NHWC', 'asoftmax')
    spherenet_case('sphere_asoftmax_nchw_112_gray', 303, 2, 112, 112, 1, 200, 'NCHW', 'asoftmax')
    heads_case()
    graphnet_cases()
