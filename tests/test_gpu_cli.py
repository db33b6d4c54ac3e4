"""-m gpu: train.py -> checkpoint -> resume -> evaluate.py end to end on a tiny PNG list (the callers
either side of the hot path)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_train_resume_evaluate(tmp_path):
    from PIL import Image
    from scipy.io import loadmat
    rng = np.random.default_rng(0)
    lines = []
    for c in range(4):
        for i in range(6):
            path = str(tmp_path / ('id%d_%d.png' % (c, i)))
            Image.fromarray(rng.integers(0, 255, (40, 40, 3), dtype=np.uint8)).save(path)
            lines.append('%s %d' % (path, c))
    (tmp_path / 'train.txt').write_text('\n'.join(lines) + '\n')
    common = ['--net_name', 'SphereNet', '--model_name', 't', '--train_list_path', str(tmp_path / 'train.txt'),
              '--input_height', '36', '--input_width', '36', '--crop_height', '32', '--crop_width', '32',
              '--batch_size', '8', '--num_gpus', '1', '--init_lr', '0.01', '--lr_decay_epoch', '2', '--max_epoches', '50',
              '--display_interval', '1', '--save_interval', '1000']
    out = _run([os.path.join(ROOT, 'train.py')] + common + ['--max_steps', '3'], str(tmp_path))
    assert 'Network parameters initialized from scratch.' in out and 'Loss #0: cross_entropy' in out
    assert 'throughput =' in out and 'Model has been saved in Iteration 2' in out
    assert os.path.exists(str(tmp_path / 'models' / 'SphereNet_t' / 'SphereNet_t.ckpt-3'))
    out = _run([os.path.join(ROOT, 'train.py')] + common + ['--max_steps', '5'], str(tmp_path))
    assert 'Model restored from' in out and 'Epoch/Step 1/3' in out            # resumed at global_step 3
    out = _run([os.path.join(ROOT, 'evaluate.py'), '--net_name', 'SphereNet', '--model_name', 't', '--fea_name', 'f',
                '--data_list_path', str(tmp_path / 'train.txt'), '--input_height', '32', '--input_width', '32',
                '--batch_size', '16'], str(tmp_path))
    assert 'Totally extracted 24 features.' in out
    m = loadmat(str(tmp_path / 'features' / 'SphereNet_t' / 'f_5.mat'))
    assert m['wfea'].shape == (24, 512) and np.isfinite(m['wfea']).all() and np.abs(m['wfea']).max() > 0


@pytest.mark.parametrize('net_name,variant,layers,classifier', [('SENet-50-triplet', 'senet', 50, False), ('ResNeXt-26-center', 'resnext', 26, True)])
def test_train_save_evaluate_graph_nets(tmp_path, net_name, variant, layers, classifier):
    """evaluate.py on the BN nets -- including config 4's SENet-50-triplet, which has NO classifier (nets/resnet.py:67-68; round 2's
    extractor died on its checkpoint with KeyError) and config 3's center-loss net, whose checkpoint also carries `centers`:
    train two steps, save, extract, and compare `wfea` with the float64 graph oracle's inference-mode features computed from
    the checkpoint's own variables and moving statistics (reference: evaluate.py:53-100)."""
    from PIL import Image
    from scipy.io import loadmat
    from oracle import graphnet as og
    rng = np.random.default_rng(3)
    lines, imgs = [], []
    for c in range(4):
        for i in range(3):
            path = str(tmp_path / ('id%d_%d.png' % (c, i)))
            a = rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)
            Image.fromarray(a).save(path)
            lines.append('%s %d' % (path, c))
            imgs.append(a)
    (tmp_path / 'train.txt').write_text('\n'.join(lines) + '\n')
    common = ['--net_name', net_name, '--model_name', 'g', '--train_list_path', str(tmp_path / 'train.txt'),
              '--input_height', '64', '--input_width', '64', '--num_gpus', '1', '--init_lr', '0.01', '--lr_decay_epoch', '2',
              '--max_epoches', '50', '--display_interval', '1', '--save_interval', '1000', '--max_steps', '2']
    common += ['--num_classes', '4', '--num_per_class', '2'] if not classifier else ['--batch_size', '8']
    out = _run([os.path.join(ROOT, 'train.py')] + common, str(tmp_path))
    assert 'Model has been saved in Iteration 1' in out
    tag = net_name + '_g'
    out = _run([os.path.join(ROOT, 'evaluate.py'), '--net_name', net_name, '--model_name', 'g', '--fea_name', 'f',
                '--data_list_path', str(tmp_path / 'train.txt'), '--input_height', '64', '--input_width', '64', '--batch_size', '8'],
               str(tmp_path))
    assert 'Totally extracted 12 features.' in out
    wfea = loadmat(str(tmp_path / 'features' / tag / 'f_2.mat'))['wfea']
    assert wfea.shape == (12, 2048) and np.isfinite(wfea).all()
    ck = torch.load(str(tmp_path / 'models' / tag / (tag + '.ckpt-2')), map_location='cpu')
    v = {k: t.numpy().astype(np.float64) for k, t in ck['variables'].items()}
    assert ('classifier/fc_classifier/weights' in v) == classifier and ('centers' in v) == ('center' in net_name)
    graph, spec, feat, _ = og.resnet_graph(layers, 3, variant)
    p0, state0 = og.init_params(spec, 1)
    p = {k: v[k] for k in p0}
    state = {k: v[k] for k in state0}
    assert any(np.abs(state[k]).max() > 0 for k in state if k.endswith('moving_mean'))        # two training steps moved the statistics
    x = (np.stack(imgs).astype(np.float64) / 255.0 - 0.5) / 0.5
    env, _, _ = og.forward(graph, p, x, train=False, state=state)
    ref = env[feat]
    err = np.abs(wfea - ref).max()
    assert err <= 1e-4 * np.abs(ref).max(), (err, np.abs(ref).max())
