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
