"""End-to-end throughput of train.py fed by the real input pipeline (list file of JPEGs -> decode workers -> pinned staging ->
GPU step), to be read against bench.py's synthetic-input figure: does the loader keep the GPU busy?

    python scripts/bench_train_e2e.py [--net SphereNet-ASoftmax] [--batch 512] [--steps 60] [--images 4096] [--src 250]

Writes N synthetic JPEGs (src x src, CASIA-WebFace crops are 250 x 250) and a list file to a temporary directory, then runs
train.py (resize to 128 x 128, random crop 112 x 112, flip -- the reference's SphereFace recipe) as a child process and prints
its own 'mean throughput' line."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='SphereNet-ASoftmax')
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--images', type=int, default=4096)
    ap.add_argument('--src', type=int, default=250)
    ap.add_argument('--mfma_dtype', default='f32')
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        base = rng.integers(0, 255, (a.src, a.src, 3), dtype=np.uint8)
        lines = []
        for i in range(a.images):
            p = os.path.join(d, '%06d.jpg' % i)
            Image.fromarray(np.roll(base, i * 7, axis=1)).save(p, quality=90)
            lines.append('%s %d' % (p, i % 1000))
        lst = os.path.join(d, 'list.txt')
        open(lst, 'w').write('\n'.join(lines) + '\n')
        cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--net_name', a.net, '--model_name', 'e2e', '--train_dir', os.path.join(d, 'train'),
               '--model_dir', os.path.join(d, 'models'), '--train_list_path', lst, '--input_height', '128', '--input_width', '128',
               '--crop_height', '112', '--crop_width', '112', '--batch_size', str(a.batch), '--num_gpus', '1', '--init_lr', '0.001',
               '--max_epoches', '1000', '--lr_decay_epoch', '400,800', '--max_steps', str(a.steps), '--display_interval', '20', '--save_interval', '100000',
               '--mfma_dtype', a.mfma_dtype]
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
        tail = [ln for ln in out.stdout.splitlines() if 'mean throughput' in ln or 'sustained' in ln or 'Error' in ln or 'error' in ln]
        print('\n'.join(tail[-4:]) if tail else out.stdout[-2000:])
        return out.returncode


if __name__ == '__main__':
    sys.exit(main())
