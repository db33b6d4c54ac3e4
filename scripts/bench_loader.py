"""Throughput of the host input pipeline (tf_face_toolbox_amd/data.py: list reader -> JPEG decode -> TF-1.x bilinear resize ->
crop / flip -> normalise -> NHWC float32 batch) in images/s at 112x112, on this host's cores -- to be read against the
rate the GPU step consumes (~10 k images/s per MI355X in fp32, ~30 k in the bf16 mode).

    python scripts/bench_loader.py [--images 2048] [--batch 512] [--batches 60] [--warmup 24] [--src 250] [--device cpu|cuda]

Writes N synthetic JPEGs of src x src pixels (CASIA-WebFace crops are 250 x 250) to a temporary directory, then times
`data.train_inputs(...)` batches (resize to 128 x 128, random crop 112 x 112, flip): the same call train.py makes."""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np          # noqa: E402
from PIL import Image       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=2048)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--batches', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=24, help='untimed batches first: more than the pipeline holds prefetched (2 per group of 16 workers), so that the timed batches are decoded at the sustained rate')
    ap.add_argument('--src', type=int, default=250)
    ap.add_argument('--device', default=None, help="default: cuda when a GPU is present (the training path: pinned staging ring + async copy), else cpu")
    ap.add_argument('--workers', type=int, default=None, help='decode worker processes (default: data.train_inputs picks; 0 = threads)')
    args = ap.parse_args()
    from tf_face_toolbox_amd import data
    if args.device is None:
        import torch
        args.device = 'cuda' if torch.cuda.is_available() else 'cpu'
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as d:
        lines = []
        base = rng.integers(0, 255, (args.src, args.src, 3), dtype=np.uint8)
        for i in range(args.images):
            p = os.path.join(d, '%06d.jpg' % i)
            Image.fromarray(np.roll(base, i * 7, axis=1)).save(p, quality=90)
            lines.append('%s %d' % (p, i % 100))
        lst = os.path.join(d, 'list.txt')
        open(lst, 'w').write('\n'.join(lines) + '\n')
        inp = data.train_inputs(lst, 128, 128, 112, 112, is_color=1, batch_size=args.batch, device=args.device, seed=0, num_workers=args.workers)
        for _ in range(max(1, args.warmup)):             # warm-up: worker start, page cache -- and the prefetched batches (a short run
            inp['images'](); inp['labels']()             # timed right after start-up is served out of the filled pipe and reads 1.5-4x high)
        t0 = time.time()
        for _ in range(args.batches):
            x = inp['images']()
            inp['labels']()
        if args.device != 'cpu':
            import torch
            torch.cuda.synchronize()
        el = time.time() - t0
        print('loader: %.0f images/s sustained (%d batches of %d after %d untimed, %dx%d JPEG -> 128x128 -> crop 112x112, workers %s, os.cpu_count=%d, usable CPUs (affinity / cgroup quota) %d, device %s), batch %s %s'
              % (args.batches * args.batch / el, args.batches, args.batch, args.warmup, args.src, args.src,
                 'auto' if args.workers is None else args.workers, os.cpu_count(), data.usable_cpus(), args.device, tuple(x.shape), x.dtype))


if __name__ == '__main__':
    main()
