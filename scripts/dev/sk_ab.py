"""A/B of environment settings over scripts/bench_kernels.py on ONE box: every setting ROUNDS times, interleaved, per-layer minimum.
    python scripts/dev/sk_ab.py B "FTE_SK=0" "FTE_SK=2 FTE_SK_TILE=2" ...
Prints one table row per setting: fwd / dgrad microseconds of the seven SphereNet layer shapes at batch B and the per-step conv totals."""
import collections
import os
import re
import subprocess
import sys

B = sys.argv[1]
settings = sys.argv[2:]
ROUNDS = int(os.environ.get('SK_AB_ROUNDS', '3'))
here = os.path.dirname(os.path.abspath(__file__))
best = collections.defaultdict(lambda: [1e9, 1e9, 1e9])
for r in range(ROUNDS):
    for s in settings:
        env = dict(os.environ)
        for kv in s.split():
            k, v = kv.split('=', 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(here, '..', 'bench_kernels.py'), B, '10'], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print('FAILED', s, out.stderr[-2000:])
            continue
        for l in out.stdout.splitlines():
            m = re.match(r'\s*(\d+)x\d+\s+(\d+)->(\d+)\s+s(\d).*fwd ([\d.]+) ms.*dgrad ([\d.]+) ms.*wgrad ([\d.]+) ms', l)
            if m:
                k = (s, '%sx%s>%s/%s' % m.group(1, 2, 3, 4))
                for i in range(3):
                    best[k][i] = min(best[k][i], float(m.group(5 + i)))
layers = []
for (s, l) in best:
    if l not in layers:
        layers.append(l)
counts = {'56x64>64/1': 2, '56x64>128/2': 1, '28x128>128/1': 4, '28x128>256/2': 1, '14x256>256/1': 8, '14x256>512/2': 1, '7x512>512/1': 2}
print('%-44s' % 'setting (fwd/dgrad/wgrad us)' + ' '.join('%19s' % l for l in layers) + '   fwd+dgrad | wgrad ms/step')
for s in settings:
    tot = sum((best[(s, l)][0] + best[(s, l)][1]) * counts.get(l, 1) for l in layers)
    wg = sum(best[(s, l)][2] * counts.get(l, 1) for l in layers)
    print('%-44s' % s + ' '.join('%6.0f/%6.0f/%6.0f' % tuple(v * 1e3 for v in best[(s, l)]) for l in layers) + '   %.3f | %.3f' % (tot, wg))
