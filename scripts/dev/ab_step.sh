#!/bin/bash
# A/B of whole bf16s training steps on ONE box: each "VAR=val" setting three times, interleaved; minimum ms per step
for r in 1 2 3; do for e in "$@"; do env $e python bench.py --mfma-dtype bf16s --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('CFG[$e]', d['ms_per_step'])"; done; done > /tmp/abs.log
python3 - "$@" <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: 99.0)
for l in open('/tmp/abs.log'):
    m = re.match(r'CFG\[(.*?)\] ([\d.]+)', l)
    if m: best[m.group(1)] = min(best[m.group(1)], float(m.group(2)))
for e in sys.argv[1:]: print(e, best[e], 'ms/step')
PY
