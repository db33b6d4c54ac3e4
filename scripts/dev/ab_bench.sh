#!/bin/bash
# A/B of whole SphereNet training steps on ONE box: bash scripts/dev/ab_bench.sh "<bench.py args>" VAR=a VAR=b ...  (3 interleaved rounds, minimum ms/step)
ARGS="$1"; shift
for r in 1 2 3; do for e in "$@"; do env $e python bench.py $ARGS --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('CFG[$e]', d['ms_per_step'])"; done; done > /tmp/abb.log
python3 - "$ARGS" "$@" <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: 999.0)
for l in open('/tmp/abb.log'):
    m = re.match(r'CFG\[(.*?)\] ([\d.]+)', l)
    if m: best[m.group(1)] = min(best[m.group(1)], float(m.group(2)))
for e in sys.argv[2:]: print(sys.argv[1], e, best[e], 'ms/step')
PY
