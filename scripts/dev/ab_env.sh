#!/bin/bash
# A/B on ONE box of environment settings: each "VAR=val" three times, interleaved; per-layer minimum of bench_s16.py
#   bash scripts/dev/ab_env.sh "FTE_IGEMM16_PERSIST=1" "FTE_IGEMM16_PERSIST=20"
for r in 1 2 3; do for e in "$@"; do env $e python scripts/bench_s16.py 512 10 2>/dev/null | grep "x" | sed "s/^/CFG[$e] /"; done; done > /tmp/abe.log
python3 - "$@" <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: [9, 9])
for l in open('/tmp/abe.log'):
    m = re.match(r'CFG\[(.*?)\]\s+(\d+)x\d+.*fwd ([\d.]+) ms.*dgrad ([\d.]+) ms', l)
    if m:
        k = (m.group(1), m.group(2)); f, d = float(m.group(3)), float(m.group(4))
        best[k][0] = min(best[k][0], f); best[k][1] = min(best[k][1], d)
for e in sys.argv[1:]:
    print(e, ' '.join('%sx: fwd %.4f dgrad %.4f |' % (hw, *best[(e, hw)]) for hw in ('56', '28', '14', '7')))
PY
