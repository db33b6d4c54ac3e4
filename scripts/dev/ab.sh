#!/bin/bash
# A/B on ONE box: each FTE_IGEMM16_DBG value three times, interleaved; prints the per-layer minimum of every configuration
#   bash scripts/dev/ab.sh "0 8 16 24"
for r in 1 2 3; do for d in $1; do FTE_IGEMM16_DBG=$d python scripts/bench_s16.py 512 10 2>/dev/null | grep "x" | sed "s/^/DBG$d /"; done; done > /tmp/ab.log
python3 - "$1" <<'PY'
import sys, re, collections
best = collections.defaultdict(lambda: [9, 9])
for l in open('/tmp/ab.log'):
    m = re.match(r'DBG(\d+)\s+(\d+)x\d+.*fwd ([\d.]+) ms.*dgrad ([\d.]+) ms', l)
    if m:
        k = (m.group(1), m.group(2)); f, d = float(m.group(3)), float(m.group(4))
        best[k][0] = min(best[k][0], f); best[k][1] = min(best[k][1], d)
for d in sys.argv[1].split():
    print('DBG', d, ' '.join('%sx: fwd %.4f dgrad %.4f |' % (hw, *best[(d, hw)]) for hw in ('56', '28', '14', '7')))
PY
