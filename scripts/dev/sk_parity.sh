#!/bin/bash
# ON THE GPU BOX: the stream-K symbols forced onto a fixed case list (ragged shapes included), every result against the float64 oracle
C=$(cat tests/golden/sk_cases.json)
for t in 2 3 0; do
  echo "== FTE_SK=2 FTE_SK_TILE=$t"
  FTE_SK=2 FTE_SK_TILE=$t FTE_SK_DEBUG=1 timeout 900 python tests/tile_worker.py "$C" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        for c in r['cases']: print(c['ok'], c['case'], c['symbols'], {k: '%.2e' % v for k, v in c['errors'].items()})
    elif l.startswith('[sk]'): print(l.rstrip())
"
done
