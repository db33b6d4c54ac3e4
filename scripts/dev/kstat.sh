#!/bin/bash
# ON THE GPU BOX: per-kernel averages of one net's training step.  usage: kstat.sh <mode> <net> <batch> <pattern> [ENV=..]
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mode=$1; net=$2; b=$3; pat=$4; shift 4
for v in "$@"; do export $v; done
export FTE_MFMA_DTYPE=$mode
rm -rf gpurun_out/kstat; mkdir -p gpurun_out/kstat
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstat -- python3 scripts/bench_net.py $net $b 10 > gpurun_out/kstat/run.log 2>&1
f=$(find gpurun_out/kstat -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$pat" <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        print('%-100s %6s calls %8.1f us  %5.2f %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
P
