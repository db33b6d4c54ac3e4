#!/bin/bash
# ON THE GPU BOX: kernel trace of eager steps and HIP-graph replays of the same step -> gpurun_out/graph_<net>/edges.txt
NET="${1:-ResNeXt-50-center}"; B="${2:-128}"; DT="${3:-bf16s}"
OUT=gpurun_out/graph_$NET
mkdir -p $OUT
export TMPDIR=/tmp
export FTE_MFMA_DTYPE=$DT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/dev/graph_edges.py run $NET $B > $OUT/run.log 2>&1
tail -3 $OUT/run.log
find $OUT -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
python3 scripts/dev/graph_edges.py cmp $OUT/kernel_trace.csv > $OUT/edges.txt 2>&1
rm -rf $OUT/trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_trace.csv')))
print(len(rows), 'rows; columns', list(rows[0].keys()))
PY
rm -f $OUT/kernel_trace.csv.gz; gzip -f $OUT/kernel_trace.csv
cat $OUT/edges.txt | cut -c1-330
