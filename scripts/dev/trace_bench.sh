#!/bin/bash
# ON THE GPU BOX: bench.py at one shard size, then a kernel trace of a few steps -> gpurun_out/<tag>/step_summary.txt
#   bash scripts/dev/trace_bench.sh 64 f32 tag
B="$1"; DT="$2"; TAG="$3"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --global-batch $B --mfma-dtype $DT --no-cpu-baseline --no-other-configs --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c1-260 > $OUT/bench.json
cat $OUT/bench.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --global-batch $B --mfma-dtype $DT --no-cpu-baseline --no-other-configs --steps 6 --warmup 4 > $OUT/trace.log 2>&1
find $OUT -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
python3 scripts/trace_steps.py $OUT/kernel_trace.csv asoftmax_kernel 4 > $OUT/step_summary.txt 2>&1
rm -rf $OUT/trace
head -${4:-45} $OUT/step_summary.txt | cut -c1-120
