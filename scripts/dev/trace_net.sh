#!/bin/bash
# ON THE GPU BOX: kernel trace of a few steps of one net -> gpurun_out/<tag>/step_summary.txt (+ the raw trace)
#   bash scripts/dev/trace_net.sh ResNeXt-50-center 128 r4_fused   (FTE_MFMA_DTYPE etc. from the environment)
NAME="$1"; B="$2"; TAG="$3"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/bench_net.py $NAME $B 4 > $OUT/trace.log 2>&1
find $OUT -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
python3 scripts/trace_steps.py $OUT/kernel_trace.csv > $OUT/step_summary.txt 2>&1
rm -rf $OUT/trace
head -60 $OUT/step_summary.txt
