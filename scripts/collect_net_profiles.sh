#!/bin/bash
# Run ON THE GPU BOX: step-level HBM roofline + rocprofv3 kernel stats + FETCH / WRITE PMC passes of one net's training step.
#   bash scripts/collect_net_profiles.sh ShuffleNet-v2-small 256 r3_shufflenet      -> gpurun_out/prof_<tag>/
NAME="$1"; B="$2"; TAG="$3"; STEPS="${4:-6}"   # FTE_MFMA_DTYPE in the environment selects the precision mode
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 scripts/net_roofline.py $NAME $B 30 > $OUT/step_roofline.md 2> $OUT/step_roofline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_net.py $NAME $B $STEPS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p_fetch -- python3 scripts/bench_net.py $NAME $B 2 > $OUT/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p_write -- python3 scripts/bench_net.py $NAME $B 2 > $OUT/p_write.log 2>&1
python3 scripts/pmc_kernels.py $OUT/p_fetch $OUT/p_write $OUT/pmc_summary.csv > $OUT/pmc_summary.log 2>&1
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name '*kernel_trace.csv' -path '*stats*' -exec cp {} $OUT/kernel_trace.csv \;
python3 scripts/trace_steps.py $OUT/kernel_trace.csv > $OUT/step_summary.txt 2>&1
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
tail -4 $OUT/step_roofline.md; head -30 $OUT/pmc_summary.csv; head -3 $OUT/step_summary.txt
