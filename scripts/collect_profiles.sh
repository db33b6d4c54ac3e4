#!/bin/bash
# Run ON THE GPU BOX (gpurun): bench line + rocprofv3 kernel stats + the three PMC passes of the same command -> gpurun_out/prof_<tag>/.
# The profiler passes add --one-stream: every step runs as bench.py's launch-record steps do (one chain of kernels), so that a kernel's
# duration and its PMC bytes are its own, as the `roofline` block's launch durations are.
#   scripts/collect_profiles.sh r2 [extra bench.py flags]
TAG="${1:-r2}"; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --one-stream "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/p_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --one-stream "$@" > $OUT/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --one-stream "$@" > $OUT/p_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --one-stream "$@" > $OUT/p_mfma.log 2>&1
python3 scripts/pmc_summary.py $OUT/p_fetch $OUT/p_write $OUT/p_mfma $OUT/bench.json $OUT/pmc_summary.csv $OUT/traffic.json > $OUT/pmc_summary.log 2>&1
# keep what travels back small: the per-dispatch CSVs are large
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
du -sh $OUT; head -c 600 $OUT/bench.json; echo; head -12 $OUT/kernel_stats.csv; head -8 $OUT/pmc_summary.csv
