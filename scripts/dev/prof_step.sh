cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_step/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/prof_step/bench.log 2>&1
find gpurun_out/prof_step -name '*kernel_stats.csv' | head -1 | xargs cat | cut -c1-220 | head -45
find gpurun_out/prof_step -name '*kernel_trace.csv' -delete
tail -1 gpurun_out/prof_step/bench.log | cut -c1-300
