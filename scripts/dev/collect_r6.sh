#!/bin/bash
# ON THE GPU BOX: the profiles of round 6 in one call -- SphereNet fp32 (bench line, rocprofv3 kernel stats, the three PMC passes), the
# bf16 / bf16s bench lines, the small-shard lines, the Winograd kernels' SQ counters / HBM traffic per resBlock shape and the in-kernel clock table.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash scripts/collect_all.sh r6 f32 > gpurun_out/collect_all_r6.log 2>&1
for m in bf16 bf16s; do python3 bench.py --mfma-dtype $m --no-cpu-baseline > gpurun_out/prof_r6/bench_$m.json 2> gpurun_out/prof_r6/bench_$m.err; done
for b in 64 128 256; do
  python3 bench.py --global-batch $b --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/prof_r6/bench_b$b.json 2> gpurun_out/prof_r6/bench_b$b.err
  FTE_CONV_ALGO=direct python3 bench.py --global-batch $b --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/prof_r6/bench_b${b}_direct.json 2>/dev/null
done
FTE_CONV_ALGO=direct python3 bench.py --no-cpu-baseline --no-other-configs > gpurun_out/prof_r6/bench_direct.json 2>/dev/null
for s in 1 2 3; do bash scripts/dev/wino_pmc.sh $s 512 > gpurun_out/wino_pmc_$s.log 2>&1; done
bash scripts/dev/build_wino_stamp.sh > /dev/null 2>&1
FTE_LIB=variants/libfte_wstamp.so python3 scripts/dev/wino_clock.py 512 > gpurun_out/prof_r6/wino_clock.md 2>&1
python3 scripts/dev/wino_bench.py 512 20 > gpurun_out/prof_r6/wino_bench.txt 2>&1
tail -c 1500 gpurun_out/prof_r6/bench_final.json; echo; head -14 gpurun_out/prof_r6/pmc_summary.csv | cut -c1-260
for b in 64 128 256; do cut -c1-200 gpurun_out/prof_r6/bench_b$b.json; cut -c1-200 gpurun_out/prof_r6/bench_b${b}_direct.json; done
cut -c1-200 gpurun_out/prof_r6/bench_direct.json; cut -c1-200 gpurun_out/prof_r6/bench_bf16.json; cut -c1-200 gpurun_out/prof_r6/bench_bf16s.json
cat gpurun_out/prof_r6/wino_clock.md | grep -v amdgpu; grep -v amdgpu gpurun_out/prof_r6/wino_bench.txt
