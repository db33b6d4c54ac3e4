#!/bin/bash
# Run ON THE GPU BOX: both precision modes through collect_profiles.sh, the PMC traffic files placed where bench.py reads them, then the
# two bench lines that carry them.  Afterwards copy gpurun_out/prof_r2k*/{kernel_stats,pmc_summary}.csv, traffic.json and the bench lines into profiles/.
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r2k > gpurun_out/collect_r2k.log 2>&1
cp gpurun_out/prof_r2k/traffic.json profiles/r2_traffic.json
bash scripts/collect_profiles.sh r2k_bf16 --mfma-dtype bf16 > gpurun_out/collect_r2k_bf16.log 2>&1
cp gpurun_out/prof_r2k_bf16/traffic.json profiles/r2_bf16_traffic.json
python3 bench.py > gpurun_out/bench_final_fp32.json 2> gpurun_out/bench_final_fp32.err
python3 bench.py --mfma-dtype bf16 > gpurun_out/bench_final_bf16.json 2> gpurun_out/bench_final_bf16.err
tail -c 1500 gpurun_out/bench_final_fp32.json
