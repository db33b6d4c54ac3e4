#!/bin/bash
# ON THE GPU BOX: every profile of round 5 in one call -- SphereNet in the three precision modes (bench line, kernel stats, PMC passes),
# the four BN nets at their shards, the small-shard bench lines, and the SQ counters of the streaming pointwise kernel.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash scripts/collect_all.sh r5 > gpurun_out/collect_all_r5.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh ResNeXt-50-center 128 r5_resnext50_bf16s_b128 > gpurun_out/collect_r5_resnext.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh SENet-50-triplet 128 r5_senet50_bf16s_b128 > gpurun_out/collect_r5_senet.log 2>&1
FTE_MFMA_DTYPE=f32 bash scripts/collect_net_profiles.sh ShuffleNet-v2-small 256 r5_shufflenet_b256 > gpurun_out/collect_r5_shufflenet.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh ResNet-50 128 r5_resnet50_bf16s_b128 > gpurun_out/collect_r5_resnet50.log 2>&1
for b in 64 128 256; do
  python3 bench.py --global-batch $b --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/prof_r5/bench_b$b.json 2> gpurun_out/prof_r5/bench_b$b.err
  FTE_SK=0 FTE_CLS_ROT=31 python3 bench.py --global-batch $b --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/prof_r5/bench_b${b}_nosk.json 2>/dev/null
done
# the 64-image shard's kernel stats (which symbols the small shard runs on)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r5/stats_b64 -- python3 bench.py --global-batch 64 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r5/stats_b64.log 2>&1
find gpurun_out/prof_r5/stats_b64 -name '*kernel_stats.csv' -exec cp {} gpurun_out/prof_r5/kernel_stats_b64.csv \;
rm -rf gpurun_out/prof_r5/stats_b64
# pw16: where its waves spend their cycles
mkdir -p gpurun_out/prof_r5_pw16
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_r5_pw16/sq -- python3 scripts/dev/pw16_bench.py > gpurun_out/prof_r5_pw16/bench.log 2>&1
python3 scripts/dev/sq_counters.py gpurun_out/prof_r5_pw16/sq gpurun_out/prof_r5_pw16/sq_counters.csv pw16 > gpurun_out/prof_r5_pw16/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r5_pw16/fetch -- python3 scripts/dev/pw16_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_r5_pw16/write -- python3 scripts/dev/pw16_bench.py > /dev/null 2>&1
python3 scripts/pmc_kernels.py gpurun_out/prof_r5_pw16/fetch gpurun_out/prof_r5_pw16/write gpurun_out/prof_r5_pw16/hbm.csv > gpurun_out/prof_r5_pw16/hbm.log 2>&1
find gpurun_out/prof_r5_pw16 -name '*counter_collection.csv' -delete; find gpurun_out/prof_r5_pw16 -name '*agent_info.csv' -delete
tail -30 gpurun_out/collect_all_r5.log
for t in resnext50_bf16s_b128 senet50_bf16s_b128 shufflenet_b256 resnet50_bf16s_b128; do head -2 gpurun_out/prof_r5_$t/step_summary.txt; tail -2 gpurun_out/prof_r5_$t/step_roofline.md; done
for b in 64 128 256; do cut -c1-220 gpurun_out/prof_r5/bench_b$b.json; cut -c1-220 gpurun_out/prof_r5/bench_b${b}_nosk.json; done
cat gpurun_out/prof_r5_pw16/sq_counters.csv | cut -c1-300
