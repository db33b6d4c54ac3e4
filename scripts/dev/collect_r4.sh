#!/bin/bash
# ON THE GPU BOX: every profile of the round in one call (SphereNet in the three precision modes, the three BN nets at their shards)
bash scripts/collect_all.sh r4 > gpurun_out/collect_all_r4.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh ResNeXt-50-center 128 r4_resnext50_bf16s_b128 > gpurun_out/collect_r4_resnext.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh SENet-50-triplet 128 r4_senet50_bf16s_b128 > gpurun_out/collect_r4_senet.log 2>&1
FTE_MFMA_DTYPE=f32 bash scripts/collect_net_profiles.sh ShuffleNet-v2-small 256 r4_shufflenet_b256 > gpurun_out/collect_r4_shufflenet.log 2>&1
FTE_MFMA_DTYPE=bf16s bash scripts/collect_net_profiles.sh ResNet-50 128 r4_resnet50_bf16s_b128 > gpurun_out/collect_r4_resnet50.log 2>&1
tail -30 gpurun_out/collect_all_r4.log
for t in resnext50_bf16s_b128 senet50_bf16s_b128 shufflenet_b256 resnet50_bf16s_b128; do head -2 gpurun_out/prof_r4_$t/step_summary.txt; tail -2 gpurun_out/prof_r4_$t/step_roofline.md; done
