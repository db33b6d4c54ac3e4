#!/bin/bash
# as ab_multi16.sh plus SphereNet at 64 images
R="$1"; shift
for r in $(seq $R); do
  for cfg in "bf16s ResNeXt-50-center 128" "bf16s SENet-50-triplet 128" "bf16s ResNet-50 128" "bf16s ShuffleNet-v2-small 512" "bf16s SphereNet-ASoftmax 512" "bf16s SphereNet-ASoftmax 64" "bf16 SphereNet-ASoftmax 512"; do
    for v in "$@"; do
      c=($cfg)
      echo -n "$v ${c[0]} | "; env $v FTE_MFMA_DTYPE=${c[0]} python3 scripts/bench_net.py ${c[1]} ${c[2]} 30 2>&1 | grep "ms/step" | sed 's/, losses.*//'
    done
  done
done
