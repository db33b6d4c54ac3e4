#!/bin/bash
# ON THE GPU BOX: several settings of environment switches over the four profiled BN-net configurations, interleaved.
#   ab_multi_env.sh <rounds> "VAR=a" "VAR=b" "VAR=c VAR2=d" ...
R="$1"; shift
for r in $(seq $R); do
  for cfg in "bf16s ResNeXt-50-center 128" "bf16s SENet-50-triplet 128" "bf16s ResNet-50 128" "f32 ShuffleNet-v2-small 256"; do
    set -- $cfg "$@"
    m=$1; net=$2; b=$3; shift 3
    for v in "$@"; do
      echo -n "$v $m | "; env $v FTE_MFMA_DTYPE=$m python3 scripts/bench_net.py $net $b 30 2>&1 | grep "ms/step" | sed 's/, losses.*//'
    done
  done
done
