#!/bin/bash
# ON THE GPU BOX: like ab_nets.sh for the four profiled configurations only.  usage: ab_nets_env.sh "VAR=a" "VAR=b" [rounds]
A="$1"; Bv="$2"; R="${3:-2}"
for r in $(seq $R); do
  for cfg in "bf16s ResNeXt-50-center 128" "bf16s SENet-50-triplet 128" "bf16s ResNet-50 128" "f32 ShuffleNet-v2-small 256"; do
    set -- $cfg
    for v in "$A" "$Bv"; do
      echo -n "$v $1 | "; env $v FTE_MFMA_DTYPE=$1 python3 scripts/bench_net.py $2 $3 30 2>&1 | grep "ms/step" | sed 's/, losses.*//'
    done
  done
done
