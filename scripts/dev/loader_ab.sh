#!/bin/bash
# A/B of the loader's GPU transform under train.py (scripts/bench_train_e2e.py): FTE_LOADER_GPU=0 (all host) against 1 (default),
# and the worker count.
for net in "ShuffleNet-v2-small f32 256" "SphereNet-ASoftmax bf16s 512"; do
  set -- $net
  for g in "0 -1" "1 -1" "1 32" "1 64" "1 96"; do
    set -- $net $g
    echo "== $1 $2 batch $3 FTE_LOADER_GPU=$4 workers=$5"
    FTE_LOADER_GPU=$4 FTE_LOADER_WORKERS=$5 python scripts/bench_train_e2e.py --net $1 --mfma_dtype $2 --batch $3 --steps 400 --images 8192
  done
done
