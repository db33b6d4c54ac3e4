#!/bin/bash
# ON THE GPU BOX: rows per group / grid cap of the LDS gather, ShuffleNet-v2 steps
for r in 1 2; do
for v in "FTE_GATHER_U=4 FTE_GATHER_BLOCKS=2048" "FTE_GATHER_U=8 FTE_GATHER_BLOCKS=2048" "FTE_GATHER_U=16 FTE_GATHER_BLOCKS=2048" "FTE_GATHER_U=4 FTE_GATHER_BLOCKS=1024" "FTE_GATHER_U=8 FTE_GATHER_BLOCKS=1024" "FTE_GATHER_U=4 FTE_GATHER_BLOCKS=4096" "FTE_GATHER_U=2 FTE_GATHER_BLOCKS=4096" "FTE_GATHER_U=8 FTE_GATHER_BLOCKS=512"; do
  echo -n "$v f32@256 | "; env $v FTE_MFMA_DTYPE=f32 python3 scripts/bench_net.py ShuffleNet-v2-small 256 30 2>&1 | grep "ms/step" | sed 's/, losses.*//'
  echo -n "$v bf16s@512 | "; env $v FTE_MFMA_DTYPE=bf16s python3 scripts/bench_net.py ShuffleNet-v2-small 512 30 2>&1 | grep "ms/step" | sed 's/, losses.*//'
done; done
