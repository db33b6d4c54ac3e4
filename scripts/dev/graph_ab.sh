#!/bin/bash
# ON THE GPU BOX (no profiler): eager step against HIP-graph replay, two-stream walk and one-stream walk
NET="${1:-ResNeXt-50-center}"; B="${2:-128}"; DT="${3:-bf16s}"
export FTE_MFMA_DTYPE=$DT
for rep in 1 2; do
for side in 1 0; do
  echo "== FTE_SIDE_STREAM=$side"
  FTE_SIDE_STREAM=$side python3 scripts/dev/graph_capture.py $NET $B 2>&1 | grep -v Warning | tail -3
done
echo "== FTE_SIDE_STREAM=1 FTE_PACK_AFTER_STEM=0"
FTE_PACK_AFTER_STEM=0 python3 scripts/dev/graph_capture.py $NET $B 2>&1 | tail -2
done
