#!/bin/bash
# ON THE GPU BOX: kernel trace of a few steps at 512 images -> gpurun_out/trfwd/kernel_trace.csv.gz (env passes through)
OUT=gpurun_out/trfwd; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --global-batch ${1:-512} --no-cpu-baseline --no-other-configs --steps 4 --warmup 3 > $OUT/trace.log 2>&1
find $OUT -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/trace; gzip -f $OUT/kernel_trace.csv; ls -la $OUT
