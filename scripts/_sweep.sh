cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "FTE_LIB=variants/libfte_prev.so" "FTE_X=1"; do
  echo "== $cfg"; env $cfg python scripts/bench_net.py ShuffleNet-v2-small 512 30 2>&1 | tail -1
done
python -m pytest tests/test_gpu_layers.py tests/test_gpu_shufflenet.py -m gpu -x -q 2>&1 | tail -2
FTE_SIDE_STREAM=0 rocprofv3 --kernel-trace -d gpurun_out/prof_b -o b --output-format csv -- python3 scripts/bench_net.py ShuffleNet-v2-small 512 10 > gpurun_out/prof_b.log 2>&1
