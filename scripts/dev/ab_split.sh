for rep in 1 2; do for sp in 256 320 192 384 128 448; do
FTE_FWD_SPLIT=$sp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('SPLIT $sp', d['ms_per_step'], d['value'], d['conv_algo']['forward_walk'])"
done; done
timeout 600 python -m pytest tests/test_gpu_spherenet.py tests/test_gpu_stress.py -q 2>&1 | tail -2
