timeout 1500 python -m pytest tests/test_gpu_wino.py tests/test_gpu_spherenet.py -q -x 2>&1 | tail -3
for gb in 512 128 64; do for cfg in 1 0 1 0; do
FTE_WINO_FILTER_PACKS=$cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --global-batch $gb 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('GB $gb FILTER_PACKS $cfg', d['ms_per_step'], d['value'])"
done; done
