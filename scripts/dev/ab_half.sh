set -x
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_tiles.py -q -x 2>&1 | tail -5
for gb in 64 128 512; do for h in 1 0; do
FTE_WINO_HALF_TILES=$h python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --global-batch $gb 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('GB $gb HALF $h', d['ms_per_step'], d['value'])"
done; done
