for rep in 1 2; do for gb in 512 256 128; do for h in 1 2; do
FTE_WINO_HALF_TILES=$h python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --global-batch $gb 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('GB $gb HALF_TILES $h', d['ms_per_step'], d['value'])"
done; done; done
