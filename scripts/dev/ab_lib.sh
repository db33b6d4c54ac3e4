# A/B of library variants on one box: bash scripts/dev/ab_lib.sh "<lib or empty> ..." [bench flags]
LIBS="$1"; shift
for rep in 1 2; do for l in $LIBS; do
  if [ "$l" = base ]; then unset FTE_LIB; else export FTE_LIB=variants/libfte_$l.so; fi
  FTE_BENCH_ANY_LIB=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" 2>&1 | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$l', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['all_mfma_kernels']['frac']); break
else: print('$l no json')"
done; done
