# A/B of one env switch on one box: bash scripts/dev/ab_env2.sh VAR "v1 v2" [bench flags]
VAR="$1"; VALS="$2"; shift; shift
for rep in 1 2 3; do for v in $VALS; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$VAR=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done; done
