cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_tiles
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tiles/trace -- python3 scripts/dev/tiles_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_tiles/trace/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'wino_tiles' in r['Kernel_Name']:
        d[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Grid_Size_Y', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(k, 'n', len(v), 'median us %.1f min %.1f' % (v[len(v) // 2], v[0]))
PY
find gpurun_out/prof_tiles -name '*.csv' -delete
