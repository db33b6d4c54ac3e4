"""Per-stream busy time of the last whole steps in a kernel trace (is the side stream the critical one?), and the top kernels per stream.
    python scripts/dev/streams.py gpurun_out/<tag>/kernel_trace.csv [marker]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'softmax_ce'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
for alt in ('triplet_dist', 'im2col_first_kernel', 'softmax'):
    if len(idx) > 3:
        break
    idx = [i for i, r in enumerate(rows) if alt in r['Kernel_Name']]
a, b = idx[-3], idx[-1]
seg = rows[a:b]
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 2e3
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in seg:
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
    e = per[r['Stream_Id']][n]
    e[0] += 1; e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 2e3
print('wall %.0f us per step' % wall)
for s, d in per.items():
    print('stream %s: busy %.0f us per step' % (s, sum(v[1] for v in d.values())))
    for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:14]:
        print('   %-62s %5.1f %8.1f us' % (n, c / 2, t))
