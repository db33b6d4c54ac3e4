"""Per-step summary of a rocprofv3 --kernel-trace CSV: wall, launches, per-kernel time, and -- for multi-stream runs --
how much kernel time overlapped (sum of durations minus the union of the busy intervals)."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'softmax_ce'
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
for alt in ('triplet_dist', 'im2col_first_kernel', 'softmax'):          # nets without the default marker (no classifier: the triplet head)
    if len(idx) > nsteps:
        break
    idx = [i for i, r in enumerate(rows) if alt in r['Kernel_Name']]
nsteps = min(nsteps, len(idx) - 1)
a, b = idx[-nsteps - 1], idx[-1]
seg = rows[a:b]
wall = (max(int(r['End_Timestamp']) for r in seg) - int(seg[0]['Start_Timestamp'])) / nsteps / 1e6
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
iv = []
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n)[:70]
    agg[n][0] += 1; agg[n][1] += e - s; busy += e - s
    iv.append((s, e))
iv.sort()
union, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print('wall %.3f ms/step, launches/step %.0f, sum of kernel durations %.3f ms, union %.3f ms, idle %.3f ms, streams %s' % (
    wall, len(seg) / nsteps, busy / nsteps / 1e6, union / nsteps / 1e6, wall - union / nsteps / 1e6,
    sorted(set(r['Stream_Id'] for r in seg))))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-72s %6.1f %8.3f ms %7.1f us' % (n, c / nsteps, d / nsteps / 1e6, d / c / 1e3))
