"""Timeline of the last whole step of a graph net in a rocprofv3 --kernel-trace CSV (scripts/dev/trace_net.sh): start (us), duration,
stream, kernel, grid -- the first N and the last N launches, and the gaps of the main stream.
    python scripts/dev/timeline_net.py gpurun_out/<tag>/kernel_trace.csv [marker] [N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'momentum_kernel'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 70
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
# a step ends with the optimizer's launches: take the launches between the last two groups of them
ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] != i + 1]
seg = rows[ends[-2] + 1:ends[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
streams = sorted({r['Stream_Id'] for r in seg})
main = max(streams, key=lambda s: sum(1 for r in seg if r['Stream_Id'] == s))
out = []
last_end = {s: None for s in streams}
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:58]
    gap = (s - last_end[r['Stream_Id']]) / 1e3 if last_end[r['Stream_Id']] else 0.0
    last_end[r['Stream_Id']] = e
    out.append('%8.1f %7.1f  gap %6.1f  %s %-58s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, 'M' if r['Stream_Id'] == main else 's', n, r['Grid_Size_X']))
print('step: %.1f us, %d launches' % ((int(seg[-1]['End_Timestamp']) - t0) / 1e3, len(seg)))
print('\n'.join(out[:N]))
print('...')
print('\n'.join(out[-N:]))
