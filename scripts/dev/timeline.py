"""Timeline of the last whole step in a rocprofv3 --kernel-trace CSV: start (us), duration (us), kernel, grid.
    python scripts/dev/timeline.py gpurun_out/<tag>/kernel_trace.csv [marker]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else '::asoftmax_kernel('
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:64]
    print('%8.1f %7.1f %s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, n, r['Grid_Size_X']))
