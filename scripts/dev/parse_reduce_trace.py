import csv, sys, glob, collections
f = glob.glob('gpurun_out/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last step: take the last 1/5 of the rows roughly; print all reduce calls with their predecessor kernel
n = len(rows)
sel = rows[int(n * 0.75):]
agg = collections.defaultdict(lambda: [0, 0.0])
prev = None
for r in sel:
    name = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'reduce_rows' in name or 'reduce_slabs' in name:
        key = (name[:40], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], prev[:60] if prev else '')
        agg[key][0] += 1; agg[key][1] += d
    prev = name.replace('void (anonymous namespace)::', '')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%6.1f us x %d  grid %s,%s,%s  %s  after %s' % (v[1] / v[0], v[0], k[1], k[2], k[3], k[0], k[4]))
