"""Per-kernel sums of the SQ counters of one `rocprofv3 --pmc ...` run (csv output), as shares of SQ_WAVE_CYCLES where that makes sense.
    python scripts/dev/sq_counters.py PMC_DIR OUT.csv [name filter]
MI355X_MICROARCH.md, "rocprofv3 PMC slots": SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall,
SQ_WAIT_INST_LDS a sub-bucket of it, SQ_ACTIVE_INST_ANY = issuing; the three add up to SQ_WAVE_CYCLES (quad-cycles)."""
import csv, glob, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ''
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = defaultdict(float)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        k = k.split('(')[0]
        if flt and flt not in k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVES':
            cnt[k] += 1
            dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
names = sorted({c for v in acc.values() for c in v})
with open(out, 'w') as f:
    f.write('kernel,dispatches,avg_us,' + ','.join(names) + ',wait_any_share,wait_inst_share,active_share,lds_issue_stall_share\n')
    for k in sorted(acc, key=lambda k: -dur[k]):
        v = acc[k]
        wc = v.get('SQ_WAVE_CYCLES', 0.0) or 1.0
        f.write('"%s",%d,%.1f,' % (k, cnt[k], dur[k] / max(cnt[k], 1) / 1e3) + ','.join('%.0f' % v.get(c, 0.0) for c in names) +
                ',%.3f,%.3f,%.3f,%.3f\n' % (v.get('SQ_WAIT_ANY', 0) / wc, v.get('SQ_WAIT_INST_ANY', 0) / wc, v.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                                            v.get('SQ_WAIT_INST_LDS', 0) / wc))
print(open(out).read()[:4000])
