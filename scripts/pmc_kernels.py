"""Per-kernel HBM traffic of ANY rocprofv3 run (the BN / ShuffleNet nets: scripts/collect_net_profiles.sh), from the two PMC passes
FETCH_SIZE and WRITE_SIZE joined with the durations of the kernel trace that every pass carries.

    python scripts/pmc_kernels.py FETCH_DIR WRITE_DIR OUT_CSV [steps]

Corrections as MI355X_MICROARCH.md's HBM section prescribes: the counters are in KiB; gfx950 reports HALF the bytes of wide
(16 B / lane) coalesced reads, so HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE.  Per kernel symbol: dispatches, average duration,
share of the summed kernel time, HBM MB per launch, achieved GB/s = bytes / duration and its fraction of the 8 TB/s peak.
The MFMA kernels (igemm*) are compute-bound: their HBM fraction is printed for completeness, their roofline is the matrix peak
(bench.py).  The last line is the run-level figure: all HBM bytes / all kernel time."""
import csv, glob, sys
from collections import defaultdict

fetch_dir, write_dir, out_csv = sys.argv[1:4]


def read(d, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0])          # dispatches, counter sum, duration sum (ns)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                a = acc[r['Kernel_Name']]
                a[0] += 1; a[1] += float(r['Counter_Value']); a[2] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return acc


fe, wr = read(fetch_dir, 'FETCH_SIZE'), read(write_dir, 'WRITE_SIZE')
rows = []
for k, (n, fsum, dur) in fe.items():
    wn, wsum, wdur = wr.get(k, (0, 0.0, 0.0))
    f_kb = fsum / n
    w_kb = wsum / wn if wn else 0.0
    us = (dur / n + (wdur / wn if wn else dur / n)) / 2 / 1e3          # the two passes time the same kernels
    hbm = (2 * f_kb + w_kb) * 1024
    rows.append([k, n, us, hbm])
tot_us = sum(r[1] * r[2] for r in rows)
tot_b = sum(r[1] * r[3] for r in rows)
rows.sort(key=lambda r: -r[1] * r[2])
with open(out_csv, 'w') as f:
    f.write('kernel,dispatches,avg_us,share_of_kernel_time,hbm_MB_per_launch(2*FETCH+WRITE),achieved_GBps,frac_of_8TBps,bound\n')
    for k, n, us, hbm in rows:
        short = k.replace('(anonymous namespace)::', '').replace('void ', '')
        depth, cut = 0, len(short)
        for i, ch in enumerate(short):                   # drop the argument list: the first '(' outside the template brackets
            if ch == '<':
                depth += 1
            elif ch == '>':
                depth -= 1
            elif ch == '(' and depth == 0:
                cut = i
                break
        short = short[:cut]
        f.write('"%s",%d,%.1f,%.4f,%.2f,%.0f,%.3f,%s\n' % (short, n, us, n * us / tot_us, hbm / 1e6, hbm / us / 1e3, hbm / us / 1e3 / 8000,
                                                        'mfma' if 'igemm' in short or 'mfma' in short else 'hbm'))
    f.write('"ALL KERNELS",%d,%.1f,1.0,%.2f,%.0f,%.3f,-\n' % (sum(r[1] for r in rows), tot_us, tot_b / 1e6, tot_b / tot_us / 1e3, tot_b / tot_us / 1e3 / 8000))
print(open(out_csv).read()[:6000])
