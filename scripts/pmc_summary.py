"""Summarises the rocprofv3 --pmc passes of bench.py into profiles/ (per kernel symbol, averages over dispatches).

    python scripts/pmc_summary.py FETCH_DIR WRITE_DIR MFMA_DIR BENCH_JSON OUT_CSV OUT_JSON

Each pass is `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1
--no-cpu-baseline` (counters in separate passes, as MI355X_MICROARCH.md's HBM section prescribes).  Corrections from the
same section: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports HALF the bytes of wide (16 B / lane) coalesced reads, so
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE.  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8);
clock = GRBM_GUI_ACTIVE / 8 / dispatch duration (reads high on sub-ms dispatches: the in-kernel stamps of
scripts/clock_probe.py are the clock figures DESIGN.md quotes).
ALGORITHMIC bytes per launch come from BENCH_JSON (a bench.py line of the same build: roofline.per_symbol, i.e. the launch
records' own operand / result tensor sizes) -- nothing is hard-coded here.  OUT_JSON is what bench.py reads back as
`roofline.traffic`; it carries the kernel-source hash it was measured on and bench.py nulls it when the sources change."""
import csv, glob, json, os, re, sys
from collections import defaultdict

fetch_dir, write_dir, mfma_dir, bench_json, out_csv, out_json = sys.argv[1:7]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                           # noqa: E402  (kernel_src_sha, symbol naming)


def read(d):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def avg_by_kernel(rows, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r['Counter_Name'] == counter:
            a = acc[r['Kernel_Name']]
            a[0] += 1; a[1] += float(r['Counter_Value'])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


def short(kernel_name):
    """'void (anonymous namespace)::igemm_kernel<64, 64, 2, 2, 0, 0, 0, 0>(IgemmParams)' -> bench.py's 'igemm_kernel<64,64,2,2,0,0,0,0>'."""
    m = re.search(r'((?:igemm(?:16(?:rw|w|p|r)?)?|wgrad16p?|wino_mm)_kernel)<([^>]*)>', kernel_name)
    if m:
        return '%s<%s>' % (m.group(1), m.group(2).replace(' ', ''))
    return 'wino_wgrad_kernel' if 'wino_wgrad_kernel(' in kernel_name else None


bline = json.loads([l for l in open(bench_json) if l.startswith('{')][-1])
per_symbol = bline['roofline']['per_symbol']
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
    alg = (per_symbol.get(short(k) or '') or {}).get('algorithmic_bytes_per_launch')
    hbm = (f_kb * 2 + w_kb) * 1024
    table.append((hbm, k, cnt, f_kb, w_kb, busy.get(k, 0.0) / (1024 * cyc) if cyc else 0.0, cyc / dur[k] if dur.get(k) else 0.0, alg))
table.sort(key=lambda t: -t[0] * t[2])
symbols = {}
with open(out_csv, 'w') as f:
    f.write('kernel,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_MB_per_launch_corrected(2*FETCH+WRITE),algorithmic_MB_per_launch,traffic_over_algorithmic,mfma_pipe_busy_frac,clock_GHz(GRBM)\n')
    for hbm, k, cnt, fk, wk, bz, clk, alg in table:
        f.write('"%s",%d,%.1f,%.1f,%.2f,%s,%s,%.3f,%.2f\n' % (k, cnt, fk, wk, hbm / 1e6, '%.2f' % (alg / 1e6) if alg else '', '%.2f' % (hbm / alg) if alg else '', bz, clk))
        if short(k):
            symbols[short(k)] = {'hbm_bytes_per_launch': int(hbm), 'dispatches_sampled': cnt, 'FETCH_SIZE_KB_avg': round(fk, 1), 'WRITE_SIZE_KB_avg': round(wk, 1),
                                 'algorithmic_bytes_per_launch': alg, 'traffic_over_algorithmic': round(hbm / alg, 3) if alg else None,
                                 'mfma_pipe_busy_frac': round(bz, 3), 'clock_GHz_GRBM': round(clk, 3)}
json.dump({'kernel_src_sha': bench.kernel_src_sha(), 'mfma_dtype': bline.get('dtype', 'f32'), 'symbols': symbols,
           'correction': 'FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced 16-B/lane reads; MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; units KiB',
           'algorithmic_bytes': 'per launch, from the launch records of %s (every operand and result tensor of the launch once)' % os.path.basename(bench_json),
           'collected_with': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; summarised by scripts/pmc_summary.py',
           'summary_file': out_csv}, open(out_json, 'w'), indent=1)
print(open(out_csv).read()[:3000])
