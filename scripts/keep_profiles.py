"""Copies what the judge reads from gpurun_out/prof_<tag>[_mode]/ (scripts/collect_all.sh, run on the GPU box) into profiles/:
kernel stats, PMC summary, traffic file, bench lines.   python scripts/keep_profiles.py r3"""
import os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ('', '_bf16', '_bf16s'):
    d = os.path.join(root, 'gpurun_out', 'prof_%s%s' % (tag, mode))
    if not os.path.isdir(d):
        continue
    pre = os.path.join(root, 'profiles', tag + mode + '_')
    for src, dst in (('kernel_stats.csv', 'bench_kernel_stats.csv'), ('pmc_summary.csv', 'pmc_hbm_traffic_summary.csv'),
                     ('traffic.json', 'traffic.json'), ('bench_final.json', 'bench_n1.json'), ('bench.json', 'bench_n1_first.json')):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), pre + dst)
            print('kept', pre + dst)
# net profiles (scripts/collect_net_profiles.sh TAG): step-level roofline, per-kernel PMC summary, kernel stats, step summary
import glob
for d in glob.glob(os.path.join(root, 'gpurun_out', 'prof_%s_*' % tag)):
    name = os.path.basename(d)[len('prof_'):]
    if not os.path.exists(os.path.join(d, 'step_roofline.md')):
        continue
    for src in ('step_roofline.md', 'pmc_summary.csv', 'kernel_stats.csv', 'step_summary.txt'):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(root, 'profiles', name + '_' + src))
            print('kept', name + '_' + src)
