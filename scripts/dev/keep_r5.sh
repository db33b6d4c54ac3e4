#!/bin/bash
# HERE (after `gpurun -- bash scripts/dev/collect_r5.sh; bash scripts/dev/pw16_pmc.sh`): copy what is judged into profiles/ and regenerate
# the lists and tables that are derived from it
set -e
cd "$(dirname "$0")/../.."
python scripts/keep_profiles.py r5 > /dev/null
P=gpurun_out/prof_r5
for b in 64 128 256; do cp $P/bench_b$b.json profiles/r5_bench_b${b}_n1.json; cp $P/bench_b${b}_nosk.json profiles/r5_bench_b${b}_n1_no_streamk.json; done
cp $P/kernel_stats_b64.csv profiles/r5_bench_b64_kernel_stats.csv
Q=gpurun_out/prof_r5_pw16
cp $Q/sq_counters.csv profiles/r5_pw16_sq_counters.csv; cp $Q/hbm.csv profiles/r5_pw16_hbm.csv
tail -10 $Q/bench_plain.log | grep -v amdgpu.ids > profiles/r5_pw16_bench.txt
python scripts/make_symbol_lists.py 5
python scripts/hbm_table.py r5 > profiles/r5_hbm_bound_kernels.md
python - <<'P'
import json
for f in ['r5_bench_n1','r5_bf16_bench_n1','r5_bf16s_bench_n1','r5_bench_b64_n1','r5_bench_b128_n1','r5_bench_b256_n1','r5_bench_b64_n1_no_streamk','r5_bench_b128_n1_no_streamk','r5_bench_b256_n1_no_streamk']:
    d = json.load(open('profiles/%s.json' % f)); r = d.get('roofline', {})
    print(f, d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), d.get('kernel_src_sha'))
P
for t in resnext50_bf16s_b128 senet50_bf16s_b128 shufflenet_b256 resnet50_bf16s_b128; do tail -3 profiles/r5_${t}_step_roofline.md | head -1 | cut -c1-200; done
cat tf_face_toolbox_amd/csrc/obj/src_sha.txt
