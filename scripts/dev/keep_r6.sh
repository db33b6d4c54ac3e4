#!/bin/bash
# HERE (after `gpurun -- bash scripts/dev/collect_r6.sh`): copy what is judged into profiles/ and regenerate the lists derived from it
set -e
cd "$(dirname "$0")/../.."
python scripts/keep_profiles.py r6 > /dev/null
P=gpurun_out/prof_r6
for b in 64 128 256; do cp $P/bench_b$b.json profiles/r6_bench_b${b}_n1.json; cp $P/bench_b${b}_direct.json profiles/r6_bench_b${b}_n1_direct.json; done
cp $P/bench_direct.json profiles/r6_bench_n1_direct.json
cp $P/bench_bf16.json profiles/r6_bf16_bench_n1.json; cp $P/bench_bf16s.json profiles/r6_bf16s_bench_n1.json
grep -v amdgpu.ids $P/wino_clock.md > profiles/r6_wino_clock.md
grep -v amdgpu.ids $P/wino_bench.txt > profiles/r6_wino_per_layer.txt
for s in 1 2 3; do
  Q=gpurun_out/prof_wino_$s
  cp $Q/sq_counters.csv profiles/r6_wino_shape${s}_sq_counters.csv
  cp $Q/sq2_counters.csv profiles/r6_wino_shape${s}_sq2_counters.csv
  grep -i "Kernel\|wino\|igemm" $Q/hbm.csv > profiles/r6_wino_shape${s}_hbm.csv
done
python scripts/make_symbol_lists.py 6
python - <<'P'
import json
for f in ['r6_bench_n1','r6_bench_n1_direct','r6_bf16_bench_n1','r6_bf16s_bench_n1','r6_bench_b64_n1','r6_bench_b128_n1','r6_bench_b256_n1','r6_bench_b64_n1_direct','r6_bench_b128_n1_direct','r6_bench_b256_n1_direct']:
    d = json.loads([l for l in open('profiles/%s.json' % f) if l.startswith('{')][-1]); r = d.get('roofline', {})
    print(f, d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), d.get('kernel_src_sha'), d.get('step_mfma_frac'), d.get('step_mfma_frac_executed'))
P
cat tf_face_toolbox_amd/csrc/obj/src_sha.txt
