#!/bin/bash
# Run ON THE GPU BOX: the precision modes through collect_profiles.sh (bench line, rocprofv3 --kernel-trace --stats, the three PMC
# passes, pmc_summary.py), then the bench lines that carry the fresh traffic files.
#   bash scripts/collect_all.sh r3 [modes...]      (default modes: f32 bf16 bf16s)
# Afterwards, here: python scripts/keep_profiles.py r3   copies gpurun_out/prof_<tag>*/ summaries into profiles/.
TAG="${1:-r3}"; shift
MODES="${@:-f32 bf16 bf16s}"
cd $GRAFT_REPO_ROOT
for m in $MODES; do
  if [ "$m" = f32 ]; then
    bash scripts/collect_profiles.sh ${TAG} > gpurun_out/collect_${TAG}.log 2>&1
    cp gpurun_out/prof_${TAG}/traffic.json profiles/${TAG}_traffic.json
    python3 bench.py > gpurun_out/prof_${TAG}/bench_final.json 2> gpurun_out/prof_${TAG}/bench_final.err
  else
    bash scripts/collect_profiles.sh ${TAG}_$m --mfma-dtype $m > gpurun_out/collect_${TAG}_$m.log 2>&1
    cp gpurun_out/prof_${TAG}_$m/traffic.json profiles/${TAG}_${m}_traffic.json
    python3 bench.py --mfma-dtype $m --no-cpu-baseline > gpurun_out/prof_${TAG}_$m/bench_final.json 2> gpurun_out/prof_${TAG}_$m/bench_final.err
  fi
done
for m in $MODES; do d=gpurun_out/prof_${TAG}; [ "$m" != f32 ] && d=${d}_$m; echo "== $m"; tail -c 700 $d/bench_final.json; echo; head -12 $d/pmc_summary.csv; done
