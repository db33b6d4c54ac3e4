cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r5_pw16
python3 scripts/dev/pw16_bench.py > gpurun_out/prof_r5_pw16/bench_plain.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_r5_pw16/sq -- python3 scripts/dev/pw16_bench.py > gpurun_out/prof_r5_pw16/bench.log 2>&1
python3 scripts/dev/sq_counters.py gpurun_out/prof_r5_pw16/sq gpurun_out/prof_r5_pw16/sq_counters.csv pw16 > gpurun_out/prof_r5_pw16/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r5_pw16/fetch -- python3 scripts/dev/pw16_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_r5_pw16/write -- python3 scripts/dev/pw16_bench.py > /dev/null 2>&1
python3 scripts/pmc_kernels.py gpurun_out/prof_r5_pw16/fetch gpurun_out/prof_r5_pw16/write gpurun_out/prof_r5_pw16/hbm.csv > gpurun_out/prof_r5_pw16/hbm.log 2>&1
find gpurun_out/prof_r5_pw16 -name '*counter_collection.csv' -delete; find gpurun_out/prof_r5_pw16 -name '*agent_info.csv' -delete
cat gpurun_out/prof_r5_pw16/bench_plain.log | tail -12; cat gpurun_out/prof_r5_pw16/sq_counters.csv | cut -c1-400; grep pw16 gpurun_out/prof_r5_pw16/hbm.csv | cut -c1-200
