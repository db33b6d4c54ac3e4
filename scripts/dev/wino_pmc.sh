# SQ counters + HBM traffic of the Winograd kernels on one resBlock shape (default stage 3 at 512 images)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SH=${1:-2}; N=${2:-512}; D=gpurun_out/prof_wino_$SH
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 scripts/dev/wino_bench.py $N 3 $SH > $D/bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $D/sq -- python3 scripts/dev/wino_bench.py $N 3 $SH > /dev/null 2>&1
python3 scripts/dev/sq_counters.py $D/sq $D/sq_counters.csv wino > $D/sq.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $D/sq2 -- python3 scripts/dev/wino_bench.py $N 3 $SH > /dev/null 2>&1
python3 scripts/dev/sq_counters.py $D/sq2 $D/sq2_counters.csv wino > $D/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 scripts/dev/wino_bench.py $N 3 $SH > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 scripts/dev/wino_bench.py $N 3 $SH > /dev/null 2>&1
python3 scripts/pmc_kernels.py $D/fetch $D/write $D/hbm.csv > $D/hbm.log 2>&1
find $D -name '*counter_collection.csv' -delete; find $D -name '*agent_info.csv' -delete; find $D -name '*kernel_trace.csv' -delete
grep winograd $D/bench.log; cat $D/sq_counters.csv | cut -c1-600; cat $D/sq2_counters.csv | cut -c1-600; grep -i "wino\|Kernel" $D/hbm.csv | cut -c1-300; find $D -name '*kernel_stats.csv' | head -1 | xargs head -12 | cut -c1-200
