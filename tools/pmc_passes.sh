TAG=r01_g
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
BENCH8="python $PWD/bench.py --batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH8 > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH8 > $OUT/pmc_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $BENCH > $OUT/pmc_sq.log 2>&1)
rm -f $OUT/*/*_kernel_trace.csv
python tools/pmc_summary.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_traffic.json
python tools/pmc_summary.py sq $OUT/pmc_sq/q_counter_collection.csv $OUT/pmc_sq.json
rm -f $OUT/pmc_*/*counter_collection.csv
