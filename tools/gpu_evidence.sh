#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, the full bench line, rocprofv3 kernel stats and the PMC passes, summarised there.
#   gpurun --timeout 2400 -- 'bash tools/gpu_evidence.sh r02_b'
# Leaves under gpurun_out/<tag>/ only what gets committed to profiles/: gpu_tests.log, bench.json, kernel_stats.csv,
# pmc_traffic.json, pmc_sq.json (the last two also as profiles/pmc_traffic.json / profiles/pmc_sq.json: what bench.py's roofline reads).
TAG=${1:-run}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1)
cp $OUT/stats/s_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# TCC passes: one micro-batch only (B=8 = the same per-launch shapes as B=64), separate passes for FETCH_SIZE and WRITE_SIZE
BENCH8="python $PWD/bench.py --batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH8 > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH8 > $OUT/pmc_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $BENCH > $OUT/pmc_sq.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/pmc_traffic.json
python tools/pmc_summary.py sq $(find $OUT/pmc_sq -name "*counter_collection.csv") $OUT/pmc_sq.json
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
ls -la $OUT
