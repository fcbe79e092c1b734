#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, bench line, rocprofv3 kernel stats and the two PMC passes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh r01_c'
# Writes everything under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (tools/pmc_summary.py).
TAG=${1:-run}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1
tail -3 $OUT/gpu_tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1)
# TCC passes: one micro-batch only (B=8 = the same per-launch shapes as B=64; the B=64 TCC pass crashed rocprofv3 in r01_c)
BENCH8="python $PWD/bench.py --batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH8 > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH8 > $OUT/pmc_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $BENCH > $OUT/pmc_sq.log 2>&1)
# keep the pull small: the traces are large, the per-kernel aggregates are what gets committed
rm -f $OUT/*/*_kernel_trace.csv
ls -la $OUT $OUT/*
