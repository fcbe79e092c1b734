#!/bin/bash
# The two TCC passes of tools/gpu_evidence.sh alone (refresh profiles/pmc_traffic.json after a kernel source changed):
#   gpurun -- 'bash tools/pmc_traffic_only.sh <tag>'   ->  gpurun_out/<tag>/pmc_traffic.json
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH8="python $PWD/bench.py --batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH8 > $OUT/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $BENCH8 > $OUT/pmc_write.log 2>&1)
python tools/pmc_summary.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/pmc_traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
