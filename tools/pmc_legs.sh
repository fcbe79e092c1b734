#!/bin/bash
# HBM counters of the Burgers and training bench legs at their own launch shapes (bench.py reports them as `roofline.traffic` of those
# legs; the S64 headline's come from tools/gpu_evidence.sh / pmc_traffic_only.sh).  Separate FETCH_SIZE / WRITE_SIZE passes, as the
# guide prescribes.      gpurun -- 'bash tools/pmc_legs.sh <tag>'   ->  gpurun_out/<tag>/pmc_traffic_{burgers,train}.json
TAG=${1:-pmc_legs}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp DPC_DEBUG=1 DPC_BURGERS_GRAPH=0
for leg in burgers train; do
  CMD="python $PWD/bench.py --workload $leg --steps 1 --warmup 1 --no-cpu-baseline --no-extras"      # (--no-extras: no x6 leg behind the same class names)
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f_$leg -o f -- $CMD > $OUT/pmc_fetch_$leg.log 2>&1)
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w_$leg -o w -- $CMD > $OUT/pmc_write_$leg.log 2>&1)
  python tools/pmc_summary.py $(find $OUT/f_$leg -name "*counter_collection.csv") $(find $OUT/w_$leg -name "*counter_collection.csv") $OUT/pmc_traffic_$leg.json
  rm -rf $OUT/f_$leg $OUT/w_$leg
done
