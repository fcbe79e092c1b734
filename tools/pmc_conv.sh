#!/bin/bash
# SQ stall attribution of the conv kernels on the micro-benchmark (gpurun -- 'bash tools/pmc_conv.sh <tag>')
TAG=${1:-conv}
ROOT=$PWD
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq -o q -- python $ROOT/tools/bench_conv.py 3 > $OUT/sq.log 2>&1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open("$OUT/sq/q_counter_collection.csv")):
    k=r["Kernel_Name"]
    if "conv3" not in k or "pack" in k: continue
    key=(k.split("(")[0][-40:], r["Grid_Size"])
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[key]+=1
for key,v in acc.items():
    wc=v["SQ_WAVE_CYCLES"]
    print(key, n[key], " ".join(f"{c.replace('SQ_','')}={v[c]/wc:.3f}" for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_LDS")), f"ldsconf/active={v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.3f}", f"mfma_busy/wavecyc={v['SQ_VALU_MFMA_BUSY_CYCLES']/wc/4:.3f}")
PY
rm -f $OUT/sq/*_kernel_trace.csv
