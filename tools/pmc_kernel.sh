#!/bin/bash
# Stall attribution of selected kernels inside the real step (one micro-batch): three SQ counter passes, per-kernel sums.
#   gpurun -- 'bash tools/pmc_kernel.sh <tag> "<kernel regex>"'
TAG=${1:-pk}
PAT=${2:-attn}
ROOT=$PWD
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BARGS=${3:---batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-extras}
B=${PMC_CMD:-"python $ROOT/bench.py $BARGS"}   # PMC_CMD="python tools/bench_igemm.py burgers 3": any other command      # e.g. third argument "--workload train --steps 1 --warmup 1 --no-cpu-baseline" for the training kernels
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
G2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
G3="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_ACTIVE_INST_MISC"
G4="SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o q -- $B > $OUT/g$i.log 2>&1)
done
python - <<PY
import csv, collections, re, glob
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("$OUT/g*/q_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not re.search(r"$PAT", k): continue
        key=(k.split("(")[0][-36:], r["Grid_Size"])
        acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[key][r["Counter_Name"]]+=1
for key,v in sorted(acc.items()):
    wc=v["SQ_WAVE_CYCLES"]/max(n[key]["SQ_WAVE_CYCLES"],1)
    print(key, "launches", n[key]["SQ_WAVES"], "wave_cycles/launch %.3g" % wc)
    print("   ", " ".join(f"{c.replace('SQ_','')}={v[c]/n[key][c]/wc:.3f}" for c in sorted(v) if c!="SQ_WAVE_CYCLES"))
    print("    per wave:", " ".join(f"{c.replace('SQ_','')}={v[c]/max(v['SQ_WAVES'],1)*n[key]['SQ_WAVES']/n[key][c]:.0f}" for c in sorted(v) if c.startswith("SQ_INSTS") or c=="SQ_WAVE_CYCLES"))
PY
rm -f $OUT/g*/*_kernel_trace.csv $OUT/g*/*counter_collection.csv
