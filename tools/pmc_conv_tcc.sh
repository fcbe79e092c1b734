#!/bin/bash
# HBM traffic of the conv kernels on the micro-benchmark
TAG=${1:-conv_tcc}
ROOT=$PWD
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- python $ROOT/tools/bench_conv.py 2 > $OUT/f.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- python $ROOT/tools/bench_conv.py 2 > $OUT/w.log 2>&1)
(cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/h -o h -- python $ROOT/tools/bench_conv.py 2 > $OUT/h.log 2>&1)
python - <<PY
import csv, collections, os
for sub, f in (("f","f"),("w","w"),("h","h")):
    path="$OUT/%s/%s_counter_collection.csv"%(sub,f)
    if not os.path.exists(path): print("missing",path); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(path)):
        k=r["Kernel_Name"]
        if "conv3" not in k or "pack" in k: continue
        key=(k.split("(")[0][-30:], r["Grid_Size"])
        acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
    for key,v in acc.items():
        print(sub, key, {c: round(x/n[(key,c)],1) for c,x in v.items()})
PY
rm -f $OUT/*/*_kernel_trace.csv
