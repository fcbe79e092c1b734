#!/bin/bash
# per-(kernel, grid) time breakdown of a command:  bash tools/trace_by_grid.sh <tag> <cmd...>
TAG=$1; shift
ROOT=$PWD
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- "$@" > $OUT/t.log 2>&1)
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open("$OUT/t/t_kernel_trace.csv")):
    k=r["Kernel_Name"].split("(")[0][-28:]
    key=(k, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))
    a=acc[key]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
tot=sum(v[1] for v in acc.values())
for key,v in sorted(acc.items(), key=lambda kv:-kv[1][1])[:28]:
    print(f"{key[0]:30s} grid {key[1]:>9s}  n {v[0]:5d}  total {v[1]:9.2f} ms  avg {v[1]/v[0]*1e3:9.1f} us  {100*v[1]/tot:5.1f}%")
print("total", tot)
PY
rm -f $OUT/t/t_kernel_trace.csv
