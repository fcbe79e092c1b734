#!/bin/bash
# every launch (in order) of the kernels whose name contains $2 in a traced command:  bash tools/trace_list.sh <tag> <substr> <cmd...>
TAG=$1; SUB=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- "$@" > $OUT/t.log 2>&1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/t/t_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    if "$SUB" in r["Kernel_Name"]:
        print(f'{r["Kernel_Name"].split("(")[0][-34:]:36s} grid {r.get("Grid_Size", r.get("Grid_Size_X","?")):>9s} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f} us')
PY
rm -f $OUT/t/t_kernel_trace.csv
