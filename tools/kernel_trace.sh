#!/bin/bash
# Per-dispatch kernel trace of one bench command (run on the GPU box via gpurun): duration, grid, LDS and kernel name of every
# launch in time order -> gpurun_out/<tag>/trace_summary.txt.
#   gpurun -- 'bash tools/kernel_trace.sh r03_i --batch 16 --micro-batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-extras'
#   TRACE_CMD="python tools/j128_profile.py 128 16" bash tools/kernel_trace.sh <tag>      (any other command instead of bench.py)
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$TRACE_CMD" ]; then
  (cd $ROOT && rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- $TRACE_CMD > $OUT/trace.log 2>&1)
else
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python $ROOT/bench.py "$@" > $OUT/trace.log 2>&1)
fi
python - "$OUT" <<'PY'
import csv, glob, sys
out_dir = sys.argv[1]
f = glob.glob(out_dir + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(out_dir + "/trace_summary.txt", "w") as out:
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.write(f"{d:9.1f} us grid {r.get('Grid_Size_X', '?'):>9s} wg {r.get('Workgroup_Size_X', '?'):>5s} lds {r.get('LDS_Block_Size', '?'):>7s} "
                  f"{r['Kernel_Name'][:110]}\n")
PY
rm -rf $OUT/tr
