#!/bin/bash
# Accounts for EVERY microsecond of a guided J128 step (~95 ms of the 554 ms step were in neither bucket of
# tools/j128_profile.py): rocprofv3 --kernel-trace --stats around the bench leg (the entry script's pipeline at 4 + 4 + 20 = 28 guided
# steps, batch 16), every kernel -- libdpc's AND torch's own (copies, cats, element-wise glue) -- summed by name and divided by 28;
# the leg's own wall-clock per step is printed beside it, the difference is GPU idle time (host-side gaps).
#   gpurun -- 'bash tools/j128_step_trace.sh r05_a'      -> gpurun_out/<tag>/j128_step_trace.txt
TAG=${1:-j128}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o t -- python $ROOT/bench.py --workload j128 > $OUT/j128_bench.json 2> $OUT/j128_bench.err)
python - "$OUT" <<'PY'
import csv, glob, json, sys
out_dir = sys.argv[1]
f = glob.glob(out_dir + "/tr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 28.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
try:
    leg = json.loads(open(out_dir + "/j128_bench.json").read().strip().splitlines()[-1])
    ms = leg["ms_per_step"]
except Exception as e:           # noqa: BLE001
    ms = float("nan")
with open(out_dir + "/j128_step_trace.txt", "w") as out:
    out.write(f"bench leg: {ms:.1f} ms per guided step (difference of the 20- and 4-step pipeline runs); all kernels of the 28 steps + set-up: "
              f"{tot:.1f} ms = {tot / steps:.1f} ms per step if set-up were free\n")
    out.write(f"{'ms/step':>9s} {'calls/step':>10s} {'avg us':>9s}  kernel\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
        t = float(r["TotalDurationNs"]) / 1e6
        out.write(f"{t / steps:9.2f} {float(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:150]}\n")
print(open(out_dir + "/j128_step_trace.txt").read()[:6000])
PY
rm -rf $OUT/tr
