#!/bin/bash
# Does the smoke evaluator (side stream) really run UNDER the next batch's sampling?  Kernel trace of a short 2-batch DDIM run of the entry
# script; prints the smoke_rollout dispatches with their start / end relative to the run and how many OTHER kernels started inside each.
#   gpurun -- 'bash tools/overlap_trace.sh r05_d'
TAG=${1:-overlap}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python $ROOT/inference/inference_2d_smoke.py --synthetic True --n_test 128 --batch_size 64 \
    --ddim_sampling_steps 12 --inference_result_path /tmp/ovl_trace > $OUT/overlap_run.log 2>&1)
python - "$OUT" <<'PY'
import csv, glob, sys
out_dir = sys.argv[1]
f = glob.glob(out_dir + "/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
ev = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
with open(out_dir + "/overlap_trace.txt", "w") as out:
    for s, e, name, q, st in ev:
        if "smoke_rollout" in name:
            inside = [x for x in ev if s < x[0] < e and "smoke_rollout" not in x[2]]
            busy = sum(min(x[1], e) - x[0] for x in inside) / 1e6
            out.write(f"smoke_rollout: start {s / 1e6:9.1f} ms, end {e / 1e6:9.1f} ms ({(e - s) / 1e6:.1f} ms), queue {q}, stream {st}; "
                      f"{len(inside)} other kernels started inside it, their durations sum to {busy:.1f} ms; queues of those: {sorted(set(x[3] for x in inside))}\n")
    convs = [x for x in ev if "conv3w" in x[2]]
    out.write(f"run: {ev[-1][1] / 1e6:.1f} ms, {len(ev)} dispatches, {len(convs)} conv3w launches, queues {sorted(set(x[3] for x in ev))}\n")
print(open(out_dir + "/overlap_trace.txt").read())
PY
rm -rf $OUT/tr
