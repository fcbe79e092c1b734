#!/bin/bash
# Interleaved A/B of an environment setting on ONE box (S64 bench, 5 timed steps, twice each):
#   gpurun -- 'bash tools/ab_env.sh "DPC_DEBUG=1 DPC_TWO_STREAMS=1" gpurun_out/ab_two_streams'
SETTING=$1
PFX=${2:-gpurun_out/ab_env}
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
for i in 1 2; do
  $B > ${PFX}_base_$i.json 2>/dev/null
  env $SETTING $B > ${PFX}_set_$i.json 2>/dev/null
done
python - "$PFX" "$SETTING" <<'PY'
import json, sys
pfx = sys.argv[1]
print("setting:", sys.argv[2])
for i in (1, 2):
    for t in ("base", "set"):
        d = json.load(open(f"{pfx}_{t}_{i}.json"))
        r = d["roofline"]
        print(t, i, "ms/step", round(d["ms_per_step"], 2), "conv class avg launch ms", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 3))
PY
