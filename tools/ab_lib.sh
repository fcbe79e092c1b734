#!/bin/bash
# Interleaved A/B of two builds of the library on ONE box: the default libdpc.so against diffphycon_amd/lib/libdpc_<tag>.so
# (tools/build_variant.py <tag> <flags>).    gpurun -- 'bash tools/ab_lib.sh <tag> [out-prefix]'
TAG=$1
PFX=${2:-gpurun_out/ab_$TAG}
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
BB="python bench.py --workload burgers --steps 10 --warmup 3 --no-cpu-baseline"
for i in 1 2; do
  $B > ${PFX}_base_$i.json 2>/dev/null
  DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$TAG.so $B > ${PFX}_${TAG}_$i.json 2>/dev/null
  $BB > ${PFX}_bbase_$i.json 2>/dev/null
  DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$TAG.so $BB > ${PFX}_b${TAG}_$i.json 2>/dev/null
done
python - "$PFX" "$TAG" <<'PY'
import json, sys
pfx, tag = sys.argv[1], sys.argv[2]
for i in (1, 2):
    for t in ("base", tag):
        d = json.load(open(f"{pfx}_{t}_{i}.json"))
        b = d["roofline"]["breakdown_ms_per_step"]
        print(t, i, "S64", round(d["ms_per_step"], 2), {k: v for k, v in b.items() if v > 1})
        d = json.load(open(f"{pfx}_b{t}_{i}.json"))
        print(t, i, "Burgers", round(d["ms_per_step"], 2), {k: v for k, v in d.get("roofline", {}).get("breakdown_ms_per_step", {}).items() if v > 0.9})
PY
