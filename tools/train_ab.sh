#!/bin/bash
# Interleaved A/B of the training step on ONE box: the product library against diffphycon_amd/lib/libdpc_<tag>.so.
#   gpurun -- 'bash tools/train_ab.sh wg3old > gpurun_out/train_ab.log'
TAG=$1
B="python bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
for v in "" $TAG "" $TAG; do
  if [ -n "$v" ]; then export DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$v.so; else unset DPC_LIB; fi
  $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['roofline']['breakdown_ms_per_step']
print('${v:-product}'.ljust(8), 'train step %.1f ms' % d['ms_per_step'], {k: v for k, v in b.items() if 'wgrad' in k}, 'frac', round(d['roofline']['frac'], 3))"
done
