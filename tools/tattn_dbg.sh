#!/bin/bash
# time the fused temporal attention kernel under DPC_TATTN_DBG experiment bits (one micro-batch of the real step)
export TMPDIR=/tmp
ROOT=$PWD
for D in "$@"; do
  O=$ROOT/gpurun_out/tdbg/$D; mkdir -p $O
  (cd /tmp && DPC_TATTN_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $ROOT/bench.py --batch 8 --micro-batch 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/log 2>&1)
  echo "dbg=$D $(grep 'attn6_kernel<64' $O/s_kernel_stats.csv | cut -d, -f2-4,7- | head -3 | tr '\n' ' ')"
  rm -f $O/*trace.csv
done
