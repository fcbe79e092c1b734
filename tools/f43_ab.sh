#!/bin/bash
# A/B of the Winograd F(4,3) form (csrc/conv3w4.hip, default) against F(2,3) (csrc/conv3w.hip) at the S64 U-Net's conv shapes, micro-batch 32
export DPC_DEBUG=1
for f43 in 1 0 1 0; do
  echo "==== DPC_CONV3W_F43=$f43"
  DPC_CONV3W_F43=$f43 python tools/bench_conv.py 10 32 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
