#!/bin/bash
# Attribution of the Winograd 3x3x3 convolution (csrc/conv3w4.hip, F(4,3); DPC_CONV3W_F43=0: csrc/conv3w.hip, F(2,3)) at the S64 U-Net's
# shapes, micro-batch 32: how much of a launch is the LOADER (halo loads + activation + transform + split + LDS writes by waves 4-7), how
# much the cross-wave EPILOGUE, how much the weight stream.  Needs the attribution build:
#     python tools/build_variant.py dbg -DDPC_ENABLE_CONV_DBG      then     gpurun -- 'bash tools/conv_ceiling.sh > gpurun_out/conv_ceiling.log'
# DPC_CONV_DBG bits (results INVALID, timing only): 32 = the loader does nothing but the barriers (STATIC LDS content: lower power draw),
# 2 = the loader skips its global loads (all VALU work stays, zero operands), 8 = no epilogue (no output transform, no stores),
# 4 = every MFMA wave streams component 0's weights (L1 hits).  F(2,3) only: 64 = loader loads and writes live bits without arithmetic.
export DPC_DEBUG=1 DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_dbg.so
for dbg in ${DBGS:-0 32 2 8 40 4}; do
  echo "==== DPC_CONV_DBG=$dbg"
  DPC_CONV_DBG=$dbg python tools/bench_conv.py 10 32 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
