#!/bin/bash
# Attribution of the Winograd 3x3x3 convolution (csrc/conv3w.hip) at the S64 U-Net's shapes, micro-batch 32: how much of a launch is
# the LOADER (halo loads + activation + transform + split + LDS writes by waves 4-7), how much the cross-wave EPILOGUE -- i.e. the
# most any re-design of either could gain (why no W-axis Winograd / tile variant was built in r04).  Needs the attribution
# build:  python tools/build_variant.py dbg -DDPC_ENABLE_CONV_DBG      then     gpurun -- 'bash tools/conv_ceiling.sh > gpurun_out/conv_ceiling.log'
# DPC_CONV_DBG bits (results INVALID, timing only): 32 = the loader does nothing but the barriers, 2 = the loader skips its global loads
# (all VALU work stays), 64 = the loader loads and writes live bits but does no activation / transform / split (r05: the honest
# ceiling of a producer + LDS-DMA design; 32 leaves STATIC LDS content, which lowers the power draw), 8 = no epilogue (no output transform, no stores), 4 = every MFMA wave streams component 0's weights.
export DPC_DEBUG=1 DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_dbg.so
for dbg in 0 64 32 2 8 40 4; do
  echo "==== DPC_CONV_DBG=$dbg"
  DPC_CONV_DBG=$dbg python tools/bench_conv.py 10 32 2>&1 | grep -v "^$"
done
