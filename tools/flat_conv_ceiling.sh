#!/bin/bash
# How loader-bound is the (1,3,3) form of conv3f3c (Burgers levels 0 / 1, the jellyfish surrogates)?  The shapes of tools/bench_igemm.py "flat"
# on the attribution build: product arithmetic vs bit 64 (loads + LDS writes, NO conversion arithmetic, live operands) vs bit 32-like zero data.
#   python tools/build_variant.py dbg -DDPC_ENABLE_CONV_DBG;  gpurun -- 'bash tools/flat_conv_ceiling.sh > gpurun_out/flat_conv_ceiling.log'
export DPC_DEBUG=1 DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_dbg.so
for dbg in ${FLAT_DBG_BITS:-0 64 2 0 64}; do
  echo "==== DPC_CONV_DBG=$dbg"
  DPC_CONV_DBG=$dbg python tools/bench_igemm.py flat 20 2>&1 | grep -v "^$\|amdgpu.ids"
done
