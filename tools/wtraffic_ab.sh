#!/bin/bash
# r06 gate of the F(4,3) design: `conv3w` (F(2,3)) with every weight fragment requested TWICE (attribution bit 1024 of csrc/conv3w.hip: the
# second request is an LDS-DMA into the unused tail of a halo buffer) -- the L2 -> L1 weight bytes per MFMA that a 4-wave F(4,3) form pays.
# Needs the attribution build:  python tools/build_variant.py dbg -DDPC_ENABLE_CONV_DBG       (profiles/r06_a_wtraffic_ab.log)
export DPC_DEBUG=1 DPC_CONV3W_F43=0 DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_dbg.so
for dbg in 0 1024 0 1024 4; do
  echo "==== DPC_CONV_DBG=$dbg"
  DPC_CONV_DBG=$dbg python tools/bench_conv.py 10 32 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
