export DPC_DEBUG=1 DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_dbg.so
for dbg in 0 1024 0 1024 4; do
  echo "==== DPC_CONV_DBG=$dbg"
  DPC_CONV_DBG=$dbg python tools/bench_conv.py 10 32 2>&1 | grep -v "^$" | grep -v amdgpu.ids
done
