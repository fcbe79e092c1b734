#!/bin/bash
# r05: the K = 32 contractions of the fused temporal attention (S^T = K Q^T, O^T = V^T P) on the exact fp32 matrix
# instruction instead of f16x3 -- interleaved A/B of the S64 headline loop's attention classes on ONE box, product build first and last.
#   python tools/build_variant.py qk -DDPC_TATTN_QK_F32=1; python tools/build_variant.py pv -DDPC_TATTN_PV_F32=1
#   python tools/build_variant.py qkpv -DDPC_TATTN_QK_F32=1 -DDPC_TATTN_PV_F32=1
#   gpurun -- 'bash tools/attn_f32_ab.sh > gpurun_out/attn_f32_ab.log'
B="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
for v in "" qk pv qkpv "" qk pv qkpv ""; do
  if [ -n "$v" ]; then export DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$v.so; else unset DPC_LIB; fi
  $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['roofline']['breakdown_ms_per_step']
print('${v:-product}'.ljust(8), 'step %.1f ms' % d['ms_per_step'], {k: v for k, v in b.items() if 'attention' in k})"
done
# parity of the variants: the fused-attention tests of the GPU suite against each variant library
for v in qk pv qkpv; do
  echo "== tests on libdpc_$v.so"
  DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$v.so python -m pytest tests/test_gpu_unet3d.py -m gpu -q -k "fixture or temporal or full_width" 2>&1 | tail -3
done
