#!/bin/bash
# What bounds the fused attention kernels (tattn3.hip / lattn3.hip): the S64 headline loop's attention classes on the
# product build, on a build whose f16x3 MATRIX instructions are skipped (-DDPC_DBG_NO_MFMA: all VALU work stays) and on a build whose
# operand CONVERSIONS are skipped (-DDPC_DBG_NO_SPLIT: the MFMAs, softmax, LayerNorm stay).  Results of the two variants are invalid:
# timing only.  Build first (on the build host):
#   python tools/build_variant.py nomfma -DDPC_DBG_NO_MFMA; python tools/build_variant.py nosplit -DDPC_DBG_NO_SPLIT
#   gpurun -- 'bash tools/attn_ceiling.sh > gpurun_out/attn_ceiling.log'
B="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
for v in "" nomfma nosplit ""; do
  if [ -n "$v" ]; then export DPC_LIB=$PWD/diffphycon_amd/lib/libdpc_$v.so; else unset DPC_LIB; fi
  $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['roofline']['breakdown_ms_per_step']
print('${v:-product}'.ljust(8), 'step %.1f ms' % d['ms_per_step'], {k: v for k, v in b.items() if 'fused' in k})"
done
