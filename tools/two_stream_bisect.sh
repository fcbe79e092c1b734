#!/bin/bash
# tools/two_stream_bisect.py under the kernel-selection switches: which family's kernels make a forward depend on a concurrent stream?
export DPC_DEBUG=1
for sw in "" "DPC_UNFUSED_ATTN=1" "DPC_CONV3W=0" "DPC_CONV3W=0 DPC_CONV3F3C=0" "DPC_IGEMM_PANEL=0" "DPC_IGEMM_TILE=0" "DPC_IGEMM_PANEL=0 DPC_IGEMM_TILE=0 DPC_IGEMM_LDSB=0" "DPC_STEM_PAIRS=0" "DPC_UNFUSED_GN=1" "DPC_FUSE_GN_RES=0" "DPC_STEM_MODE=x6" "DPC_ATTN_MODE=x6" "DPC_IGEMM_MODE=x6" "DPC_CONV_MODE=x6"; do
  echo "==== $sw"
  env $sw python tools/two_stream_bisect.py ${1:-30} ${2:-32} ${3:-8} ${4:-32} 2>&1 | grep -v amdgpu.ids | tail -4
done
