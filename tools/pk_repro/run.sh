#!/bin/bash
# Build (on the build host: hipcc cross-compiles) and run (on the GPU box) the packed-fp32 reproducer in both forms.
#   bash tools/pk_repro/run.sh build        gpurun -- 'bash tools/pk_repro/run.sh run 400 > gpurun_out/pk_repro.log'
D=$(dirname $0)
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $D/pk_repro.hip -o $D/pk_repro_pk --save-temps=obj 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $D/pk_repro.hip -o $D/pk_repro_pk
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops $D/pk_repro.hip -o $D/pk_repro_nopk 2>/dev/null
  for v in pk nopk; do
    echo "$v: $(/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $D/pk_repro_$v > /dev/null 2>&1; for f in $D/pk_repro_$v.*gfx950*; do /opt/rocm/lib/llvm/bin/llvm-objdump -d $f; done | grep -c 'v_pk_\(mul\|add\|fma\)_f32') packed fp32 instructions in the ISA"
    rm -f $D/pk_repro_$v.*-gfx950* $D/pk_repro_$v.*host*
  done
  rm -f $D/*.bc $D/*.s $D/*.o $D/*.hipi $D/*.out $D/*.hipfb $D/*.cui $D/*.txt 2>/dev/null
else
  for v in pk nopk pk nopk; do echo "== pk_repro_$v"; timeout 600 $D/pk_repro_$v ${2:-400}; done
fi
