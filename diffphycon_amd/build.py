"""Builds libdpc.so (all HIP sources, gfx950 only) in-tree with hipcc.  `python -m diffphycon_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdpc.so")
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 adds / muls into v_pk_*_f32, which beside an MFMA stream costs more
# issue time than the two scalar ops (r02: smoke step 303.7 -> 298.5 ms with the flag; MI355X_MICROARCH.md says the same)
# -packed-fp32-ops (r04): NO v_pk_{mul,add,fma}_f32 anywhere in the library.  With them, kernels gave wrong results in a few per cent of
# launches whenever ANOTHER kernel was resident on the GPU at the same time (a second stream or a second process): r02 saw it in
# ln_apply (`v_pk_mul_f32 ... op_sel` on a freshly loaded (mean, rstd) pair: wrong low lanes, worked around locally), r03 in the two-rank
# entry-script tests (2 of 70 Burgers runs; attributed to an MFMA operand re-load), r04 pinned it down: the K = 32 qkv projection behind
# a LayerNorm (the same (x - mean) * rstd * gamma expression in the implicit GEMM's prologue) next to a ConvTranspose on a second stream
# -- 11 of 400 repetitions wrong in lanes 48..63, 0 of 400 with this flag; whole U-Net forwards on two streams 13 of 400 -> 0 of 400,
# the 2-D net 3 of 400 -> 0 of 400 (tools/det_ops2.py, two_stream_bisect*.py; profiles/r04_packed_fp32_*.log).  No such
# hazard is in the ISA guide; an idle GPU never shows it.  Speed: neutral (r03 A/B 265.1 vs 264.9 ms per S64 step; r04: profiles/).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# Per-file additions.  The fused attention kernels are bound by their VALU instruction count (~16 VALU instructions per
# MFMA): there, and only there, a multiply that feeds an add may contract into one v_fma_f32 -- nothing in those kernels is compared bit
# for bit with a reference (their results are checked to 1e-5 of the output range).  Everywhere else contraction stays off: the update
# kernels and the PDE evaluators are bit-exact restatements of the reference's fp32 arithmetic.
SRC_FLAGS = {"tattn3.hip": ["-ffp-contract=fast"], "lattn3.hip": ["-ffp-contract=fast"]}
# Kernels that must compile WITHOUT register spills: a spilling build of the 128-register implicit-GEMM kernel has (twice) given
# batch-size dependent results at full size; the build fails instead of shipping one.
# conv3w_kernel: its loader waves count their own vmcnt -- a compiler-inserted scratch reload there waits for every load in flight.
NO_SPILL = {"igemm6.hip": ("igemm3_kernel",), "conv3w.hip": ("conv3w_kernel",), "conv3w4.hip": ("conv3w4_kernel",), "conv3f3c.hip": ("conv3f3c_kernel",), "igemm_wide.hip": ("igemm3w_kernel",),
            "igemm_panel.hip": ("igemm3p_kernel",), "igemm_tile.hip": ("igemm3t_kernel",), "stem7x6.hip": ("stem7p_kernel",), "igemm_img.hip": ("igemm3i_kernel",)}


LLVM_BIN = os.environ.get("DPC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PACKED_FP32 = r"\bv_pk_(mul|add|fma)_f32\b"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def scan_packed_fp32(path):
    """Number of packed fp32 VALU instructions (v_pk_{mul,add,fma}_f32) in the gfx950 code objects of a linked library / object file:
    the fat binary is unbundled into a temporary directory (llvm-objdump --offloading writes next to its input -- never into lib/)
    and every device ELF is disassembled.  DESIGN.md 6.2: with these instructions a kernel gives wrong lanes in a few per cent of
    launches whenever a second kernel is resident; the product library must contain none (`_lib.lib()` checks the stamp below)."""
    import re
    import shutil
    import tempfile
    tool = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(tool):
        raise RuntimeError(f"{tool} not found: the build disassembles the linked device code to stamp the library (packed fp32 "
                           "instruction count); point DPC_LLVM_BIN at the directory that holds llvm-objdump")
    d = tempfile.mkdtemp(prefix="dpc_scan_")
    try:
        tmp = os.path.join(d, os.path.basename(path))
        shutil.copy(path, tmp)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "amdgcn" in f]
        if not objs:
            raise RuntimeError(f"{path}: no gfx950 code object found in the fat binary")
        n = 0
        for o in objs:
            asm = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", o], check=True, stdout=subprocess.PIPE, text=True).stdout
            n += len(re.findall(PACKED_FP32, asm))
        return n, len(objs)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _stamp_object(n_packed, n_objs, flags, outdir=None):
    """lib/build_stamp.o: `dpc_build_info()` (include/dpc.h) -- what the build MEASURED on the linked device code, not what it was asked
    to do: the count of packed fp32 instructions found by scan_packed_fp32 and the flags of the compile."""
    outdir = outdir or LIBDIR
    src = os.path.join(outdir, "build_stamp.c")
    text = f"packed_fp32_insts={n_packed};code_objects={n_objs};flags={' '.join(flags)}".replace("\\", "/").replace('"', "'")
    with open(src, "w") as f:
        f.write('/* generated by diffphycon_amd/build.py */\nconst char* dpc_build_info(void) { return "' + text + '"; }\n')
    obj = os.path.join(outdir, "build_stamp.o")
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O1", "-fPIC", "-c", src, "-o", obj])
    return obj


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dpc.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link lib/libdpc.so (cross-compiles without a GPU)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DPC_EXTRA_FLAGS", "").split()       # e.g. -DDPC_CONV_STAMPS for tools/conv_stamps.py
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *SRC_FLAGS.get(src, []), *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if src in NO_SPILL:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if src in NO_SPILL:
            name, bad = None, []
            for line in out.splitlines():
                if "Function Name:" in line:
                    name = line.split("Function Name:")[1].split()[0]
                elif "VGPRs Spill:" in line and name and any(k in name for k in NO_SPILL[src]):
                    if int(line.split("VGPRs Spill:")[1].split()[0]) != 0:
                        bad.append(name)
            if bad:
                raise RuntimeError(f"{src}: register spills in {bad}: this kernel must not spill (see NO_SPILL in build.py)")
            out = "\n".join(l for l in out.splitlines() if "-Rpass-analysis" not in l and not l.startswith(" ") )
        # (the host pass of hipcc does not know the device-only feature name and says so once per file)
        out = "\n".join(l for l in out.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
        if verbose and out.strip():
            print(out)
    # link once without the stamp, measure the device code of THAT file, then link the stamp in (the device code is unchanged by it)
    pre = LIB + ".prestamp"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", pre, *objs])
    n_packed, n_objs = scan_packed_fp32(pre)
    os.remove(pre)
    stamp = _stamp_object(n_packed, n_objs, [*FLAGS, *extra])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, stamp]
    if verbose:
        print(" ".join(cmd), flush=True)
        print(f"device code: {n_objs} code objects, {n_packed} packed fp32 instructions", flush=True)
    subprocess.check_call(cmd)
    if n_packed and os.environ.get("DPC_ALLOW_PACKED_FP32") != "1":
        raise RuntimeError(f"{LIB}: {n_packed} packed fp32 VALU instructions in the device code (build flags lost? DESIGN.md 6.2); "
                           "DPC_ALLOW_PACKED_FP32=1 builds and loads such a library for A/B experiments")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
