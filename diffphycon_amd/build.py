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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# Kernels that must compile WITHOUT register spills: a spilling build of the 128-register implicit-GEMM kernel has (twice) given
# batch-size dependent results at full size; the build fails instead of shipping one.
# conv3w_kernel: its loader waves count their own vmcnt -- a compiler-inserted scratch reload there waits for every load in flight.
NO_SPILL = {"igemm6.hip": ("igemm3_kernel",), "conv3w.hip": ("conv3w_kernel",), "igemm_wide.hip": ("igemm3w_kernel",),
            "igemm_panel.hip": ("igemm3p_kernel",), "igemm_tile.hip": ("igemm3t_kernel",), "stem7x6.hip": ("stem7p_kernel",)}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


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
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
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
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
