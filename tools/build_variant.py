"""Build an instrumented copy of the library next to the product build:
    python tools/build_variant.py <tag> [extra hipcc flags ...]   ->  diffphycon_amd/lib/libdpc_<tag>.so   (use with DPC_LIB=...)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import build as B  # noqa: E402

tag, extra = sys.argv[1], sys.argv[2:]
flags = list(B.FLAGS)
if "--with-packed-fp32" in extra:        # A/B of the hazard itself: the product flags WITHOUT -packed-fp32-ops
    extra = [e for e in extra if e != "--with-packed-fp32"]
    i = flags.index("-packed-fp32-ops")
    del flags[i - 3:i + 1]               # -Xclang -target-feature -Xclang -packed-fp32-ops
objdir = os.path.join(B.LIBDIR, "_" + tag)
os.makedirs(objdir, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
procs, objs = [], []
for src in B.sources():
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    procs.append(subprocess.Popen([hipcc, *flags, *B.SRC_FLAGS.get(src, []), *extra, "-c", os.path.join(B.CSRC, src), "-o", obj],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    objs.append(obj)
for p in procs:
    out, _ = p.communicate()
    if p.returncode:
        raise SystemExit(out)
lib = os.path.join(B.LIBDIR, f"libdpc_{tag}.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
# the variant carries an honest stamp too (a variant WITH packed fp32 ops then needs DPC_ALLOW_PACKED_FP32=1 to load: _lib._check_build)
n_packed, n_objs = B.scan_packed_fp32(lib)
vstamp = B._stamp_object(n_packed, n_objs, [*flags, *extra], outdir=objdir)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, vstamp])
print(lib, f"({n_packed} packed fp32 instructions)")
