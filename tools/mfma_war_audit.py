"""Static audit of the hazard of DESIGN.md 6.2: an LDS / global load whose destination registers were an operand of a recently issued
MFMA.  The matrix pipe reads an MFMA's B operand while the instruction executes, not when it issues; a load that is issued right
behind the MFMA and returns early (or an MFMA held up by a foreign wave's MFMAs) overwrites the operand.  The kernels keep a distance
by construction (scheduling barriers, fragment rings); this tool measures it in the ISA hipcc actually emits:

    python tools/mfma_war_audit.py [file.hip ...]      (default: the MFMA kernels of the default path)

For every kernel: the SMALLEST number of MFMAs issued between an MFMA that reads a register as srcA / srcB and a later load into
that register, looking back over the last LOOKBACK MFMAs in program order (straight-line scan of the kernel text, loop bodies scanned
twice so that the back edge is covered).  distance 0 = the load directly follows the group of MFMAs that read the register.
tests/test_mfma_war_audit.py pins the B-operand distance of the halo convolution kernels at >= 4.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffphycon_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
LOOKBACK = 12
DEFAULT = ["conv3w.hip", "conv3f3c.hip", "igemm6.hip", "igemm_panel.hip", "igemm_tile.hip", "igemm_wide.hip", "stem7x6.hip", "tattn3.hip",
           "lattn3.hip", "wgrad3.hip"]

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    kind = m.group(1)
    if m.group(4) is not None:
        return {(kind, int(m.group(4)))}
    return {(kind, i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}


def kernels(asm):
    """name -> list of instruction lines"""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"):
            if t.startswith("s_endpgm"):
                cur.append(t)
            out[name] = cur
            cur = None
            continue
        if t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
            cur.append(t.split(";")[0].strip())
    return out


def audit(lines, lds_only=True):
    """returns {operand: (min distance in MFMAs, example)} for operand in 'A', 'B'.  lds_only: LDS reads (64-128 cycles: the
    hazardous kind); otherwise global / buffer / scratch loads as well (>= 500 cycles: an MFMA queue never outlasts them)."""
    best = {"A": None, "B": None}
    loads = ("ds_read", "ds_load") if lds_only else ("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")
    recent = []          # (index of mfma in issue order, srcA regs, srcB regs, text)
    n_mfma = 0
    for text in lines + lines:          # (second pass: loop back edges)
        op = text.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            ops = [o.strip() for o in text[len(op):].split(",")]
            if len(ops) >= 3:
                recent.append((n_mfma, regs(ops[1]), regs(ops[2]), text))
                recent = recent[-LOOKBACK:]
            n_mfma += 1
        elif op.startswith(loads):
            dst = regs(text[len(op):].split(",")[0])
            for idx, ra, rb, mtext in recent:
                for which, rset in (("A", ra), ("B", rb)):
                    if dst & rset:
                        d = n_mfma - 1 - idx          # MFMAs issued after the reader, before this load
                        if best[which] is None or d < best[which][0]:
                            best[which] = (d, f"{mtext}  ...  {text}")
    return best, n_mfma // 2


def compile_asm(src):
    p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-o", "-", os.path.join(CSRC, src)], capture_output=True, text=True)
    if p.returncode:
        raise SystemExit(p.stderr[-2000:])
    return p.stdout


def main(files):
    rows = []
    for src in files:
        ks = kernels(compile_asm(src))
        for name, lines in ks.items():
            best, n = audit(lines)
            if n == 0:
                continue
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            short = re.sub(r"\(.*$", "", short).replace("void dpc::", "")
            rows.append((src, short, n, best))
    print(f"{'file':16s} {'kernel':44s} {'MFMAs':>6s}  {'LDS re-load of A':>17s}  {'LDS re-load of B':>17s}     (MFMAs issued in between; - = never within {LOOKBACK})")
    for src, short, n, best in rows:
        fa = "-" if best["A"] is None else str(best["A"][0])
        fb = "-" if best["B"] is None else str(best["B"][0])
        print(f"{src:16s} {short[:44]:44s} {n:6d}  {fa:>17s}  {fb:>17s}")
        if os.environ.get("AUDIT_EXAMPLES") and best["B"] is not None:
            print("      B:", best["B"][1])
    return rows


if __name__ == "__main__":
    main(sys.argv[1:] or DEFAULT)
