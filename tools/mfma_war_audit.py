"""Static audit of the hazard of DESIGN.md 6.2: an LDS / global load whose destination registers were an operand of a recently issued
MFMA.  The matrix pipe reads an MFMA's B operand while the instruction executes, not when it issues; a load that is issued right
behind the MFMA and returns early (or an MFMA held up by a foreign wave's MFMAs) overwrites the operand.  The kernels keep a distance
by construction (scheduling barriers, fragment rings); this tool measures it in the ISA hipcc actually emits:

    python tools/mfma_war_audit.py [file.hip ...]      (default: the MFMA kernels of the default path)

For every kernel: the SMALLEST number of MFMAs issued between an MFMA that reads a register as srcA / srcB and a later load into
that register, looking back over the last LOOKBACK MFMAs on EVERY control-flow path that reaches the load (basic blocks + a backward
walk over all predecessors, so loop back edges and out-of-line blocks are followed exactly; r03's version scanned the kernel text twice
in a row, which paired the last MFMAs of a loop-free kernel with its first loads).  distance 0 = the load directly follows the
group of MFMAs that read the register.  An MFMA whose RESULT has been read by a VALU / store instruction before the load (and every
MFMA older than it: one in-order matrix pipe per wave) has completed and is not counted.  tests/test_mfma_war_audit.py pins the B-operand distance of every MFMA kernel of the
default path at >= 4.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffphycon_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang",
         "-packed-fp32-ops", "-S", "--cuda-device-only"]          # (= diffphycon_amd/build.py FLAGS)
LOOKBACK = 12
DEFAULT = ["conv3w4.hip", "conv3w.hip", "conv3f3c.hip", "igemm6.hip", "igemm_panel.hip", "igemm_tile.hip", "igemm_wide.hip", "igemm_img.hip", "stem7x6.hip", "tattn3.hip",
           "lattn3.hip", "wgrad3.hip", "attn.hip", "surr.hip", "train.hip", "unet2d.hip"]

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
        if re.match(r"^\.LBB\w+:", t):
            cur.append(t.split(":")[0] + ":")              # labels stay: loops are found from the backward branches
        elif t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
            cur.append(t.split(";")[0].strip())
    return out


def _blocks(lines):
    """Basic blocks of a kernel's instruction list: [(first, last + 1)], successors and predecessors by block index."""
    leaders = {0}
    label_at = {}
    for i, t in enumerate(lines):
        if t.endswith(":"):
            leaders.add(i)
            label_at[t[:-1]] = i
        elif t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(lines):
            leaders.add(i + 1)
    starts = sorted(leaders)
    blocks = [(st, starts[k + 1] if k + 1 < len(starts) else len(lines)) for k, st in enumerate(starts)]
    index_of = {st: k for k, (st, _) in enumerate(blocks)}
    succ = [[] for _ in blocks]
    for k, (st, en) in enumerate(blocks):
        last = lines[en - 1] if en > st else ""
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", last)
        if m and m.group(1) in label_at:
            succ[k].append(index_of[label_at[m.group(1)]])
        if not last.startswith(("s_branch", "s_endpgm", "s_setpc")) and k + 1 < len(blocks):
            succ[k].append(k + 1)
    pred = [[] for _ in blocks]
    for k, ss in enumerate(succ):
        for t in ss:
            pred[t].append(k)
    return blocks, pred


def audit(lines, lds_only=True):
    """returns {operand: (min distance in MFMAs, example)} for operand in 'A', 'B', and the number of MFMAs of the kernel.
    For every load, the control-flow graph is walked BACKWARDS from it (all predecessors, loops included) until LOOKBACK MFMAs have been
    passed on a path; an MFMA on the way that reads one of the load's destination registers gives a distance = MFMAs between them.
    lds_only: LDS reads (64-128 cycles: the hazardous kind); otherwise global / buffer / scratch loads as well."""
    best = {"A": None, "B": None}
    loads = ("ds_read", "ds_load") if lds_only else ("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")
    blocks, pred = _blocks(lines)
    parsed = {}
    touched = {}          # non-MFMA instruction -> registers it names (reads, over-approximated by all its operands)
    n_mfma = 0
    for i, text in enumerate(lines):
        if text.endswith(":"):
            continue
        op = text.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            ops = [o.strip() for o in text[len(op):].split(",")]
            if len(ops) >= 3:
                parsed[i] = (regs(ops[1]), regs(ops[2]), regs(ops[0]))
            n_mfma += 1
        elif not op.startswith(("s_", "ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
            r = set()
            for tok in text[len(op):].split(","):
                r |= regs(tok)
            if r:
                touched[i] = r
    block_of = {}
    for k, (st, en) in enumerate(blocks):
        for i in range(st, en):
            block_of[i] = k
    for i, text in enumerate(lines):
        if text.endswith(":") or not text.split()[0].startswith(loads):
            continue
        op = text.split()[0]
        dst = regs(text[len(op):].split(",")[0])
        if not dst:
            continue
        # `used` = registers named by the VALU / store / LDS-write instructions between the walk position and the load.  An MFMA whose
        # RESULT one of them reads has completed before the load issues (RAW on the accumulator), and so has every older MFMA of the wave
        # (one matrix pipe, in order): the walk stops there -- a finished MFMA no longer reads its operands.
        seen = {}
        stack = [(block_of[i], i - 1, 0, frozenset())]          # (block, index to walk back from, MFMAs passed, used)
        while stack:
            k, j, cnt, used0 = stack.pop()
            used = set(used0)
            st = blocks[k][0]
            done = False
            while j >= st and cnt < LOOKBACK:
                if j in parsed:
                    ra, rb, rd = parsed[j]
                    if rd & used:
                        done = True
                        break
                    for which, rset in (("A", ra), ("B", rb)):
                        if dst & rset and (best[which] is None or cnt < best[which][0]):
                            best[which] = (cnt, f"{lines[j]}  ...  {text}")
                    cnt += 1
                elif j in touched:
                    used |= touched[j]
                j -= 1
            if done or cnt >= LOOKBACK:
                continue
            fu = frozenset(used)
            for q in pred[k]:
                old = seen.get((q, cnt))
                if old is not None and fu >= old:               # already walked with no more knowledge of completed MFMAs
                    continue
                nu = fu if old is None else (fu & old)
                seen[(q, cnt)] = nu
                stack.append((q, blocks[q][1] - 1, cnt, nu))
    return best, n_mfma


def compile_asm(src):
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from diffphycon_amd.build import SRC_FLAGS                     # per-file additions of the product build (a later flag wins)
    p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *SRC_FLAGS.get(src, []), "-I", CSRC, "-o", "-", os.path.join(CSRC, src)],
                       capture_output=True, text=True)
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
