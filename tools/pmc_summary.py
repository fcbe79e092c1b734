#!/usr/bin/env python
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; they do not fit one pass, see
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") into profiles/pmc_traffic.json:

    {profile class: {"hbm_bytes_per_launch": ..., "fetch_kb_raw": ..., "write_kb_raw": ..., "launches": ...}}

Corrections per that guide's HBM section: FETCH_SIZE is reported in KB and, on gfx950, at exactly half of the bytes
of a wide coalesced read -> x2; WRITE_SIZE (KB) is taken as is (uncalibrated, stated in the JSON).

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

# kernel symbol -> bench.py / ProfScope class name
CLASSES = [
    (r"conv3(x6|f3[bc]?)_kernel<128", "conv3x6_bn128"), (r"conv3(x6|f3[bc]?)_kernel<64", "conv3x6_bn64"),
    (r"conv3w4?_kernel<(true|false), 128>", "conv3x6_bn128"), (r"conv3w4?_kernel<(true|false), 64>", "conv3x6_bn64"),
    (r"conv3h_kernel<128", "conv3h_bn128"), (r"conv3h_kernel<64", "conv3h_bn64"),
    (r"igemm3p_kernel<\d+, 1, 1, 2, 2", "igemm_bn64"), (r"igemm3p_kernel<", "igemm_bn128"),       # (row panels: N = 32 NT WN)
    (r"igemm3w_kernel<(true|false), 64>", "igemm_bn64"), (r"igemm3w_kernel<", "igemm_bn128"),
    (r"igemm3t_kernel<2,", "igemm_bn64"), (r"igemm3t_kernel<", "igemm_bn128"),                      # (pixel tiles: 64 -> 64 / 128 -> 128)
    (r"igemm3i_kernel<", "igemm_bn128"),                                                           # (image tiles: N % 128 == 0)
    (r"igemm[63]?_kernel<128", "igemm_bn128"), (r"igemm[63]?_kernel<64|conv1x1_rows_kernel", "igemm_bn64"),
    (r"stem7x6_kernel|stem7p_kernel|stem_kernel", "stem_gather"), (r"tattn(_fused|6|3w?)_kernel", "temporal_attention_fused"),
    (r"lattn(3|6?_(ctx|out))_kernel", "linear_attention_fused"), (r"gn_(partial|finalize|finalize_fused|apply)_kernel", "groupnorm_silu"),
    (r"ln_stats_kernel", "ln_stats"), (r"ddpm_update_smoke_kernel", "ddpm_update"),
    (r"philox_normal_kernel", "philox_normal"), (r"attention_kernel", "attention_core"),
    (r"wgrad3_kernel", "conv3_wgrad_f16x3"), (r"wgrad_kernel", "conv_wgrad"), (r"tattn_bwd", "attention_bwd"),
    (r"unet2d|conv2d", "unet2d"), (r"cg_|pressure|advect|smoke_", "smoke_rollout"), (r"burgers", "burgers"),
]


# profile class -> the kernel source file(s) whose code its default-mode launches run: the traffic numbers are stamped with a hash
# of these files and bench.py reports `traffic: null` when the stamp no longer matches the tree (a changed kernel = stale counters)
SOURCES = {
    "conv3x6_bn64": ["conv3w4.hip", "conv3w.hip", "conv3f3c.hip"], "conv3x6_bn128": ["conv3w4.hip", "conv3w.hip", "conv3f3c.hip"], "igemm_bn64": ["igemm6.hip", "igemm_panel.hip", "igemm_wide.hip", "igemm_tile.hip", "igemm_img.hip", "igemm_epilogue.h"],
    "igemm_bn128": ["igemm6.hip", "igemm_panel.hip", "igemm_wide.hip", "igemm_tile.hip", "igemm_img.hip", "igemm_epilogue.h"], "stem_gather": ["stem7x6.hip"], "temporal_attention_fused": ["tattn3.hip"],
    "linear_attention_fused": ["lattn3.hip"], "groupnorm_silu": ["norm.hip"], "ln_stats": ["norm.hip"], "attention_core": ["attn.hip"],
    "ddpm_update": ["update.hip"], "conv3_wgrad_f16x3": ["wgrad3.hip"], "conv_wgrad": ["train.hip"], "attention_bwd": ["train.hip"],
}
CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diffphycon_amd", "csrc")


def source_stamp(cls):
    """sha256[:16] over the kernel source files of a profile class (None for classes without a mapping)."""
    import hashlib
    files = SOURCES.get(cls)
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        try:
            h.update(open(os.path.join(CSRC, f), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def classify(name):
    for pat, cls in CLASSES:
        if re.search(pat, name):
            return cls
    return None


def read(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            cls = classify(row["Kernel_Name"])
            if cls is None:
                continue
            a = acc[cls]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def sq_summary(path, dst):
    """SQ pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) -> per-class MFMA-pipe utilisation.
    GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over all 256 CUs x 4 SIMDs."""
    acc, n, dur = defaultdict(lambda: defaultdict(float)), defaultdict(int), defaultdict(float)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            cls = classify(row["Kernel_Name"])
            if cls is None:
                continue
            acc[cls][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[cls] += 1
                dur[cls] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    out = {}
    for cls, v in sorted(acc.items()):
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        out[cls] = {"launches": n[cls], "avg_us": dur[cls] / n[cls] / 1e3,
                    "shader_clock_ghz": cyc / max(dur[cls], 1),
                    "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4) if cyc else 0.0,
                    "kernel_source_sha16": source_stamp(cls)}
        print(f"{cls:28s} launches {n[cls]:6d} avg {out[cls]['avg_us']:9.1f} us  clock {out[cls]['shader_clock_ghz']:.2f} GHz"
              f"  MFMA busy {100 * out[cls]['mfma_busy_frac']:5.1f} %")
    out["_note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; mfma_busy_frac = MFMA busy cycles / "
                    "(GRBM_GUI_ACTIVE/8 XCDs x 256 CUs x 4 SIMDs); clock = active cycles per XCD / kernel duration")
    json.dump(out, open(dst, "w"), indent=1)


def main():
    if sys.argv[1] == "sq":
        return sq_summary(sys.argv[2], sys.argv[3])
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    out = {}
    for cls in sorted(set(fetch) | set(write)):
        fk, fn = fetch.get(cls, [0.0, 0])
        wk, wn = write.get(cls, [0.0, 0])
        n = max(fn, wn, 1)
        out[cls] = {"launches": n, "fetch_kb_raw_per_launch": fk / max(fn, 1), "write_kb_raw_per_launch": wk / max(wn, 1),
                    "hbm_bytes_per_launch": (2.0 * fk / max(fn, 1) + wk / max(wn, 1)) * 1024.0,
                    "kernel_source_sha16": source_stamp(cls)}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950 correction, "
                    "MI355X_MICROARCH.md HBM section), KB -> bytes; WRITE_SIZE uncalibrated; averages over all launches "
                    "of the class in the profiled bench run")
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        if not k.startswith("_"):
            print(f"{k:28s} launches {v['launches']:6d}  HBM {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
