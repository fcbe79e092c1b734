"""Fills the measured numbers of DESIGN.md section 6 / 7 from an evidence pass (gpurun_out/<tag>/ or profiles/<tag>_*) so that the text and
the committed files cannot drift apart:   python tools/assemble_design.py <tag> <head.md> <sec6.md> <sec7.md> <sec8.md>  > DESIGN.md
(section texts carry @NAME@ placeholders; an unknown placeholder is an error)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def load(name):
    for p in (os.path.join(ROOT, "profiles", f"{tag}_{name}"), os.path.join(ROOT, "gpurun_out", tag, name)):
        if os.path.exists(p):
            return json.loads(open(p).read())
    raise SystemExit(f"no {name} for {tag}")


b, sq, tr = load("bench.json"), load("pmc_sq.json"), load("pmc_traffic.json")
r = b["roofline"]
bd = r["breakdown_ms_per_step"]
cpu = b["cpu_baseline"]
v = {
    "S64MS": f"{b['ms_per_step']:.1f}", "S64V": f"{b['value']:.4f}", "S64F": f"{b['roofline_step']['frac']:.3f}",
    "X6MS": f"{b['ms_per_step_exact']:.1f}", "X6V": f"{b['value_exact']:.4f}",
    "S128MS": f"{b['s128']['ms_per_step']:.1f}", "S128V": f"{b['s128']['value']:.4f}", "S128F": f"{b['s128']['roofline_step']['frac']:.3f}",
    "J128MS": f"{b['j128']['ms_per_step']:.1f}", "J128V": f"{b['j128']['value']:.4f}",
    "TRMS": f"{b['train']['ms_per_step']:.1f}", "TRV": f"{b['train']['value']:.1f}",
    "BUMS": f"{b['burgers']['ms_per_step']:.2f}", "BUV": f"{b['burgers']['value']:.2f}", "BUF": f"{b['burgers']['roofline_step']['frac']:.3f}",
    "CORES": str(cpu["cores"]), "CPUS": f"{cpu['legs'][0]['mean_s_per_step']:.2f}", "CPUV": f"{cpu['value']:.2e}",
    "CVACH": f"{r['achieved']:.1f}", "CVFRAC": f"{r['frac']:.3f}", "CVISS": f"{r['mfma_issue_frac']:.3f}",
    "BUSY64": f"{sq['conv3x6_bn64']['mfma_busy_frac']:.2f}", "BUSY128": f"{sq['conv3x6_bn128']['mfma_busy_frac']:.2f}",
    "CLK64": f"{sq['conv3x6_bn64']['shader_clock_ghz']:.2f}", "CLK128": f"{sq['conv3x6_bn128']['shader_clock_ghz']:.2f}",
    "TRAF": f"{tr['conv3x6_bn64']['hbm_bytes_per_launch'] / 1e6:.0f}",
    "CONVMS": f"{bd['conv3x6_bn64'] + bd['conv3x6_bn128']:.1f}",
    "BREAKDOWN": ", ".join(f"{k} {x:.1f}" for k, x in sorted(bd.items(), key=lambda kv: -kv[1]) if x >= 0.9) + " ms",
}
out = []
for path in sys.argv[2:]:
    text = open(path).read()
    for name in set(re.findall(r"@([A-Z0-9]+)@", text)):
        if name not in v:
            raise SystemExit(f"{path}: unknown placeholder @{name}@")
        text = text.replace(f"@{name}@", v[name])
    out.append(text.rstrip("\n") + "\n")
sys.stdout.write("\n".join(out))
