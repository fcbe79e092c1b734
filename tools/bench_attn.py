"""Micro-benchmark + error of the dense attention core (dpc_attention_core, csrc/attn.hip) at the S64 mid level: one sequence per
frame of 16 x 16 = 256 tokens, 4 heads x 32, micro-batch 32 x 32 frames.   python tools/bench_attn.py [reps]
The kernel runs exact fp32 products on the fp32 MFMA in every arithmetic mode: an f16x3 form of its two products (r03 experiment,
DESIGN.md section 7) was correct (4.6e-7) but SLOWER -- 1184 vs 637 us at 1024 x 256 tokens: six operand splits per 32-key tile
(~220 VALU instructions per wave) cost more than the 13 k matrix-pipe cycles they save."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
heads = 4
for BF, N in ((1024, 256), (64, 1024)):
    g = torch.Generator(device=dev).manual_seed(4)
    qkv = torch.randn(BF, N, 3 * heads * 32, device=dev, generator=g) * 1.5
    out = torch.empty(BF, N, heads * 32, device=dev)
    run = lambda: L.check(L.lib().dpc_attention_core(L.ptr(qkv), L.ptr(out), heads, N, BF, 1, N, 0, 1, None, None, None, L.stream()))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nb = min(BF, 8)
    q, k, v = [z.reshape(nb, N, heads, 32).permute(0, 2, 1, 3).double() for z in qkv[:nb].chunk(3, dim=-1)]
    ref = torch.softmax(torch.einsum("bhid,bhjd->bhij", q * float(torch.tensor(32.0 ** -0.5, dtype=torch.float32)), k), -1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(nb, N, heads * 32)
    err = ((out[:nb].double() - ref).abs().max() / ref.abs().max()).item()
    fl = 4.0 * BF * N * N * 32 * heads
    print(f"{BF} x {N} tokens: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s  err {err:.2e} of the output range  mode {os.environ.get('DPC_ATTN_MODE', 'f16x3')}")
