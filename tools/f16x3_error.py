#!/usr/bin/env python
"""Error model of the split-operand MFMA schemes on conv-shaped dot products (K = 27*64), against fp64.

  fp32      : plain fp32 products + fp32 accumulation (what the native fp32 MFMA / a CPU does)
  bf16x6    : 3-way bf16 split, 6 partial products, fp32 accumulation
  f16x3     : 2-way fp16 split (RNE, operands pre-scaled by 2^4 / 2^12), 3 partial products, fp32 accumulation
Partial products of 8/11-bit significands are exact in fp32, so each scheme is simulated as fp32 accumulation of the
exact partial products (numpy float32 cumulative sums in a fixed order).
"""
import numpy as np

rng = np.random.default_rng(0)
K, R = 27 * 64, 2000
a = rng.standard_normal((R, K)).astype(np.float32)
a = (a / (1 + np.exp(-a))).astype(np.float32)              # SiLU-shaped activations
w = (rng.standard_normal((R, K)) * 0.024).astype(np.float32)
exact = (a.astype(np.float64) * w.astype(np.float64)).sum(1)
scale = np.sqrt((a.astype(np.float64) ** 2 * w.astype(np.float64) ** 2).sum(1))   # ~ |sum| for random signs


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split_bf16(x):
    p1 = bf16(x); r = x - p1; p2 = bf16(r); p3 = bf16(r - p2)
    return p1, p2, p3


def split_f16(x, s):
    xs = (x * np.float32(s)).astype(np.float32)
    h1 = xs.astype(np.float16).astype(np.float32)
    h2 = (xs - h1).astype(np.float16).astype(np.float32)
    return h1, h2


def acc32(terms):
    """fp32 accumulation over k of the sum of the given exact partial products (each product exact in fp32/fp64)."""
    acc = np.zeros(R, np.float32)
    for k in range(K):
        for t in terms:
            acc = (acc + (t[0][:, k].astype(np.float64) * t[1][:, k].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc.astype(np.float64)


def report(name, got):
    e = np.abs(got - exact) / scale
    print(f"{name:8s} max {e.max():.3e}  rms {np.sqrt((e ** 2).mean()):.3e}   (relative to the rms magnitude of the sum)")


report("fp32", acc32([(a, w)]))
a1, a2, a3 = split_bf16(a); w1, w2, w3 = split_bf16(w)
report("bf16x6", acc32([(a1, w3), (a2, w2), (a3, w1), (a1, w2), (a2, w1), (a1, w1)]))
h1, h2 = split_f16(a, 16.0); g1, g2 = split_f16(w, 4096.0)
report("f16x3", acc32([(h1, g2), (h2, g1), (h1, g1)]) / 65536.0)
# representation-only error (no accumulation rounding): what the split itself loses
rep = ((h1 + h2).astype(np.float64) * (g1 + g2).astype(np.float64) - (h2.astype(np.float64) * g2.astype(np.float64))).sum(1) / 65536.0
print(f"f16x3 representation-only error: max {np.abs(rep - exact).max() / scale.mean():.3e}")
