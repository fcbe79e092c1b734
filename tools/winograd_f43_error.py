#!/usr/bin/env python
"""Gate for a one-axis Winograd F(4,3) over frames in csrc/conv3w.hip: the error of the f16x3 product
scheme (22-bit split operands, 3 partial products, fp32 accumulation) under the F(4,3) transforms, next to the direct form and the
shipped F(2,3), on conv-shaped reductions of the eight S64 layer shapes (K = Cin x 9 plane taps per component), against fp64.

What the kernel would do is restated step by step: data transform B^T d in fp32 (after the fused activation) BEFORE the fp16
split, weight transform G g in fp32 before the split, per-chunk (16 channels = one MFMA k-step) fp32 accumulation of the three
partial products (each exact in fp32), output transform A^T m in fp32.  Reported as the tests report it: max |err| / (max - min)
of the fp64 output (tests/test_gpu_ops.py relerr; bar 3e-6 per convolution, SURVEY 8d per block 1e-5), plus the fp16-window
headroom: the largest pre-scale (power of two) for which |V| of an activation of magnitude 4094 (the range every f16x3 kernel
guarantees) still converts to a finite fp16.

Point sets: `std` = Lavin's (0, +-1, +-2, inf); `half` = (0, +-1, +-1/2, inf) (the transposed conditioning: small |B^T|, large |A^T|);
`mix` = (0, +-1, 1/2, -2, inf) (Barabasz et al.'s best F(4,3) set).  Transforms are built by the Toom-Cook construction below and
checked against the direct convolution in fp64 before use.
"""
import itertools
import sys

import numpy as np


def toom_cook(points, m=4, r=3):
    """F(m, r) matrices (AT m x n, G n x r, BT n x n) for n = m + r - 1 points (the last one = infinity), cross-correlation form:
    y = AT [(G g) * (BT d)].  Construction: Lagrange interpolation at the finite points (Toom-Cook), as in Lavin & Gray."""
    from fractions import Fraction as Fr
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1
    # A^T: rows = powers of the points (m x n), last column = unit for the infinite point
    AT = [[pts[j] ** i for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    # G: rows = points' powers scaled by the Lagrange denominators
    def denom(j):
        d = Fr(1)
        for k in range(n - 1):
            if k != j:
                d *= pts[j] - pts[k]
        return d
    G = [[pts[j] ** i / denom(j) for i in range(r)] for j in range(n - 1)] + [[Fr(1 if i == r - 1 else 0) for i in range(r)]]
    # B^T: rows = coefficients of prod_{k != j} (x - p_k) for the finite points, last row = coefficients of prod_k (x - p_k)
    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    BT = []
    for j in range(n - 1):
        pol = [Fr(1)]
        for k in range(n - 1):
            if k != j:
                pol = poly_mul(pol, [-pts[k], Fr(1)])
        BT.append(pol + [Fr(0)] * (n - len(pol)))
    pol = [Fr(1)]
    for k in range(n - 1):
        pol = poly_mul(pol, [-pts[k], Fr(1)])
    BT.append(pol)
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), f(BT)


def check(AT, G, BT):
    rng = np.random.default_rng(1)
    n = BT.shape[0]
    m = AT.shape[0]
    d = rng.standard_normal(n)
    g = rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[t] * d[i + t] for t in range(3)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-12), (y, ref)


def f16split(x, scale):
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    with np.errstate(over="ignore"):
        h1 = xs.astype(np.float16).astype(np.float32)
        h2 = (xs - h1).astype(np.float16).astype(np.float32)
    return h1, h2


def dot3(a, w, sa, sw):
    """f16x3 reduction over the last axis of a [R, J] against w [J]: chunks of 16 = one MFMA k-step, three MFMAs per chunk in the
    kernel's order (small terms first), each MFMA adds its exact 16-term sum to the fp32 accumulator with one rounding."""
    a1, a2 = f16split(a, sa)
    w1, w2 = f16split(w, sw)
    R, J = a.shape
    acc = np.zeros(R, np.float32)
    for c in range(0, J, 16):
        s = slice(c, c + 16)
        for (x, y) in ((a1[:, s], w2[s]), (a2[:, s], w1[s]), (a1[:, s], w1[s])):
            acc = (acc.astype(np.float64) + (x.astype(np.float64) * y.astype(np.float64)).sum(1)).astype(np.float32)
    return acc / np.float32(sa * sw)


def f32mat(M, xs):
    """rows of M applied to the list xs in fp32, left to right (what a chain of v_fma / v_add does)"""
    out = []
    for row in M:
        acc = None
        for c, x in zip(row, xs):
            if c == 0:
                continue
            t = (np.float32(c) * x).astype(np.float32)
            acc = t if acc is None else (acc + t).astype(np.float32)
        out.append(acc if acc is not None else np.zeros_like(xs[0]))
    return out


def run(Cin, R, seed, act):
    rng = np.random.default_rng(seed)
    J = 9 * Cin
    nfr = 6
    d = rng.standard_normal((nfr, R, J)).astype(np.float32)
    if act == "silu":
        d = (d / (1 + np.exp(-d))).astype(np.float32)
    g = (rng.standard_normal((3, J)) / np.sqrt(27 * Cin)).astype(np.float32)
    d64, g64 = d.astype(np.float64), g.astype(np.float64)
    ref = np.stack([sum((d64[i + t] * g64[t]).sum(1) for t in range(3)) for i in range(4)])      # [4, R]
    rngv = ref.max() - ref.min()
    res = {}
    # direct: 3 frame taps x J
    out = np.stack([dot3(np.concatenate([d[i + t] for t in range(3)], 1), np.concatenate([g[t] for t in range(3)]), 16.0, 4096.0)
                    for i in range(4)])
    res["direct"] = np.abs(out - ref).max() / rngv
    # F(2,3), as shipped (SAW = 8, SW = 4096)
    AT2, G2, BT2 = toom_cook([0, 1, -1], m=2)
    out = []
    for p in range(2):
        V = f32mat(BT2, [d[2 * p + i] for i in range(4)])
        U = f32mat(G2, [g[0], g[1], g[2]])
        m = [dot3(V[k], U[k], 8.0, 4096.0) for k in range(4)]
        out += f32mat(AT2, m)
    res["F(2,3)"] = np.abs(np.stack(out) - ref).max() / rngv
    for name, pts in (("F(4,3) std", [0, 1, -1, 2, -2]), ("F(4,3) half", [0, 1, -1, 0.5, -0.5]), ("F(4,3) mix", [0, 1, -1, 0.5, -2])):
        AT, G, BT = toom_cook(pts)
        # move the Lagrange denominators' magnitude so that G's rows are O(1/2) and B^T keeps small integers where possible: scale row j of
        # G by s_j and row j of B^T by 1 / s_j is a free choice; keep the construction's own split (denominators in G) -- it is Lavin's
        check(AT, G, BT)
        V = f32mat(BT, [d[i] for i in range(6)])
        U = f32mat(G, [g[0], g[1], g[2]])
        bsum = np.abs(BT).sum(1).max()
        sa = 2.0 ** np.floor(np.log2(65504.0 / (bsum * 4094.0)))
        m = [dot3(V[k], U[k], sa, 4096.0) for k in range(6)]
        out = np.stack(f32mat(AT, m))
        e = np.abs(out - ref) / rngv
        res[name] = e.max()
        res[name + " per-output"] = e.max(1)
        res[name + " prescale"] = sa
        res[name + " |BT| row sums"] = np.abs(BT).sum(1)
    return res


def main():
    shapes = [64, 128, 192, 256, 384, 512]          # Cin of the S64 3x3x3 layers (incl. the concatenated up-path inputs)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    print(f"{R} output points per shape and output frame; error = max |err| / range of the fp64 output")
    for act in ("silu", "plain"):
        print(f"\n== activations: {act}")
        for Cin in shapes:
            r = run(Cin, R, Cin, act)
            line = f"Cin {Cin:4d}: direct {r['direct']:.2e}  F(2,3) {r['F(2,3)']:.2e}"
            for n in ("F(4,3) std", "F(4,3) half", "F(4,3) mix"):
                line += f"  {n} {r[n]:.2e}"
            print(line)
            if Cin == shapes[0] and act == "silu":
                for n in ("F(4,3) std", "F(4,3) half", "F(4,3) mix"):
                    print(f"      {n}: per output frame {np.array2string(r[n + ' per-output'], precision=2)}  pre-scale 2^{int(np.log2(r[n + ' prescale']))}"
                          f"  |B^T| row sums {r[n + ' |BT| row sums']}")


if __name__ == "__main__":
    main()
