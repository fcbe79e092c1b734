"""The minimal-filtering identity behind csrc/conv3w.hip (Winograd F(2,3) along the frame axis of the 3x3x3 convolution,
video_diffusion_pytorch_conv3d.py:189-204), restated in NumPy exactly as the kernel's loader, weight pack and epilogue apply it:
    V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3),  U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2),  m = U * V,
    out(2p) = m0 + m1 + m2,  out(2p + 1) = m1 - m2 - m3        for input frames d0..d3 = 2p - 1 .. 2p + 2 (zero padded)."""
import numpy as np


def conv_frames_direct(x, g):
    """cross-correlation along axis 0 with zero padding 1 (what nn.Conv3d does along the frame axis)"""
    F = x.shape[0]
    xp = np.concatenate([np.zeros_like(x[:1]), x, np.zeros_like(x[:1])])
    return np.stack([sum(g[t] * xp[f + t] for t in range(3)) for f in range(F)])


def conv_frames_winograd(x, g):
    F = x.shape[0]
    Fp = (F + 1) // 2 * 2
    xp = np.concatenate([np.zeros_like(x[:1]), x, np.zeros((Fp - F + 2,) + x.shape[1:], x.dtype)])     # frame f at index f + 1
    U = (g[0], (g[0] + g[1] + g[2]) / 2, (g[0] - g[1] + g[2]) / 2, g[2])
    out = np.zeros((Fp,) + x.shape[1:], x.dtype)
    for p in range(Fp // 2):
        d0, d1, d2, d3 = (xp[2 * p + i] for i in range(4))
        V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
        m = [U[k] * V[k] for k in range(4)]
        out[2 * p] = m[0] + m[1] + m[2]
        out[2 * p + 1] = m[1] - m[2] - m[3]
    return out[:F]


def test_f23_over_frames_equals_the_direct_convolution():
    rng = np.random.default_rng(0)
    for F in (2, 4, 5, 17, 20):
        x = rng.standard_normal((F, 3, 4))
        g = rng.standard_normal(3)
        np.testing.assert_allclose(conv_frames_winograd(x, g), conv_frames_direct(x, g), rtol=0, atol=1e-12)


def test_product_count():
    """36 instead of 54 tap products per output-frame pair and (h, w) tap set: 4 components x 9 taps vs 2 frames x 27 taps"""
    assert 4 * 9 * 3 == 2 * 27 * 2


# ---- F(4,3) over frames, interpolation points (0, 1, -1, 1/2, -2, inf): csrc/conv3w4.hip (r06, the default 3x3x3 form)
def conv_frames_winograd_f43(x, g):
    """restated exactly as the kernel applies it: loader B^T (with its shared sub-expressions), weight pack G, epilogue A^T;
    input frames d0..d5 = f0 - 1 .. f0 + 4 (zero padded), four output frames per group"""
    F = x.shape[0]
    Fp = (F + 3) // 4 * 4
    xp = np.concatenate([np.zeros_like(x[:1]), x, np.zeros((Fp - F + 4,) + x.shape[1:], x.dtype)])     # frame f at index f + 1
    U = (g[0], ((g[0] + g[2]) + g[1]) / 3, (g[1] - (g[0] + g[2])) / 3, -(16 * g[0] + 8 * g[1] + 4 * g[2]) / 15,
         (4 * g[2] - 2 * g[1] + g[0]) / 15, g[2])
    out = np.zeros((Fp,) + x.shape[1:], x.dtype)
    for p in range(Fp // 4):
        d = [xp[4 * p + i] for i in range(6)]
        a, b = d[4] - d[2], d[3] - d[1]
        V = (1.5 * b + (-2 * d[2] + (d[0] + d[4])),
             2.5 * d[3] + (0.5 * d[2] + (d[4] - d[1])),
             0.5 * d[3] + (-2.5 * d[2] + (d[4] + d[1])),
             2 * b + a,
             -0.5 * b + a,
             1.5 * a + (-2 * d[3] + (d[5] + d[1])))
        m = [U[k] * V[k] for k in range(6)]
        s, dd = m[1] + m[2], m[1] - m[2]
        out[4 * p] = ((s + m[3]) + m[4]) + m[0]
        out[4 * p + 1] = -2 * m[4] + (0.5 * m[3] + dd)
        out[4 * p + 2] = 4 * m[4] + (0.25 * m[3] + s)
        out[4 * p + 3] = (-8 * m[4] + (0.125 * m[3] + dd)) + m[5]
    return out[:F]


def test_f43_over_frames_equals_the_direct_convolution():
    rng = np.random.default_rng(1)
    for F in (4, 5, 8, 17, 18, 19, 20, 32):
        x = rng.standard_normal((F, 3, 4))
        g = rng.standard_normal(3)
        np.testing.assert_allclose(conv_frames_winograd_f43(x, g), conv_frames_direct(x, g), rtol=0, atol=1e-12)


def test_f43_matches_the_toom_cook_construction():
    """the constants written into the kernel are the Toom-Cook matrices of the point set (tools/winograd_f43_error.py builds them
    from the points alone): B^T rows, G rows and A^T rows"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from winograd_f43_error import toom_cook
    AT, G, BT = toom_cook([0, 1, -1, 0.5, -2])
    np.testing.assert_allclose(BT, [[1, -1.5, -2, 1.5, 1, 0], [0, -1, .5, 2.5, 1, 0], [0, 1, -2.5, .5, 1, 0], [0, -2, -1, 2, 1, 0],
                                    [0, .5, -1, -.5, 1, 0], [0, 1, -1.5, -2, 1.5, 1]], atol=1e-15)
    np.testing.assert_allclose(G, [[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                                   [1 / 15, -2 / 15, 4 / 15], [0, 0, 1]], atol=1e-15)
    np.testing.assert_allclose(AT, [[1, 1, 1, 1, 1, 0], [0, 1, -1, .5, -2, 0], [0, 1, 1, .25, 4, 0], [0, 1, -1, .125, -8, 1]], atol=1e-15)


def test_f43_product_count():
    """54 instead of 72 (F(2,3)) or 108 (direct) tap products per four output frames and (h, w) tap set"""
    assert 6 * 9 == 54 and 2 * 4 * 9 == 72 and 4 * 27 == 108
