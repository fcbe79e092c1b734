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
