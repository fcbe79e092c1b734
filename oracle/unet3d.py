"""ORACLE (test infrastructure, not product): CPU restatement of the space-time U-Net denoiser.

Follows /root/reference/model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py
(`Unet3D_with_Conv3D`, :356-552) as a *functional* fp32 torch-CPU program over a plain
state-dict with the reference's parameter names.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Parity status: pinned against the reference itself, imported in the build container through
tools/refshim.py, on the fixtures in tests/golden/unet3d_*.npz (tests/test_oracle_unet3d.py).
The rotary embedding is third-party (`rotary-embedding-torch==0.8.4`, environment.yaml:88),
not vendored in the reference and not installable offline: its published algorithm is restated
in `rotary()` below and that one function is "parity unpinned" (see DESIGN.md).
"""
import math

import torch
import torch.nn.functional as F


class Unet3DConfig:
    """Hyper-parameters of Unet3D_with_Conv3D.__init__ (…conv3d.py:357-372)."""

    def __init__(self, dim=64, dim_mults=(1, 2, 4), channels=6, out_dim=None, attn_heads=4,
                 attn_dim_head=32, init_kernel_size=7, resnet_groups=8):
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.channels = channels
        self.out_dim = channels if out_dim is None else out_dim
        self.heads = attn_heads
        self.dim_head = attn_dim_head
        self.init_kernel_size = init_kernel_size
        self.groups = resnet_groups

    @property
    def dims(self):
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self):
        d = self.dims
        return list(zip(d[:-1], d[1:]))


# ----------------------------------------------------------------------------- positional terms

def relative_position_bucket(n_frames, num_buckets=32, max_distance=32):
    """T5 bucket table [i, j] (int64), …conv3d.py:86-104 with max_distance=32 (:384).

    Integer result, but it goes through an fp32 log exactly as the reference does; the table
    is pinned bit-exact by tests/golden/relpos_bucket.npz.
    """
    q = torch.arange(n_frames, dtype=torch.long)
    rel = q[None, :] - q[:, None]                      # k_pos - q_pos   (:109)
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    ).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def rel_pos_bias(sd, n_frames):
    """[heads, i, j] bias = Embedding(bucket) (…conv3d.py:106-112)."""
    w = sd["time_rel_pos_bias.relative_attention_bias.weight"]   # [32, heads]
    return w[relative_position_bucket(n_frames)].permute(2, 0, 1).contiguous()


def rotary(t, theta=10000.0):
    """Interleaved-pair RoPE over the sequence axis (-2) on all `d` dims of a head.

    Restated from rotary-embedding-torch 0.8.4 `RotaryEmbedding(dim).rotate_queries_or_keys`:
    freqs_i = theta^(-2i/d), angle[n, 2i] = angle[n, 2i+1] = n*freqs_i,
    out = t*cos + rotate_half(t)*sin with (x0,x1) -> (-x1,x0).   PARITY UNPINNED.
    """
    n, d = t.shape[-2], t.shape[-1]
    freqs = 1.0 / (theta ** (torch.arange(0, d, 2)[: d // 2].float() / d))
    ang = torch.einsum("i,j->ij", torch.arange(n).float(), freqs)
    ang = ang.repeat_interleave(2, dim=-1)
    x = t.reshape(*t.shape[:-1], d // 2, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def sinusoidal_pos_emb(time, dim):
    """…conv3d.py:139-151 (int64 time × fp32 frequencies)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = time[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


# ----------------------------------------------------------------------------- blocks

def channel_layernorm(x, gamma, eps=1e-5):
    """LayerNorm over dim=1, biased variance, gamma only (…conv3d.py:165-174)."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * gamma


def block(sd, p, x, groups, scale_shift=None):
    """Conv3d 3x3x3 -> GroupNorm -> (scale+1, shift) -> SiLU (…conv3d.py:189-204)."""
    x = F.conv3d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1)
    x = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(sd, p, x, t_emb, groups):
    """…conv3d.py:206-230."""
    scale_shift = None
    if (p + ".mlp.1.weight") in sd:
        te = F.linear(F.silu(t_emb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        te = te[:, :, None, None, None]
        scale_shift = te.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, groups, scale_shift)
    h = block(sd, p + ".block2", h, groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv3d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def spatial_linear_attention(sd, p, x, heads):
    """Residual(PreNorm(SpatialLinearAttention)) (…conv3d.py:232-257, 441)."""
    b, c, f, h, w = x.shape
    y = channel_layernorm(x, sd[p + ".norm.gamma"])
    y = y.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    qkv = F.conv2d(y, sd[p + ".fn.to_qkv.weight"])
    q, k, v = [z.reshape(b * f, heads, -1, h * w) for z in qkv.chunk(3, dim=1)]
    dh = q.shape[2]
    q = q.softmax(dim=-2)
    k = k.softmax(dim=-1)
    q = q * dh ** -0.5
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q)
    out = out.reshape(b * f, heads * dh, h, w)
    out = F.conv2d(out, sd[p + ".fn.to_out.weight"], sd[p + ".fn.to_out.bias"])
    out = out.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)
    return out + x


def attention(sd, p, x, heads, pos_bias=None, use_rotary=False):
    """`Attention.forward` on x[..., n, c] (…conv3d.py:293-352) with the focus-present branches inert."""
    qkv = F.linear(x, sd[p + ".to_qkv.weight"]).chunk(3, dim=-1)
    n = x.shape[-2]
    q, k, v = [z.reshape(*z.shape[:-1], heads, -1).transpose(-2, -3) for z in qkv]   # ... h n d
    q = q * q.shape[-1] ** -0.5
    if use_rotary:
        q = rotary(q)
        k = rotary(k)
    sim = torch.einsum("...hid,...hjd->...hij", q, k)
    if pos_bias is not None:
        sim = sim + pos_bias
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("...hij,...hjd->...hid", attn, v)
    out = out.transpose(-2, -3).reshape(*x.shape[:-2], n, -1)
    return F.linear(out, sd[p + ".to_out.weight"])


def temporal_attention(sd, p, x, heads, pos_bias):
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention+rotary))) (:382,394,442)."""
    b, c, f, h, w = x.shape
    y = channel_layernorm(x, sd[p + ".norm.gamma"])
    y = y.permute(0, 3, 4, 2, 1).reshape(b, h * w, f, c)
    y = attention(sd, p + ".fn.fn", y, heads, pos_bias=pos_bias, use_rotary=True)
    y = y.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
    return y + x


def mid_spatial_attention(sd, p, x, heads):
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b f (h w) c', Attention))) (:449-451)."""
    b, c, f, h, w = x.shape
    y = channel_layernorm(x, sd[p + ".norm.gamma"])
    y = y.permute(0, 2, 3, 4, 1).reshape(b, f, h * w, c)
    y = attention(sd, p + ".fn.fn", y, heads)
    y = y.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    return y + x


# ----------------------------------------------------------------------------- the model

def unet3d_forward(sd, cfg, x, time, taps=None):
    """`Unet3D_with_Conv3D.forward(x[B,F,C,H,W], time[B]) -> [B,F,C_out,H,W]` (…conv3d.py:486-552).

    `taps`: optional dict that receives named intermediate activations (channels-first) for
    per-block kernel tests.
    """
    def tap(name, v):
        if taps is not None:
            taps[name] = v.detach().clone()

    g, heads = cfg.groups, cfg.heads
    x = x.permute(0, 2, 1, 3, 4)
    bias = rel_pos_bias(sd, x.shape[2])
    pad = cfg.init_kernel_size // 2
    x = F.conv3d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=pad)
    tap("init_conv", x)
    x = temporal_attention(sd, "init_temporal_attn.fn", x, heads, bias)
    tap("init_temporal_attn", x)
    r = x.clone()

    t = sinusoidal_pos_emb(time, cfg.dim)
    t = F.linear(t, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    t = F.gelu(t)
    t = F.linear(t, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    tap("time_mlp", t)

    hs = []
    n_res = len(cfg.in_out)
    for i in range(n_res):
        p = f"downs.{i}"
        x = resnet_block(sd, p + ".0", x, t, g)
        tap(p + ".0", x)
        x = resnet_block(sd, p + ".1", x, t, g)
        tap(p + ".1", x)
        x = spatial_linear_attention(sd, p + ".2.fn", x, heads)
        tap(p + ".2", x)
        x = temporal_attention(sd, p + ".3.fn", x, heads, bias)
        tap(p + ".3", x)
        hs.append(x)
        if i < n_res - 1:
            x = F.conv3d(x, sd[p + ".4.weight"], sd[p + ".4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
            tap(p + ".4", x)

    x = resnet_block(sd, "mid_block1", x, t, g)
    tap("mid_block1", x)
    x = mid_spatial_attention(sd, "mid_spatial_attn.fn", x, heads)
    tap("mid_spatial_attn", x)
    x = temporal_attention(sd, "mid_temporal_attn.fn", x, heads, bias)
    tap("mid_temporal_attn", x)
    x = resnet_block(sd, "mid_block2", x, t, g)
    tap("mid_block2", x)

    for i in range(n_res):
        p = f"ups.{i}"
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(sd, p + ".0", x, t, g)
        tap(p + ".0", x)
        x = resnet_block(sd, p + ".1", x, t, g)
        tap(p + ".1", x)
        x = spatial_linear_attention(sd, p + ".2.fn", x, heads)
        tap(p + ".2", x)
        x = temporal_attention(sd, p + ".3.fn", x, heads, bias)
        tap(p + ".3", x)
        if i < n_res - 1:
            x = F.conv_transpose3d(x, sd[p + ".4.weight"], sd[p + ".4.bias"], stride=(1, 2, 2), padding=(0, 1, 1))
            tap(p + ".4", x)

    x = torch.cat((x, r), dim=1)
    x = resnet_block(sd, "final_conv.0", x, None, g)
    tap("final_conv.0", x)
    x = F.conv3d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.permute(0, 2, 1, 3, 4)


# ----------------------------------------------------------------------------- synthetic weights

def param_shapes(cfg):
    """Ordered (name, shape) list in the reference's registration order (…conv3d.py:380-471)."""
    out = []
    hid = cfg.heads * cfg.dim_head
    tdim = cfg.dim * 4
    k = cfg.init_kernel_size

    def tattn(p, d):
        out.append((p + ".fn.fn.fn.to_qkv.weight", (hid * 3, d)))
        out.append((p + ".fn.fn.fn.to_out.weight", (d, hid)))
        out.append((p + ".fn.norm.gamma", (1, d, 1, 1, 1)))

    def sattn(p, d):
        out.append((p + ".fn.fn.to_qkv.weight", (hid * 3, d, 1, 1)))
        out.append((p + ".fn.fn.to_out.weight", (d, hid, 1, 1)))
        out.append((p + ".fn.fn.to_out.bias", (d,)))
        out.append((p + ".fn.norm.gamma", (1, d, 1, 1, 1)))

    def res(p, di, do, temb=True):
        if temb:
            out.append((p + ".mlp.1.weight", (do * 2, tdim)))
            out.append((p + ".mlp.1.bias", (do * 2,)))
        for b, ci in ((".block1", di), (".block2", do)):
            out.append((p + b + ".proj.weight", (do, ci, 3, 3, 3)))
            out.append((p + b + ".proj.bias", (do,)))
            out.append((p + b + ".norm.weight", (do,)))
            out.append((p + b + ".norm.bias", (do,)))
        if di != do:
            out.append((p + ".res_conv.weight", (do, di, 1, 1, 1)))
            out.append((p + ".res_conv.bias", (do,)))

    out.append(("time_rel_pos_bias.relative_attention_bias.weight", (32, cfg.heads)))
    out.append(("init_conv.weight", (cfg.dim, cfg.channels, k, k, k)))
    out.append(("init_conv.bias", (cfg.dim,)))
    tattn("init_temporal_attn", cfg.dim)
    out.append(("time_mlp.1.weight", (tdim, cfg.dim)))
    out.append(("time_mlp.1.bias", (tdim,)))
    out.append(("time_mlp.3.weight", (tdim, tdim)))
    out.append(("time_mlp.3.bias", (tdim,)))
    n_res = len(cfg.in_out)
    for i, (di, do) in enumerate(cfg.in_out):
        p = f"downs.{i}"
        res(p + ".0", di, do)
        res(p + ".1", do, do)
        sattn(p + ".2", do)
        tattn(p + ".3", do)
        if i < n_res - 1:
            out.append((p + ".4.weight", (do, do, 1, 4, 4)))
            out.append((p + ".4.bias", (do,)))
    mid = cfg.dims[-1]
    res("mid_block1", mid, mid)
    # mid spatial attention: dense Attention without rotary (…conv3d.py:449)
    out.append(("mid_spatial_attn.fn.fn.fn.to_qkv.weight", (hid * 3, mid)))
    out.append(("mid_spatial_attn.fn.fn.fn.to_out.weight", (mid, hid)))
    out.append(("mid_spatial_attn.fn.norm.gamma", (1, mid, 1, 1, 1)))
    tattn("mid_temporal_attn", mid)
    res("mid_block2", mid, mid)
    for i, (di, do) in enumerate(reversed(cfg.in_out)):
        p = f"ups.{i}"
        res(p + ".0", do * 2, di)
        res(p + ".1", di, di)
        sattn(p + ".2", di)
        tattn(p + ".3", di)
        if i < n_res - 1:
            out.append((p + ".4.weight", (di, di, 1, 4, 4)))     # ConvTranspose3d: [Cin, Cout, kD, kH, kW]
            out.append((p + ".4.bias", (di,)))
    res("final_conv.0", cfg.dim * 2, cfg.dim, temb=False)
    out.append(("final_conv.1.weight", (cfg.out_dim, cfg.dim, 1, 1, 1)))
    out.append(("final_conv.1.bias", (cfg.out_dim,)))
    return out


def synthetic_state_dict(cfg, seed=0, scale=1.0):
    """Seeded synthetic weights with fan-in scaling (no checkpoints exist offline).

    Not the reference's default init (that one is pinned through the golden fixtures); this is
    the weight recipe used for GPU-vs-oracle parity at full width and for bench.py, generated
    identically on any box from `seed` by a NumPy PCG64 stream (independent of torch RNG).
    """
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("gamma") or name.endswith("norm.weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith("bias"):
            v = 0.05 * rng.standard_normal(shape)
        elif name.endswith("relative_attention_bias.weight"):
            v = rng.standard_normal(shape)
        else:
            if name.endswith(".4.weight") and name.startswith("ups"):
                fan_in = shape[0] * 4          # transposed conv: each output sees 2x2 taps x Cin
            else:
                fan_in = int(np.prod(shape[1:]))
            v = rng.uniform(-1.0, 1.0, size=shape) * math.sqrt(3.0 / fan_in) * scale
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd
