"""ORACLE (test infrastructure, not product): CPU restatement of the Burgers space-time U-Net `Unet2D`.

Follows /root/reference/model/burgers_1d/unet.py (`Unet2D` :267-431 and its blocks :22-264) as a functional fp32
torch-CPU program over a plain state-dict with the reference's parameter names.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity status: pinned against the reference module itself (imported in the build container through tools/refshim.py)
on tests/golden/unet2d_*.npz, incl. intermediate taps (tests/test_oracle_unet2d.py).
"""
import math

import torch
import torch.nn.functional as F


class Unet2DConfig:
    """Unet2D.__init__ arguments that the Burgers scripts vary (unet.py:272-288, train_1d_burgers.py:127-143)."""

    def __init__(self, dim=64, dim_mults=(1, 2, 4, 8), channels=2, out_dim=2, resnet_block_groups=1, attn_heads=4,
                 attn_dim_head=32):
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.channels = channels
        self.out_dim = channels if out_dim is None else out_dim
        self.groups = resnet_block_groups
        self.heads = attn_heads
        self.dim_head = attn_dim_head

    @property
    def dims(self):
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self):
        d = self.dims
        return list(zip(d[:-1], d[1:]))


def sinusoidal_pos_emb(time, dim, theta=10000):
    """unet.py:89-98 (even dim)."""
    half = dim // 2
    e = math.log(theta) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = time[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def layernorm(x, g, eps=1e-5):
    """Channel LayerNorm, unet.py:59-69."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


def block(sd, p, x, groups, scale_shift=None):
    """Conv2d 3x3 -> GroupNorm -> (scale+1, shift) -> SiLU, unet.py:134-155."""
    x = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1)
    x = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(sd, p, x, t_emb, groups):
    """unet.py:157-191."""
    te = F.linear(F.silu(t_emb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, :, None, None]
    scale_shift = te.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, groups, scale_shift)
    h = block(sd, p + ".block2", h, groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def linear_attention(sd, p, x, heads):
    """Residual(PreNorm(LinearAttention)), unet.py:193-236 (`p` is the Residual module, e.g. 'downs.0.2')."""
    b, c, hh, ww = x.shape
    y = layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = [t.reshape(b, heads, -1, hh * ww) for t in qkv]
    scale = q.shape[2] ** -0.5
    q = q.softmax(dim=-2)
    k = k.softmax(dim=-1)
    q = q * scale
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q)
    out = out.reshape(b, -1, hh, ww)
    out = F.conv2d(out, sd[p + ".fn.fn.to_out.0.weight"], sd[p + ".fn.fn.to_out.0.bias"])
    out = layernorm(out, sd[p + ".fn.fn.to_out.1.g"])
    return out + x


def attention(sd, p, x, heads):
    """Residual(PreNorm(Attention)), unet.py:238-272."""
    b, c, hh, ww = x.shape
    y = layernorm(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = [t.reshape(b, heads, -1, hh * ww) for t in qkv]
    q = q * q.shape[2] ** -0.5
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)
    out = out.permute(0, 1, 3, 2).reshape(b, -1, hh, ww)          # 'b h (x y) d -> b (h d) x y'
    out = F.conv2d(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"])
    return out + x


def downsample(sd, p, x):
    """Downsample2d: pixel-unshuffle(2) + 1x1 conv, unet.py:46-50."""
    b, c, hh, ww = x.shape
    x = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
    return F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"])


def upsample(sd, p, x):
    """Upsample2d: nearest x2 + 3x3 conv, unet.py:40-44."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"], padding=1)


def unet2d_forward(sd, cfg, x, time, taps=None):
    """Unet2D.forward, unet.py:387-431.  x [B, channels, Nt, Nx] fp32, time int64 [B]."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v.clone()

    n_res = len(cfg.in_out)
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    tap("init_conv", x)
    r = x.clone()
    t = sinusoidal_pos_emb(time, cfg.dim)
    t = F.linear(t, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    t = F.gelu(t)
    t = F.linear(t, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    tap("time_mlp", t)
    h = []
    for i in range(n_res):
        p = f"downs.{i}"
        x = resnet_block(sd, p + ".0", x, t, cfg.groups)
        tap(p + ".0", x)
        h.append(x)
        x = resnet_block(sd, p + ".1", x, t, cfg.groups)
        tap(p + ".1", x)
        x = linear_attention(sd, p + ".2", x, cfg.heads)
        tap(p + ".2", x)
        h.append(x)
        if i < n_res - 1:
            x = downsample(sd, p + ".3", x)
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = resnet_block(sd, "mid_block1", x, t, cfg.groups)
    tap("mid_block1", x)
    x = attention(sd, "mid_attn", x, cfg.heads)
    tap("mid_attn", x)
    x = resnet_block(sd, "mid_block2", x, t, cfg.groups)
    tap("mid_block2", x)
    for i in range(n_res):
        p = f"ups.{i}"
        x = torch.cat((x, h.pop()), dim=1)
        x = resnet_block(sd, p + ".0", x, t, cfg.groups)
        tap(p + ".0", x)
        x = torch.cat((x, h.pop()), dim=1)
        x = resnet_block(sd, p + ".1", x, t, cfg.groups)
        tap(p + ".1", x)
        x = linear_attention(sd, p + ".2", x, cfg.heads)
        tap(p + ".2", x)
        if i < n_res - 1:
            x = upsample(sd, p + ".3", x)
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = torch.cat((x, r), dim=1)
    x = resnet_block(sd, "final_res_block", x, t, cfg.groups)
    tap("final_res_block", x)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


def param_shapes(cfg):
    """(name, shape, kind) in the reference's registration order; kind = init recipe (torch defaults)."""
    out = []
    hid = cfg.heads * cfg.dim_head
    tdim = cfg.dim * 4
    in_out = cfg.in_out
    n_res = len(in_out)

    def res(p, di, do):
        out.append((p + ".mlp.1.weight", (do * 2, tdim), "w"))
        out.append((p + ".mlp.1.bias", (do * 2,), "b:" + str(tdim)))
        for b, ci in ((".block1", di), (".block2", do)):
            out.append((p + b + ".proj.weight", (do, ci, 3, 3), "w"))
            out.append((p + b + ".proj.bias", (do,), "b:" + str(ci * 9)))
            out.append((p + b + ".norm.weight", (do,), "one"))
            out.append((p + b + ".norm.bias", (do,), "zero"))
        if di != do:
            out.append((p + ".res_conv.weight", (do, di, 1, 1), "w"))
            out.append((p + ".res_conv.bias", (do,), "b:" + str(di)))

    def lin_attn(p, d):
        out.append((p + ".fn.fn.to_qkv.weight", (hid * 3, d, 1, 1), "w"))
        out.append((p + ".fn.fn.to_out.0.weight", (d, hid, 1, 1), "w"))
        out.append((p + ".fn.fn.to_out.0.bias", (d,), "b:" + str(hid)))
        out.append((p + ".fn.fn.to_out.1.g", (1, d, 1, 1), "one"))
        out.append((p + ".fn.norm.g", (1, d, 1, 1), "one"))

    out.append(("time_mlp.1.weight", (tdim, cfg.dim), "w"))
    out.append(("time_mlp.1.bias", (tdim,), "b:" + str(cfg.dim)))
    out.append(("time_mlp.3.weight", (tdim, tdim), "w"))
    out.append(("time_mlp.3.bias", (tdim,), "b:" + str(tdim)))
    out.append(("init_conv.weight", (cfg.dim, cfg.channels, 7, 7), "w"))
    out.append(("init_conv.bias", (cfg.dim,), "b:" + str(cfg.channels * 49)))
    for i, (di, do) in enumerate(in_out):
        p = f"downs.{i}"
        res(p + ".0", di, di)
        res(p + ".1", di, di)
        lin_attn(p + ".2", di)
        if i < n_res - 1:
            out.append((p + ".3.1.weight", (do, di * 4, 1, 1), "w"))
            out.append((p + ".3.1.bias", (do,), "b:" + str(di * 4)))
        else:
            out.append((p + ".3.weight", (do, di, 3, 3), "w"))
            out.append((p + ".3.bias", (do,), "b:" + str(di * 9)))
    mid = cfg.dims[-1]
    res("mid_block1", mid, mid)
    out.append(("mid_attn.fn.fn.to_qkv.weight", (hid * 3, mid, 1, 1), "w"))
    out.append(("mid_attn.fn.fn.to_out.weight", (mid, hid, 1, 1), "w"))
    out.append(("mid_attn.fn.fn.to_out.bias", (mid,), "b:" + str(hid)))
    out.append(("mid_attn.fn.norm.g", (1, mid, 1, 1), "one"))
    res("mid_block2", mid, mid)
    for i, (di, do) in enumerate(reversed(in_out)):
        p = f"ups.{i}"
        res(p + ".0", do + di, do)
        res(p + ".1", do + di, do)
        lin_attn(p + ".2", do)
        if i < n_res - 1:
            out.append((p + ".3.1.weight", (di, do, 3, 3), "w"))
            out.append((p + ".3.1.bias", (di,), "b:" + str(do * 9)))
        else:
            out.append((p + ".3.weight", (di, do, 3, 3), "w"))
            out.append((p + ".3.bias", (di,), "b:" + str(do * 9)))
    res("final_res_block", cfg.dim * 2, cfg.dim)
    out.append(("final_conv.weight", (cfg.out_dim, cfg.dim, 1, 1), "w"))
    out.append(("final_conv.bias", (cfg.out_dim,), "b:" + str(cfg.dim)))
    return out


def synthetic_state_dict(cfg, seed=0, perturb_norms=True):
    """Seeded random weights with torch's default init bounds (norm gains/biases perturbed so that they matter)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, kind in param_shapes(cfg):
        if kind == "one":
            v = torch.ones(shape)
            if perturb_norms:
                v = v + 0.1 * torch.randn(shape, generator=g)
        elif kind == "zero":
            v = torch.zeros(shape)
            if perturb_norms:
                v = v + 0.1 * torch.randn(shape, generator=g)
        elif kind.startswith("b:"):
            bound = 1.0 / math.sqrt(int(kind[2:]))
            v = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(shape, generator=g) * 2 - 1) * bound
        sd[name] = v
    return sd
