"""The two learned 2-D surrogates of the jellyfish task, with the reference's constructor and state_dict layout
(/root/reference/diffusion/diffusion_2d_jellyfish.py: `Unet` :276-403 boundary updater, `ForceUnet` :406-481).

These nets sit INSIDE the guidance gradient (inference_2d_jellyfish.py:85-114 differentiates through both).  The sampler does
NOT run these modules: since r02 the design gradient is computed by `model/surrogates_hip.py` (forward and input-gradient backward
of both nets on libdpc kernels, no autograd).  The stock-torch modules below remain as the checkpoint CONTAINER (same
state_dict keys as the reference, `HipUnet` / `HipForceUnet` read their weights from them) and as the test reference the HIP
path is compared with (tests/test_gpu_surrogates_hip.py)."""
import math

import torch
import torch.nn.functional as F
from torch import nn


class _WSConv2d(nn.Conv2d):
    """Weight-standardised conv (:107-120): per-output-filter zero mean / unit biased variance, eps 1e-5 in fp32."""

    def forward(self, x):
        eps = 1e-5 if x.dtype == torch.float32 else 1e-3
        w = self.weight
        mean = w.mean(dim=(1, 2, 3), keepdim=True)
        var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
        return F.conv2d(x, (w - mean) * (var + eps).rsqrt(), self.bias, self.stride, self.padding, self.dilation, self.groups)


class _ChanLN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))

    def forward(self, x):
        eps = 1e-5 if x.dtype == torch.float32 else 1e-3
        var = torch.var(x, dim=1, unbiased=False, keepdim=True)
        mean = torch.mean(x, dim=1, keepdim=True)
        return (x - mean) * (var + eps).rsqrt() * self.g


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _ChanLN(dim)

    def forward(self, x):
        return self.fn(self.norm(x))


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x) + x


class _Unshuffle(nn.Module):
    """'b c (h p1) (w p2) -> b (c p1 p2) h w' with p1 = p2 = 2 (:101-105)."""

    def forward(self, x):
        b, c, h, w = x.shape
        return x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)


def _down(dim, dim_out):
    return nn.Sequential(_Unshuffle(), nn.Conv2d(dim * 4, dim_out, 1))


def _up(dim, dim_out):
    return nn.Sequential(nn.Upsample(scale_factor=2, mode="nearest"), nn.Conv2d(dim, dim_out, 3, padding=1))


class _SinEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        e = math.log(10000) / (half - 1)
        e = torch.exp(torch.arange(half, device=t.device) * -e)
        e = t[:, None] * e[None, :]
        return torch.cat((e.sin(), e.cos()), dim=-1)


class _Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = _WSConv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)

    def forward(self, x, scale_shift=None):
        x = self.norm(self.proj(x))
        if scale_shift is not None:
            scale, shift = scale_shift
            x = x * (scale + 1) + shift
        return F.silu(x)


class _ResBlock(nn.Module):
    def __init__(self, dim, dim_out, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if time_emb_dim is not None else None
        self.block1 = _Block(dim, dim_out, groups)
        self.block2 = _Block(dim_out, dim_out, groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, t_emb=None):
        ss = None
        if self.mlp is not None and t_emb is not None:
            ss = self.mlp(t_emb)[:, :, None, None].chunk(2, dim=1)
        h = self.block2(self.block1(x, ss))
        return h + self.res_conv(x)


class _LinAttn(nn.Module):
    """:213-243 — note the extra `v / (h*w)` (absent from the 3-D U-Net's spatial linear attention)."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        hid = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hid * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hid, dim, 1), _ChanLN(dim))

    def forward(self, x):
        b, c, h, w = x.shape
        q, k, v = [t.reshape(b, self.heads, -1, h * w) for t in self.to_qkv(x).chunk(3, dim=1)]
        q = q.softmax(dim=-2) * self.scale
        k = k.softmax(dim=-1)
        v = v / (h * w)
        ctx = torch.einsum("bhdn,bhen->bhde", k, v)
        out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, -1, h, w)
        return self.to_out(out)


class _Attn(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale, self.heads = dim_head ** -0.5, heads
        hid = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hid * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hid, dim, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        q, k, v = [t.reshape(b, self.heads, -1, h * w) for t in self.to_qkv(x).chunk(3, dim=1)]
        sim = torch.einsum("bhdi,bhdj->bhij", q * self.scale, k)
        out = torch.einsum("bhij,bhdj->bhid", sim.softmax(dim=-1), v)
        return self.to_out(out.permute(0, 1, 3, 2).reshape(b, -1, h, w))


def _encoder(dims, time_dim, groups):
    in_out = list(zip(dims[:-1], dims[1:]))
    downs = nn.ModuleList([])
    for i, (di, do) in enumerate(in_out):
        last = i >= len(in_out) - 1
        downs.append(nn.ModuleList([
            _ResBlock(di, di, time_dim, groups), _ResBlock(di, di, time_dim, groups),
            _Residual(_PreNorm(di, _LinAttn(di))),
            _down(di, do) if not last else nn.Conv2d(di, do, 3, padding=1)]))
    return in_out, downs


class Unet(nn.Module):
    """Boundary updater: (bd_0 [N,3,H,W], delta_theta [N]) -> bd [N,3,H,W]."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError("only the configuration inference_2d_jellyfish.py builds (:268-273) is supported")
        self.channels, self.self_condition = channels, False
        init_dim = dim if init_dim is None else init_dim
        self.init_conv = nn.Conv2d(channels, init_dim, 7, padding=3)
        dims = [init_dim] + [dim * m for m in dim_mults]
        time_dim = dim * 4
        self.random_or_learned_sinusoidal_cond = False
        self.time_mlp = nn.Sequential(_SinEmb(dim), nn.Linear(dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        in_out, self.downs = _encoder(dims, time_dim, resnet_block_groups)
        self.ups = nn.ModuleList([])
        mid = dims[-1]
        self.mid_block1 = _ResBlock(mid, mid, time_dim, resnet_block_groups)
        self.mid_attn = _Residual(_PreNorm(mid, _Attn(mid)))
        self.mid_block2 = _ResBlock(mid, mid, time_dim, resnet_block_groups)
        for i, (di, do) in enumerate(reversed(in_out)):
            last = i == len(in_out) - 1
            self.ups.append(nn.ModuleList([
                _ResBlock(do + di, do, time_dim, resnet_block_groups), _ResBlock(do + di, do, time_dim, resnet_block_groups),
                _Residual(_PreNorm(do, _LinAttn(do))),
                _up(do, di) if not last else nn.Conv2d(do, di, 3, padding=1)]))
        self.out_dim = channels if out_dim is None else out_dim
        self.final_res_block = _ResBlock(dim * 2, dim, time_dim, resnet_block_groups)
        self.final_conv = nn.Conv2d(dim, self.out_dim, 1)

    def forward(self, x, time, x_self_cond=None):
        x = self.init_conv(x)
        r = x.clone()
        t = self.time_mlp(time)
        h = []
        for b1, b2, attn, down in self.downs:
            x = b1(x, t)
            h.append(x)
            x = attn(b2(x, t))
            h.append(x)
            x = down(x)
        x = self.mid_block2(self.mid_attn(self.mid_block1(x, t)), t)
        for b1, b2, attn, up in self.ups:
            x = b1(torch.cat((x, h.pop()), dim=1), t)
            x = attn(b2(torch.cat((x, h.pop()), dim=1), t))
            x = up(x)
        return self.final_conv(self.final_res_block(torch.cat((x, r), dim=1), t))


class ForceUnet(nn.Module):
    """Force surrogate: cat(pressure, bd) [N,4,H,W] -> force [N,out_dim] (encoder + spatial mean + Linear(512, out))."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False):
        super().__init__()
        if self_condition or learned_variance:
            raise NotImplementedError
        self.channels, self.self_condition = channels, False
        init_dim = dim if init_dim is None else init_dim
        self.init_conv = nn.Conv2d(channels, init_dim, 7, padding=3)
        dims = [init_dim] + [dim * m for m in dim_mults]
        _, self.downs = _encoder(dims, None, resnet_block_groups)
        self.ups = nn.ModuleList([])
        mid = dims[-1]
        self.mid_block1 = _ResBlock(mid, mid, None, resnet_block_groups)
        self.mid_attn = _Residual(_PreNorm(mid, _Attn(mid)))
        self.mid_block2 = _ResBlock(mid, mid, None, resnet_block_groups)
        self.final = nn.Linear(512, out_dim)           # hard-wired 512 = 64 * 8 in the reference (:454)

    def forward(self, x, x_self_cond=None):
        x = self.init_conv(x)
        for b1, b2, attn, down in self.downs:
            x = down(attn(b2(b1(x))))
        x = self.mid_block2(self.mid_attn(self.mid_block1(x)))
        return self.final(x.mean(dim=-1).mean(dim=-1))
