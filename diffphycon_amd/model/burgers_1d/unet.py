"""Drop-in `Unet2D`: the reference's constructor / state-dict / forward contract
(/root/reference/model/burgers_1d/unet.py:267-431) on top of libdpc.

Parameters live under exactly the reference's state_dict keys, so `cos10000-model-{m}.pt` checkpoints written by the
reference's Trainer (diffusion/diffusion_1d_burgers.py:949) load with `load_state_dict`.  `forward` runs entirely in
hand-written HIP kernels through the C ABI (include/dpc.h, dpc_unet2d_*); there is no torch-op fallback.
"""
import ctypes as C
import math

import torch
from torch import nn

from ... import _lib


def _param_shapes(dim, dim_mults, channels, out_dim, heads, dim_head):
    """(name, shape, kind) in the reference's registration order."""
    out = []
    hid = heads * dim_head
    tdim = dim * 4
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
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

    out.append(("time_mlp.1.weight", (tdim, dim), "w"))
    out.append(("time_mlp.1.bias", (tdim,), "b:" + str(dim)))
    out.append(("time_mlp.3.weight", (tdim, tdim), "w"))
    out.append(("time_mlp.3.bias", (tdim,), "b:" + str(tdim)))
    out.append(("init_conv.weight", (dim, channels, 7, 7), "w"))
    out.append(("init_conv.bias", (dim,), "b:" + str(channels * 49)))
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
    mid = dims[-1]
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
    res("final_res_block", dim * 2, dim)
    out.append(("final_conv.weight", (out_dim, dim, 1, 1), "w"))
    out.append(("final_conv.bias", (out_dim,), "b:" + str(dim)))
    return out


class _Node(nn.Module):
    """Plain container so that parameters appear under the reference's dotted state_dict keys."""


class Unet2D(nn.Module):
    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=2, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, sinusoidal_pos_emb_theta=10000,
                 attn_dim_head=32, attn_heads=4, condition_on_residual=None, micro_batch=0, arithmetic=None):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError("only the configuration get_2d_ddpm builds is supported (train_1d_burgers.py:127-143)")
        if condition_on_residual is not None:
            raise NotImplementedError("condition_on_residual raises NotImplementedError inside the reference's sampler "
                                      "as well (diffusion_1d_burgers.py:560-565)")
        if init_dim not in (None, dim) or attn_dim_head != 32:
            raise NotImplementedError("init_dim must equal dim and attn_dim_head must be 32")
        self.channels = channels                      # read by GaussianDiffusion (diffusion_1d_burgers.py:236)
        self.self_condition = False
        self.condition_on_residual = None
        self.random_or_learned_sinusoidal_cond = False
        self.dim, self.dim_mults = dim, tuple(dim_mults)
        self.out_dim = channels if out_dim is None else out_dim
        self.attn_heads, self.attn_dim_head = attn_heads, attn_dim_head
        self.resnet_block_groups = resnet_block_groups
        self.theta = sinusoidal_pos_emb_theta
        self.micro_batch = micro_batch
        self.arithmetic = arithmetic          # None: the process-wide libdpc mode (default f16x3); 'x6' | 'f32' = exact products
        self._names = []
        for name, shape, kind in _param_shapes(dim, self.dim_mults, channels, self.out_dim, attn_heads, attn_dim_head):
            self._register(name, self._init(shape, kind))
            self._names.append(name)
        self._handle = None
        self._dirty = True
        self._version = 0                     # upload generation: bumped whenever _sync re-uploads (graph caches key on it)
        self._taps = False
        self._ws = None
        self._device = None
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_dirty", True))

    @staticmethod
    def _init(shape, kind):
        if kind == "one":
            return torch.ones(shape)
        if kind == "zero":
            return torch.zeros(shape)
        if kind.startswith("b:"):
            bound = 1.0 / math.sqrt(int(kind[2:]))
            return torch.empty(shape).uniform_(-bound, bound)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(shape).uniform_(-bound, bound)

    def _register(self, name, value):
        parts = name.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))

    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def _ensure_handle(self):
        if self._handle is None:
            cfg = _lib.Unet2DCfg()
            cfg.dim, cfg.n_mults = self.dim, len(self.dim_mults)
            for i, m in enumerate(self.dim_mults):
                cfg.dim_mults[i] = m
            cfg.channels, cfg.out_dim = self.channels, self.out_dim
            cfg.attn_heads, cfg.attn_dim_head = self.attn_heads, self.attn_dim_head
            cfg.groups, cfg.micro_batch = self.resnet_block_groups, self.micro_batch
            h = C.c_void_p()
            _lib.create_with_mode(self.arithmetic, lambda: _lib.check(_lib.lib().dpc_unet2d_create(C.byref(cfg), C.byref(h))))
            self._handle = h

    @property
    def modes(self):
        self._ensure_handle()
        return _lib.lib().dpc_unet2d_modes(self._handle).decode()

    def _sync(self, device):
        L = _lib.lib()
        self._ensure_handle()
        if self._dirty or self._device != device:
            sd = self.state_dict()
            for name in self._names:
                w = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * w.dim())(*w.shape)
                _lib.check(L.dpc_unet2d_load(self._handle, name.encode(), _lib.ptr(w), shape, w.dim(), _lib.stream()))
            half = self.dim // 2
            freqs = torch.exp(torch.arange(half) * -(math.log(self.theta) / (half - 1))).float().to(device)   # (:93-95)
            _lib.check(L.dpc_unet2d_set_tables(self._handle, _lib.ptr(freqs), _lib.stream()))
            torch.cuda.current_stream().synchronize()
            _lib.check(L.dpc_unet2d_finalize(self._handle))
            self._dirty = False
            self._device = device
            self._version += 1

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().dpc_unet2d_destroy(self._handle)
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, x, time, x_self_cond=None, residual=None):
        """x [B, channels, Nt(16), Nx(128)], time int64 [B] -> [B, out_dim, Nt, Nx]   (unet.py:387-431)."""
        if x_self_cond is not None or residual is not None:
            raise NotImplementedError("self-conditioning / residual conditioning are not used by DiffPhyCon sampling")
        if not x.is_cuda:
            raise RuntimeError("Unet2D (libdpc) runs on the GPU only")
        B, Cc, H, W = x.shape
        assert Cc == self.channels
        x = x.contiguous().float()
        time = time.to(device=x.device, dtype=torch.long).contiguous()
        self._sync(x.device)
        L = _lib.lib()
        need = L.dpc_unet2d_workspace_bytes(self._handle, B, H, W)
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != x.device:
            self._ws = None
            self._ws = _lib.workspace(need, x.device)
        out = torch.empty((B, self.out_dim, H, W), device=x.device, dtype=torch.float32)
        _lib.check(L.dpc_unet2d_forward(self._handle, _lib.ptr(x), _lib.ptr(time, torch.long), _lib.ptr(out), B, H, W,
                                        C.c_void_p(self._ws.data_ptr()), self._ws.numel(), _lib.stream()))
        return out

    # test hooks
    def debug_taps(self, enable=True):
        self._ensure_handle()
        self._taps = bool(enable)
        _lib.check(_lib.lib().dpc_unet2d_debug_taps(self._handle, int(enable)))

    def get_tap(self, name, shape, device):
        out = torch.empty(shape, device=device, dtype=torch.float32)
        _lib.check(_lib.lib().dpc_unet2d_get_tap(self._handle, name.encode(), _lib.ptr(out), out.numel(), _lib.stream()))
        return out
