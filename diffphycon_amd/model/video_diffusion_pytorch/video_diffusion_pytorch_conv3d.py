"""Drop-in `Unet3D_with_Conv3D`: the reference's constructor / state-dict / forward contract
(/root/reference/model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:356-552) on top of libdpc.

The module holds its parameters under exactly the reference's state_dict keys, so checkpoints written by the
reference's Trainer.save (diffusion/diffusion_2d_smoke.py:942-954) load with `load_state_dict`.  `forward`
runs entirely in hand-written HIP kernels through the C ABI (include/dpc.h); there is no torch-op fallback.
"""
import ctypes as C
import math

import torch
from torch import nn

from ... import _lib


def exists(x):
    return x is not None


def _relative_position_bucket(n_frames, num_buckets=32, max_distance=32):
    """Host-side T5 bucket table (…conv3d.py:86-104), same fp32 arithmetic as the reference."""
    q = torch.arange(n_frames, dtype=torch.long)
    rel = q[None, :] - q[:, None]
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = torch.abs(n)
    max_exact = nb // 2
    is_small = n < max_exact
    val_if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    ).long()
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, nb - 1))
    return ret + torch.where(is_small, n, val_if_large)


def _rotary_tables(n_frames, dim_head, theta=10000.0):
    """cos/sin [F, dim_head] of the interleaved-pair rotary embedding (rotary-embedding-torch 0.8.4 semantics:
    freqs_i = theta^(-2i/d), pair (2i, 2i+1) shares angle pos*freqs_i).  See DESIGN.md: parity unpinned."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim_head, 2)[: dim_head // 2].float() / dim_head))
    ang = torch.einsum("i,j->ij", torch.arange(n_frames).float(), freqs).repeat_interleave(2, dim=-1)
    return ang.cos().contiguous(), ang.sin().contiguous()


def _param_shapes(dim, dim_mults, channels, out_dim, heads, dim_head, k):
    """(name, shape, kind) in the reference's registration order."""
    out = []
    hid = heads * dim_head
    tdim = dim * 4
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))

    def tattn(p, d):
        out.append((p + ".fn.fn.fn.to_qkv.weight", (hid * 3, d), "w"))
        out.append((p + ".fn.fn.fn.to_out.weight", (d, hid), "w"))
        out.append((p + ".fn.norm.gamma", (1, d, 1, 1, 1), "one"))

    def sattn(p, d):
        out.append((p + ".fn.fn.to_qkv.weight", (hid * 3, d, 1, 1), "w"))
        out.append((p + ".fn.fn.to_out.weight", (d, hid, 1, 1), "w"))
        out.append((p + ".fn.fn.to_out.bias", (d,), "b:" + str(hid)))
        out.append((p + ".fn.norm.gamma", (1, d, 1, 1, 1), "one"))

    def res(p, di, do, temb=True):
        if temb:
            out.append((p + ".mlp.1.weight", (do * 2, tdim), "w"))
            out.append((p + ".mlp.1.bias", (do * 2,), "b:" + str(tdim)))
        for b, ci in ((".block1", di), (".block2", do)):
            out.append((p + b + ".proj.weight", (do, ci, 3, 3, 3), "w"))
            out.append((p + b + ".proj.bias", (do,), "b:" + str(ci * 27)))
            out.append((p + b + ".norm.weight", (do,), "one"))
            out.append((p + b + ".norm.bias", (do,), "zero"))
        if di != do:
            out.append((p + ".res_conv.weight", (do, di, 1, 1, 1), "w"))
            out.append((p + ".res_conv.bias", (do,), "b:" + str(di)))

    out.append(("time_rel_pos_bias.relative_attention_bias.weight", (32, heads), "normal"))
    out.append(("init_conv.weight", (dim, channels, k, k, k), "w"))
    out.append(("init_conv.bias", (dim,), "b:" + str(channels * k ** 3)))
    tattn("init_temporal_attn", dim)
    out.append(("time_mlp.1.weight", (tdim, dim), "w"))
    out.append(("time_mlp.1.bias", (tdim,), "b:" + str(dim)))
    out.append(("time_mlp.3.weight", (tdim, tdim), "w"))
    out.append(("time_mlp.3.bias", (tdim,), "b:" + str(tdim)))
    n_res = len(in_out)
    for i, (di, do) in enumerate(in_out):
        p = f"downs.{i}"
        res(p + ".0", di, do)
        res(p + ".1", do, do)
        sattn(p + ".2", do)
        tattn(p + ".3", do)
        if i < n_res - 1:
            out.append((p + ".4.weight", (do, do, 1, 4, 4), "w"))
            out.append((p + ".4.bias", (do,), "b:" + str(do * 16)))
    mid = dims[-1]
    res("mid_block1", mid, mid)
    out.append(("mid_spatial_attn.fn.fn.fn.to_qkv.weight", (hid * 3, mid), "w"))
    out.append(("mid_spatial_attn.fn.fn.fn.to_out.weight", (mid, hid), "w"))
    out.append(("mid_spatial_attn.fn.norm.gamma", (1, mid, 1, 1, 1), "one"))
    tattn("mid_temporal_attn", mid)
    res("mid_block2", mid, mid)
    for i, (di, do) in enumerate(reversed(in_out)):
        p = f"ups.{i}"
        res(p + ".0", do * 2, di)
        res(p + ".1", di, di)
        sattn(p + ".2", di)
        tattn(p + ".3", di)
        if i < n_res - 1:
            out.append((p + ".4.weight", (di, di, 1, 4, 4), "wT"))
            out.append((p + ".4.bias", (di,), "b:" + str(di * 16)))
    res("final_conv.0", dim * 2, dim, temb=False)
    out.append(("final_conv.1.weight", (out_dim, dim, 1, 1, 1), "w"))
    out.append(("final_conv.1.bias", (out_dim,), "b:" + str(dim)))
    return out


class _Node(nn.Module):
    """Plain container so that parameters appear under the reference's dotted state_dict keys."""


class Unet3D_with_Conv3D(nn.Module):
    def __init__(self, dim, cond_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=6, attn_heads=4,
                 attn_dim_head=32, use_bert_text_cond=False, init_dim=None, init_kernel_size=7,
                 use_sparse_linear_attn=True, block_type="resnet", resnet_groups=8, micro_batch=0, arithmetic=None):
        super().__init__()
        if cond_dim is not None or use_bert_text_cond:
            raise NotImplementedError("text / cond_dim conditioning is unused on the DiffPhyCon path (…conv3d.py:366)")
        if init_dim not in (None, dim) or not use_sparse_linear_attn or block_type != "resnet":
            raise NotImplementedError("only the configuration the reference's inference scripts build is supported")
        if attn_dim_head != 32:
            raise NotImplementedError("attn_dim_head must be 32")
        self.channels = channels                      # read by GaussianDiffusion (diffusion_2d_smoke.py:481)
        self.self_condition = False                   # (:482)
        self.dim, self.dim_mults = dim, tuple(dim_mults)
        self.out_dim = channels if out_dim is None else out_dim
        self.attn_heads, self.attn_dim_head = attn_heads, attn_dim_head
        self.init_kernel_size, self.resnet_groups = init_kernel_size, resnet_groups
        self.micro_batch = micro_batch
        self.arithmetic = arithmetic          # None: the process-wide libdpc mode (default f16x3); 'x6' | 'f32' = exact products
        self._names = []
        for name, shape, kind in _param_shapes(dim, self.dim_mults, channels, self.out_dim, attn_heads, attn_dim_head,
                                               init_kernel_size):
            self._register(name, self._init(shape, kind))
            self._names.append(name)
        self._handle = None
        self._dirty = True
        self._frames = None
        self._ws = None
        self._device = None
        # fires also when a parent module (GaussianDiffusion) loads a checkpoint
        self.register_load_state_dict_post_hook(lambda module, _keys: setattr(module, "_dirty", True))

    # ------------------------------------------------------------------ parameters
    @staticmethod
    def _init(shape, kind):
        if kind == "one":
            return torch.ones(shape)
        if kind == "zero":
            return torch.zeros(shape)
        if kind == "normal":
            return torch.randn(shape)
        if kind.startswith("b:"):
            bound = 1.0 / math.sqrt(int(kind[2:]))
            return torch.empty(shape).uniform_(-bound, bound)
        fan_in = 1
        for s in (shape[1:] if kind == "w" else (shape[0],) + tuple(shape[2:])):
            fan_in *= s
        if kind == "wT":
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]      # torch's convention for ConvTranspose
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

    def load_state_dict(self, state_dict, strict=True, **kw):
        # the reference's checkpoints carry the rotary buffers '…rotary_emb.freqs' (not learned); ignore them
        sd = {k: v for k, v in state_dict.items() if not k.endswith("rotary_emb.freqs")}
        self._dirty = True
        return super().load_state_dict(sd, strict=strict, **kw)

    # ------------------------------------------------------------------ libdpc handle
    def _ensure_handle(self):
        L = _lib.lib()
        if self._handle is None:
            cfg = _lib.Unet3DCfg()
            cfg.dim, cfg.n_mults = self.dim, len(self.dim_mults)
            for i, m in enumerate(self.dim_mults):
                cfg.dim_mults[i] = m
            cfg.channels, cfg.out_dim = self.channels, self.out_dim
            cfg.attn_heads, cfg.attn_dim_head = self.attn_heads, self.attn_dim_head
            cfg.init_kernel, cfg.groups, cfg.micro_batch = self.init_kernel_size, self.resnet_groups, self.micro_batch
            h = C.c_void_p()
            _lib.create_with_mode(self.arithmetic, lambda: _lib.check(L.dpc_unet3d_create(C.byref(cfg), C.byref(h))))
            self._handle = h

    def _sync(self, device, frames):
        L = _lib.lib()
        self._ensure_handle()
        changed = False
        if self._dirty or self._device != device:
            changed = True
            sd = self.state_dict()
            for name in self._names:
                w = sd[name].detach().to(device=device, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * w.dim())(*w.shape)
                _lib.check(L.dpc_unet3d_load(self._handle, name.encode(), _lib.ptr(w), shape, w.dim(), _lib.stream()))
            self._frames = None
            self._dirty = False
            self._device = device
        if self._frames != frames:
            changed = True
            emb = self.state_dict()["time_rel_pos_bias.relative_attention_bias.weight"].detach().float().cpu()
            bias = emb[_relative_position_bucket(frames)].permute(2, 0, 1).contiguous()          # (:111-112)
            cos, sin = _rotary_tables(frames, min(32, self.attn_dim_head))
            half = self.dim // 2
            freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))              # (:146-148)
            tabs = [t.to(device).contiguous() for t in (bias, cos, sin, freqs.float())]
            _lib.check(L.dpc_unet3d_set_tables(self._handle, frames, *[_lib.ptr(t) for t in tabs], _lib.stream()))
            torch.cuda.current_stream().synchronize()      # tables are copied by the library before `tabs` dies
            self._frames = frames
        if changed:         # (finalize reads the weight-range flag back: a host sync, so only after loads -- never per forward)
            _lib.check(L.dpc_unet3d_finalize(self._handle))

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().dpc_unet3d_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, time, cond=None, null_cond_prob=0., focus_present_mask=None, prob_focus_present=0.):
        """x [B,F,C,H,W], time int64 [B] -> [B,F,out_dim,H,W]   (…conv3d.py:486-552).

        `cond` must be None (has_cond is False on this path); the focus-present arguments are inert for
        prob_focus_present == 0, the only value the reference's sampler uses.
        """
        if cond is not None or focus_present_mask is not None or prob_focus_present != 0.:
            raise NotImplementedError("conditioning / focus-present masks are not used by DiffPhyCon sampling")
        if not x.is_cuda:
            raise RuntimeError("Unet3D_with_Conv3D (libdpc) runs on the GPU only")
        B, F, Cc, H, W = x.shape
        assert Cc == self.channels
        # a channel slice x_full[:, :, a:a+C] of a contiguous tensor is read in place (no copy): the prior
        # model's input x[:, :, 3:5] (diffusion_2d_smoke.py:612) is such a view
        ctot, coff = 0, 0
        if (not x.is_contiguous() and x.dtype == torch.float32 and x.stride(4) == 1 and x.stride(3) == W
                and x.stride(2) == H * W and x.stride(1) % (H * W) == 0 and x.stride(0) == F * x.stride(1)
                and x.storage_offset() % (H * W) == 0):
            ctot = x.stride(1) // (H * W)
            coff = (x.storage_offset() // (H * W)) % ctot
            if coff + Cc > ctot:
                ctot = 0
        if ctot:
            base = x.as_strided((B, F, ctot, H, W), (F * ctot * H * W, ctot * H * W, H * W, W, 1),
                                x.storage_offset() - coff * H * W)
            x = base
        else:
            x = x.contiguous().float()
        time = time.to(device=x.device, dtype=torch.long).contiguous()
        self._sync(x.device, F)
        L = _lib.lib()
        need = L.dpc_unet3d_workspace_bytes(self._handle, B, F, H, W)
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != x.device:
            self._ws = None
            self._ws = _lib.workspace(need, x.device)
        out = torch.empty((B, F, self.out_dim, H, W), device=x.device, dtype=torch.float32)
        _lib.check(L.dpc_unet3d_forward(self._handle, _lib.ptr(x), ctot, coff, _lib.ptr(time, torch.long),
                                        _lib.ptr(out), B, F, H, W, C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                        _lib.stream()))
        return out

    @property
    def modes(self):
        """Arithmetic modes the libdpc handle captured, e.g. 'conv=f16x3,igemm=f16x3,attn=f16x3,stem=f16x3'."""
        self._ensure_handle()
        return _lib.lib().dpc_unet3d_modes(self._handle).decode()

    def set_range_check(self, enable=True):
        """Make forward() fail loudly when an activation leaves the range the f16x3 mode represents (|x| <= 4094) instead
        of clamping it (include/dpc.h: dpc_unet3d_set_range_check).  A validation aid for new checkpoints; costs a sync."""
        self._ensure_handle()
        _lib.check(_lib.lib().dpc_unet3d_set_range_check(self._handle, int(enable)))

    def set_arithmetic(self, arithmetic):
        """Re-create the library handle in another arithmetic mode (None = the process-wide default, 'f16x3' | 'x6' | 'f32'): the mode
        is captured when a handle is created (include/dpc.h), so the old handle is destroyed and the weights are re-loaded -- and
        re-packed for the new mode -- at the next forward.  Used by the samplers' opt-in exact retry (GaussianDiffusion.retry_exact)."""
        if arithmetic == self.arithmetic:
            return
        if self._handle is not None:
            _lib.lib().dpc_unet3d_destroy(self._handle)
            self._handle = None
        self.arithmetic = arithmetic
        self._dirty, self._frames, self._device = True, None, None

    def check_range(self, reset=True):
        """Raise if any forward since the last check produced a residual-stream activation outside the f16x3 range (|x| > 4094
        or non-finite): the always-on sentinel of include/dpc.h dpc_unet3d_range_status.  One host sync; the samplers call it
        once at the end of sample(), never per step."""
        if self._handle is not None:
            _lib.check(_lib.lib().dpc_unet3d_range_status(self._handle, int(reset), _lib.stream()))

    # test hook
    def debug_taps(self, enable=True):
        self._ensure_handle()
        _lib.check(_lib.lib().dpc_unet3d_debug_taps(self._handle, int(enable)))

    def get_tap(self, name, shape, device):
        out = torch.empty(shape, device=device, dtype=torch.float32)
        _lib.check(_lib.lib().dpc_unet3d_get_tap(self._handle, name.encode(), _lib.ptr(out), out.numel(), _lib.stream()))
        return out
