"""Training pass of `Unet3D_with_Conv3D` on libdpc: forward WITH a tape and the hand-written reverse pass that produces every
parameter gradient of /root/reference/model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:356-552 under the loss of
/root/reference/diffusion/diffusion_2d_smoke.py p_losses :809-831 -- what `accelerator.backward(loss)` (Trainer.train :1025)
computes in the reference.  No autograd graph, no torch convolution / matmul: every tensor-sized operation is a HIP kernel
behind include/dpc.h (csrc/train.hip, csrc/surr.hip and the forward kernels); torch allocates HBM and does [B, C]-sized
bookkeeping (sinusoidal embedding, bucket table gather).

Parameters live in ONE flat fp32 buffer (`w`), gradients in another (`g`): the module's nn.Parameters are views into `w`, so
`state_dict()` / checkpoints see trained weights; the optimizer, the gradient norm and the data-parallel all-reduce are one
launch / one collective over the flat buffers.

Activations are channels-last fp32 [B F H W, C].  Arithmetic: forward GEMM-shaped ops in the library's default mode (f16x3);
backward-DATA convolutions are the forward operators on flipped / transposed weights in `bwd_mode` -- "x6" (default: exact fp32
products on the bf16 matrix cores; gradients span 1e-9 .. 1e-2 and need no range management) or "f16x3" with a power-of-two loss
scale (`loss_scale`, undone in the optimizer kernel).  WEIGHT gradients: `wgrad_mode` "f16x3" (default) runs the 3x3x3 convolutions'
weight gradients -- 95 % of the weight-gradient flop -- on the fp16 matrix cores from 22-bit split operands (x * 2^4, dy * 2^24 /
loss_scale, both SATURATING at 65504: |x| <= 4094, |d loss / d conv output| <= 3.9e-3 * loss_scale; csrc/wgrad3.hip raises a device
word when it clamps, `Trainer.check_gradient_range()` / dpc_train_range_status read it), every other geometry -- and every
geometry under wgrad_mode "f32" -- uses exact fp32 products on the native fp32 MFMA.
"""
import ctypes as C
import math

import torch

from ... import _lib
from .video_diffusion_pytorch_conv3d import _relative_position_bucket, _rotary_tables


def _flipT(w):
    """backward-data weight of a stride-1 'same' conv: [N, K, kd, kh, kw] -> [K, N, kd, kh, kw], taps reversed."""
    return w.flip(2, 3, 4).transpose(0, 1).contiguous()


def _as5(w):
    """Linear [N, K] / Conv2d [N, K, 1, 1] / Conv3d weights as [N, K, kd, kh, kw]."""
    if w.dim() == 2:
        return w[:, :, None, None, None]
    if w.dim() == 4:
        return w[:, :, None]
    return w


def _parity_weights(wT):
    """ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) weight [Kin, Nout, 1, 4, 4] (...conv3d.py:159-160) as its four output-parity
    classes: out[2i+a][2j+b] = a dense 2 x 2 stride-1 convolution with padding (1-a, 1-b) and weights [Nout, Kin, 1, 2, 2]."""
    taps = ((3, 1), (2, 0))
    out = []
    for a in range(2):
        for b in range(2):
            wc = wT[:, :, 0][:, :, list(taps[a])][:, :, :, list(taps[b])]           # [Kin, Nout, 2, 2]
            out.append((a, b, wc.permute(1, 0, 2, 3)[:, :, None].contiguous()))
    return out


class _Ctx:
    """Device, flat parameter / gradient views, shared scratch and thin wrappers over the C ABI."""

    def __init__(self, device, groups, heads, fwd_mode, bwd_mode):
        self.device, self.groups, self.heads = device, groups, heads
        self.fwd_mode, self.bwd_mode = (fwd_mode or ""), bwd_mode
        self.W, self.G = {}, {}
        self._ws = None
        self.act_scale = 0.0           # operand scale of the f16x3 backward-data convolutions (0: the library default 2^4)
        self.wgrad_dy_scale = 0.0      # != 0: 3x3x3 weight gradients on the fp16 matrix cores, dy pre-scaled by this power of two
        # f16x3 backward-data convolutions clamp |scaled gradient| > 65504 / 2^4: every layer's weight-gradient launch watches the
        # same tensor and raises the sentinel (include/dpc.h: dpc_conv_wgrad_cl dy_abs_limit) -- no clamp goes unseen
        self.dgrad_limit = 4094.0 if bwd_mode == "f16x3" else 0.0

    def ws(self, nbytes):
        nbytes = int(nbytes) + 512
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return C.c_void_p(self._ws.data_ptr()), self._ws.numel()

    def empty(self, *shape):
        return torch.empty(*shape, device=self.device, dtype=torch.float32)

    # ---- norms
    def gn_stats(self, x, B, R, Cc):
        st = self.empty(B, self.groups, 2)
        p, n = self.ws(_lib.lib().dpc_gn_workspace_bytes(B, Cc))
        _lib.check(_lib.lib().dpc_gn_stats(_lib.ptr(x), _lib.ptr(st), B, R, Cc, self.groups, p, n, _lib.stream()))
        return st

    def gn_apply(self, x, st, gamma, beta, ss, B, R, Cc, resid=None):
        out = torch.empty_like(x)
        _lib.check(_lib.lib().dpc_gn_apply(_lib.ptr(x), _lib.ptr(out), _lib.ptr(resid), _lib.ptr(st), _lib.ptr(gamma), _lib.ptr(beta),
                                           _lib.ptr(ss), B, R, Cc, self.groups, _lib.stream()))
        return out

    def gn_bwd(self, x, dy, st, gname, bname, ss, B, R, Cc):
        """-> dx, d(scale | shift) [B, 2C] or None; writes d gamma / d beta into the gradient buffer."""
        dx = torch.empty_like(x)
        dss = self.empty(B, 2 * Cc) if ss is not None else None
        p, n = self.ws(_lib.lib().dpc_gn_workspace_bytes(B, Cc))
        _lib.check(_lib.lib().dpc_gn_silu_bwd_params(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(st), _lib.ptr(self.W[gname]), _lib.ptr(self.W[bname]),
                                                     _lib.ptr(ss), _lib.ptr(dx), _lib.ptr(dss), _lib.ptr(self.G[gname]),
                                                     _lib.ptr(self.G[bname]), B, R, Cc, self.groups, p, n, _lib.stream()))
        return dx, dss

    def ln_stats(self, x):
        st = self.empty(x.shape[0], 2)
        _lib.check(_lib.lib().dpc_ln_stats(_lib.ptr(x), _lib.ptr(st), x.shape[0], x.shape[1], _lib.stream()))
        return st

    def ln_apply(self, x, st, g):
        out = torch.empty_like(x)
        _lib.check(_lib.lib().dpc_ln_apply(_lib.ptr(x), _lib.ptr(st), _lib.ptr(g), None, _lib.ptr(out), x.shape[0], x.shape[1], _lib.stream()))
        return out

    def ln_bwd(self, x, st, g, dy, dx):
        """dx += LN backward of dy (dx is the residual branch's gradient)."""
        _lib.check(_lib.lib().dpc_ln_bwd(_lib.ptr(x), _lib.ptr(st), _lib.ptr(g), _lib.ptr(dy), _lib.ptr(dx), x.shape[0], x.shape[1], 1,
                                         _lib.stream()))
        return dx

    def colsum(self, dy, out, x=None, st=None):
        Cc = dy.shape[1]
        p, n = self.ws(_lib.lib().dpc_colsum_workspace_bytes(Cc))
        _lib.check(_lib.lib().dpc_colsum(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(st), _lib.ptr(out), dy.shape[0], Cc, 1.0, 0, p, n, _lib.stream()))

    # ---- small dense layers
    def linear(self, x, Wn, bn, in_act=0):
        Wt = self.W[Wn]
        out = self.empty(x.shape[0], Wt.shape[0])
        _lib.check(_lib.lib().dpc_small_linear(_lib.ptr(x), _lib.ptr(Wt), _lib.ptr(self.W[bn]), _lib.ptr(out), x.shape[0], Wt.shape[1],
                                               Wt.shape[0], in_act, 0, _lib.stream()))
        return out

    def linear_bwd(self, dy, x, Wn, bn, in_act, dx=None, accumulate=False):
        Wt = self.W[Wn]
        _lib.check(_lib.lib().dpc_small_linear_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(Wt), _lib.ptr(dx), _lib.ptr(self.G[Wn]),
                                                   _lib.ptr(self.G[bn]), x.shape[0], Wt.shape[1], Wt.shape[0], in_act,
                                                   1 if accumulate else 0, _lib.stream()))

    def add_(self, y, x):
        _lib.check(_lib.lib().dpc_add_inplace(_lib.ptr(y), _lib.ptr(x), y.numel(), _lib.stream()))
        return y

    # ---- weight gradient
    def wgrad(self, x, dy, wname, geom, B, F, Hi, Wi, Ho, Wo, c_valid=0, ctot=None, coff=0, dw=None, x_is_grad=False):
        kd, kh, kw, sh, sw, pd, ph, pw = geom
        Cc, N = x.shape[1], dy.shape[1]
        dw = self.G[wname] if dw is None else dw
        ctot = Cc if ctot is None else ctot
        L = _lib.lib()
        p, n = self.ws(L.dpc_conv_wgrad_workspace_bytes(Cc, N, kd, kh, kw, B * F * Ho))
        _lib.check(L.dpc_conv_wgrad_cl(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), B, F, Hi, Wi, Cc, Ho, Wo, N, kd, kh, kw, sh, sw, pd, ph, pw,
                                       c_valid, ctot, coff, 1.0, self.wgrad_dy_scale, self.dgrad_limit, 2 if x_is_grad else 0, p, n,
                                       _lib.stream()))


class _Pack:
    """dpc_conv_t owner that re-packs in place."""

    def __init__(self, mode):
        self.h, self.mode = C.c_void_p(), mode
        self.N = self.K = 0

    def pack(self, w5, geom):
        w5 = w5.contiguous()
        self.N, self.K = w5.shape[0], w5.shape[1]
        kd, kh, kw, sh, sw, pd, ph, pw = geom
        _lib.check(_lib.lib().dpc_conv3_pack(_lib.ptr(w5), self.N, self.K, kd, kh, kw, sh, sw, pd, ph, pw, self.mode.encode(),
                                             C.byref(self.h), _lib.stream()))

    def run(self, a0, B, F, Hi, Wi, Ho=None, Wo=None, a1=None, bias=None, resid=None, out=None, ln=None, out_mode=0, par=(0, 0),
            act_scale=0.0):
        Ho = Hi if Ho is None else Ho
        Wo = Wi if Wo is None else Wo
        C0 = a0.shape[1]
        C1 = a1.shape[1] if a1 is not None else 0
        if out is None:
            rows = B * F * Ho * Wo * (4 if out_mode == 2 else 1)
            out = torch.empty(rows, self.N, device=a0.device, dtype=torch.float32)
        _lib.check(_lib.lib().dpc_conv3_run(self.h, _lib.ptr(a0), _lib.ptr(a1), C0, C1, _lib.ptr(bias), _lib.ptr(resid), _lib.ptr(out), B, F,
                                            Hi, Wi, Ho, Wo, _lib.ptr(ln[0]) if ln else None, _lib.ptr(ln[1]) if ln else None, out_mode,
                                            par[0], par[1], act_scale, _lib.stream()))
        return out

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        L = getattr(_lib, "_lib", None) if _lib is not None else None
        if h and L is not None:
            L.dpc_conv_free(h)


class _Conv:
    """A stride-1 'same' Conv3d / 1x1 conv / Linear of the net: forward, backward-data per input source, weight + bias gradient."""

    def __init__(self, ctx, wname, bname, k, splits=None):
        self.ctx, self.wname, self.bname = ctx, wname, bname
        self.geom = (k, k, k, 1, 1, k // 2, k // 2, k // 2)
        w = _as5(ctx.W[wname])
        self.N, self.K = w.shape[0], w.shape[1]
        self.splits = splits or (self.K,)
        self.f = _Pack(ctx.fwd_mode)
        self.d = [_Pack(ctx.bwd_mode) for _ in self.splits]

    def repack(self):
        w = _as5(self.ctx.W[self.wname])
        self.f.pack(w, self.geom)
        c0 = 0
        for d, cs in zip(self.d, self.splits):
            d.pack(_flipT(w[:, c0:c0 + cs]), self.geom)
            c0 += cs

    def fwd(self, a0, dims, a1=None, resid=None, ln=None, out=None, out_mode=0):
        B, F, H, W = dims
        bias = self.ctx.W[self.bname] if self.bname else None
        return self.f.run(a0, B, F, H, W, a1=a1, bias=bias, resid=resid, ln=ln, out=out, out_mode=out_mode)

    def dgrad(self, dy, dims, src=0, resid=None, out=None):
        B, F, H, W = dims
        return self.d[src].run(dy, B, F, H, W, resid=resid, out=out, act_scale=self.ctx.act_scale)

    def wgrad(self, xs, dy, dims):
        """xs: the input source tensors (1 or 2), dy: gradient of the conv output."""
        B, F, H, W = dims
        c0 = 0
        for x in xs:
            self.ctx.wgrad(x, dy, self.wname, self.geom, B, F, H, W, H, W, ctot=self.K, coff=c0)
            c0 += x.shape[1]
        if self.bname:
            self.ctx.colsum(dy, self.ctx.G[self.bname])


class _Res:
    """ResnetBlock (...conv3d.py:206-230) on cat(x0, x1)."""

    def __init__(self, ctx, p, C0, C1, Cout, has_time):
        self.ctx, self.p, self.C0, self.C1, self.Cout = ctx, p, C0, C1, Cout
        splits = (C0, C1) if C1 else (C0,)
        self.c1 = _Conv(ctx, p + ".block1.proj.weight", p + ".block1.proj.bias", 3, splits)
        self.c2 = _Conv(ctx, p + ".block2.proj.weight", p + ".block2.proj.bias", 3)
        self.cr = _Conv(ctx, p + ".res_conv.weight", p + ".res_conv.bias", 1, splits) if (p + ".res_conv.weight") in ctx.W else None
        self.has_time = has_time
        assert self.cr is not None or (C1 == 0 and C0 == Cout)

    def convs(self):
        return [c for c in (self.c1, self.c2, self.cr) if c is not None]

    def forward(self, x0, x1, t, dims):
        ctx, p, Cc = self.ctx, self.p, self.Cout
        B, F, H, W = dims
        R = F * H * W
        ss = ctx.linear(t, p + ".mlp.1.weight", p + ".mlp.1.bias", in_act=1) if self.has_time else None
        raw1 = self.c1.fwd(x0, dims, a1=x1)
        st1 = ctx.gn_stats(raw1, B, R, Cc)
        a1 = ctx.gn_apply(raw1, st1, ctx.W[p + ".block1.norm.weight"], ctx.W[p + ".block1.norm.bias"], ss, B, R, Cc)
        raw2 = self.c2.fwd(a1, dims)
        st2 = ctx.gn_stats(raw2, B, R, Cc)
        res = self.cr.fwd(x0, dims, a1=x1) if self.cr is not None else x0
        out = ctx.gn_apply(raw2, st2, ctx.W[p + ".block2.norm.weight"], ctx.W[p + ".block2.norm.bias"], None, B, R, Cc, resid=res)
        self.tape = (x0, x1, raw1, st1, a1, raw2, st2, ss, t, dims)
        return out

    def backward(self, dout, dt, need_dx=True):
        """dout: gradient of the block output.  Returns (dx0, dx1); accumulates the gradient of SiLU's input t into dt."""
        x0, x1, raw1, st1, a1, raw2, st2, ss, t, dims = self.tape
        self.tape = None
        ctx, p, Cc = self.ctx, self.p, self.Cout
        B, F, H, W = dims
        R = F * H * W
        xs = [x0] + ([x1] if x1 is not None else [])
        d_raw2, _ = ctx.gn_bwd(raw2, dout, st2, p + ".block2.norm.weight", p + ".block2.norm.bias", None, B, R, Cc)
        self.c2.wgrad([a1], d_raw2, dims)
        d_a1 = self.c2.dgrad(d_raw2, dims)
        del d_raw2, raw2, a1
        d_raw1, dss = ctx.gn_bwd(raw1, d_a1, st1, p + ".block1.norm.weight", p + ".block1.norm.bias", ss, B, R, Cc)
        del d_a1, raw1
        self.c1.wgrad(xs, d_raw1, dims)
        if self.has_time:
            ctx.linear_bwd(dss, t, p + ".mlp.1.weight", p + ".mlp.1.bias", 1, dx=dt, accumulate=True)
        if self.cr is not None:
            self.cr.wgrad(xs, dout, dims)
        if not need_dx:
            return None, None
        dx0 = self.c1.dgrad(d_raw1, dims, 0)
        dx1 = self.c1.dgrad(d_raw1, dims, 1) if x1 is not None else None
        if self.cr is None:
            ctx.add_(dx0, dout)
        else:
            self.cr.dgrad(dout, dims, 0, resid=dx0, out=dx0)
            if x1 is not None:
                self.cr.dgrad(dout, dims, 1, resid=dx1, out=dx1)
        return dx0, dx1


class _AttnBase:
    """Residual(PreNorm(attention)) (:195-204, :123-130): LayerNorm -> to_qkv -> core -> to_out (+ x)."""

    def __init__(self, ctx, p, Cc, inner, has_out_bias):
        self.ctx, self.p, self.Cc = ctx, p, Cc
        self.gname = p + ".fn.norm.gamma"
        self.cq = _Conv(ctx, p + inner + ".to_qkv.weight", None, 1)
        self.co = _Conv(ctx, p + inner + ".to_out.weight", (p + inner + ".to_out.bias") if has_out_bias else None, 1)

    def convs(self):
        return [self.cq, self.co]

    def gamma(self):
        return self.ctx.W[self.gname].reshape(-1)

    def _pre(self, x, dims):
        st = self.ctx.ln_stats(x)
        qkv = self.cq.fwd(x, dims, ln=(st, self.gamma()))
        return st, qkv

    def _post_backward(self, x, st, qkv, att, dqkv, dy, dims):
        """Given dqkv: weight gradients of to_qkv / gamma and the input gradient accumulated into dy (returned as dx)."""
        ctx = self.ctx
        xn = ctx.ln_apply(x, st, self.gamma())
        self.cq.wgrad([xn], dqkv, dims)
        del xn
        d_xn = self.cq.dgrad(dqkv, dims)
        ctx.colsum(d_xn, ctx.G[self.gname].reshape(-1), x=x, st=st)
        return ctx.ln_bwd(x, st, self.gamma(), d_xn, dy)


class _TAttn(_AttnBase):
    """temporal attention (:276-352, EinopsToAndFrom 'b c f h w' -> 'b (h w) f c'): sequences = pixels, tokens = frames."""

    def __init__(self, ctx, p, Cc, net):
        super().__init__(ctx, p, Cc, ".fn.fn.fn", False)
        self.net = net

    def forward(self, x, dims):
        B, F, H, W = dims
        st, qkv = self._pre(x, dims)
        att = self.ctx.empty(x.shape[0], self.ctx.heads * 32)
        n = self.net
        _lib.check(_lib.lib().dpc_attention_core(_lib.ptr(qkv), _lib.ptr(att), self.ctx.heads, F, B * H * W, H * W, F * H * W, 1, H * W,
                                                 _lib.ptr(n.rot_cos), _lib.ptr(n.rot_sin), _lib.ptr(n.pos_bias), _lib.stream()))
        y = self.co.fwd(att, dims, resid=x)
        self.tape = (x, st, qkv, att, dims)
        return y

    def backward(self, dy):
        x, st, qkv, att, dims = self.tape
        self.tape = None
        B, F, H, W = dims
        ctx, n, L = self.ctx, self.net, _lib.lib()
        self.co.wgrad([att], dy, dims)
        d_att = self.co.dgrad(dy, dims)
        dqkv = torch.empty_like(qkv)
        p, nb = ctx.ws(L.dpc_attention_bwd_seq_workspace_bytes(ctx.heads, F))
        _lib.check(L.dpc_attention_bwd_seq(_lib.ptr(qkv), _lib.ptr(d_att), _lib.ptr(dqkv), _lib.ptr(n.d_pos_bias), ctx.heads, F, B * H * W,
                                           H * W, F * H * W, 1, H * W, _lib.ptr(n.rot_cos), _lib.ptr(n.rot_sin), _lib.ptr(n.pos_bias),
                                           1, p, nb, _lib.stream()))
        del d_att
        return self._post_backward(x, st, qkv, att, dqkv, dy, dims)


class _SAttn(_AttnBase):
    """spatial linear attention (:232-257): per frame image, tokens = pixels."""

    def __init__(self, ctx, p, Cc):
        super().__init__(ctx, p, Cc, ".fn.fn", True)

    def forward(self, x, dims):
        B, F, H, W = dims
        L = _lib.lib()
        st, qkv = self._pre(x, dims)
        att = self.ctx.empty(x.shape[0], self.ctx.heads * 32)
        tape = torch.empty(L.dpc_linear_attention_tape_bytes(B * F, self.ctx.heads), dtype=torch.uint8, device=x.device)
        _lib.check(L.dpc_linear_attention_fwd_save(_lib.ptr(qkv), _lib.ptr(att), self.ctx.heads, B * F, H * W, C.c_void_p(tape.data_ptr()),
                                                   tape.numel(), _lib.stream()))
        y = self.co.fwd(att, dims, resid=x)
        self.tape = (x, st, qkv, att, tape, dims)
        return y

    def backward(self, dy):
        x, st, qkv, att, tape, dims = self.tape
        self.tape = None
        B, F, H, W = dims
        self.co.wgrad([att], dy, dims)
        d_att = self.co.dgrad(dy, dims)
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().dpc_linear_attention_bwd(_lib.ptr(qkv), _lib.ptr(d_att), _lib.ptr(dqkv), self.ctx.heads, B * F, H * W,
                                                       C.c_void_p(tape.data_ptr()), tape.numel(), _lib.stream()))
        del d_att
        return self._post_backward(x, st, qkv, att, dqkv, dy, dims)


class _MAttn(_AttnBase):
    """mid_spatial_attn (:455-457): dense attention over the pixels of a frame, no rotary, no bias."""

    def __init__(self, ctx, p, Cc):
        super().__init__(ctx, p, Cc, ".fn.fn.fn", False)

    def forward(self, x, dims):
        B, F, H, W = dims
        st, qkv = self._pre(x, dims)
        att = self.ctx.empty(x.shape[0], self.ctx.heads * 32)
        _lib.check(_lib.lib().dpc_attention_core(_lib.ptr(qkv), _lib.ptr(att), self.ctx.heads, H * W, B * F, 1, H * W, 0, 1, None, None, None,
                                                 _lib.stream()))
        y = self.co.fwd(att, dims, resid=x)
        self.tape = (x, st, qkv, att, dims)
        return y

    def backward(self, dy):
        x, st, qkv, att, dims = self.tape
        self.tape = None
        B, F, H, W = dims
        if H * W > 256:
            raise NotImplementedError("mid_spatial_attn backward: at most 256 pixels per frame at the bottleneck (S64: 16 x 16)")
        self.co.wgrad([att], dy, dims)
        d_att = self.co.dgrad(dy, dims)
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().dpc_attention_bwd(_lib.ptr(qkv), _lib.ptr(d_att), _lib.ptr(dqkv), self.ctx.heads, B * F, H * W, _lib.stream()))
        del d_att
        return self._post_backward(x, st, qkv, att, dqkv, dy, dims)


_G144 = (1, 4, 4, 2, 2, 0, 1, 1)


class _Down:
    """Downsample = Conv3d (1,4,4)/(1,2,2)/(0,1,1) (:162-163); backward-data = the transposed convolution as 4 parity classes."""

    def __init__(self, ctx, p):
        self.ctx, self.wname, self.bname = ctx, p + ".weight", p + ".bias"
        self.f = _Pack(ctx.fwd_mode)
        self.d = [_Pack(ctx.bwd_mode) for _ in range(4)]

    def repack(self):
        w = self.ctx.W[self.wname]
        self.f.pack(w, _G144)
        for d, (a, b, wc) in zip(self.d, _parity_weights(w)):          # w as ConvTranspose weight [in = Cout][out = Cin]
            d.pack(wc, (1, 2, 2, 1, 1, 0, 1 - a, 1 - b))

    def convs(self):
        return [self]

    def forward(self, x, dims):
        B, F, H, W = dims
        self.tape = (x, dims)
        return self.f.run(x, B, F, H, W, H // 2, W // 2, bias=self.ctx.W[self.bname])

    def backward(self, dy):
        x, (B, F, H, W) = self.tape
        self.tape = None
        ctx = self.ctx
        ctx.wgrad(x, dy, self.wname, _G144, B, F, H, W, H // 2, W // 2)
        ctx.colsum(dy, ctx.G[self.bname])
        dx = torch.empty_like(x)
        for d, (a, b) in zip(self.d, ((0, 0), (0, 1), (1, 0), (1, 1))):
            d.run(dy, B, F, H // 2, W // 2, out=dx, out_mode=2, par=(a, b), act_scale=ctx.act_scale)
        return dx


class _Up:
    """Upsample = ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) (:159-160): forward as 4 parity classes; backward-data = the strided
    convolution with the same weight tensor; weight gradient = the strided conv's with the roles of input / output swapped."""

    def __init__(self, ctx, p):
        self.ctx, self.wname, self.bname = ctx, p + ".weight", p + ".bias"
        self.f = [_Pack(ctx.fwd_mode) for _ in range(4)]
        self.d = _Pack(ctx.bwd_mode)

    def repack(self):
        w = self.ctx.W[self.wname]                                       # [Cin][Cout][1][4][4]
        for f, (a, b, wc) in zip(self.f, _parity_weights(w)):
            f.pack(wc, (1, 2, 2, 1, 1, 0, 1 - a, 1 - b))
        self.d.pack(w, _G144)                                            # as Conv3d weight [N = Cin][K = Cout]

    def convs(self):
        return [self]

    def forward(self, x, dims):
        B, F, H, W = dims
        out = self.ctx.empty(B * F * 4 * H * W, self.ctx.W[self.wname].shape[1])
        for f, (a, b) in zip(self.f, ((0, 0), (0, 1), (1, 0), (1, 1))):
            f.run(x, B, F, H, W, bias=self.ctx.W[self.bname], out=out, out_mode=2, par=(a, b))
        self.tape = (x, dims)
        return out

    def backward(self, dy):
        x, (B, F, H, W) = self.tape
        self.tape = None
        ctx = self.ctx
        ctx.wgrad(dy, x, self.wname, _G144, B, F, 2 * H, 2 * W, H, W, x_is_grad=True)
        ctx.colsum(dy, ctx.G[self.bname])
        return self.d.run(dy, B, F, 2 * H, 2 * W, H, W, act_scale=ctx.act_scale)


class TrainableUnet3D:
    """Forward-with-tape and backward of a `Unet3D_with_Conv3D` module (its parameters are re-pointed into a flat buffer)."""

    def __init__(self, module, device=None, bwd_mode="x6", fwd_mode=None, loss_scale=1.0, wgrad_mode="f16x3"):
        device = torch.device(device or "cuda")
        if device.type != "cuda":
            raise RuntimeError("TrainableUnet3D needs a GPU (libdpc has no CPU path)")
        if bwd_mode not in ("x6", "f16x3", "f32"):
            raise ValueError("bwd_mode: 'x6' | 'f16x3' | 'f32'")
        self.module, self.device = module, device
        m = module
        self.dim, self.mults, self.channels, self.out_dim = m.dim, tuple(m.dim_mults), m.channels, m.out_dim
        self.ctx = ctx = _Ctx(device, m.resnet_groups, m.attn_heads, fwd_mode, bwd_mode)
        if wgrad_mode not in ("f16x3", "f32"):
            raise ValueError("wgrad_mode: 'f16x3' (3x3x3 weight gradients on the fp16 matrix cores) | 'f32' (native fp32 MFMA)")
        self.wgrad_mode = wgrad_mode
        self.set_loss_scale(loss_scale)
        # ---- flat parameter / gradient buffers; module parameters become views
        self.names = list(m._names)
        params = dict(m.named_parameters())
        self.offsets, total = {}, 0
        for k in self.names:
            self.offsets[k] = total
            total += (params[k].numel() + 63) // 64 * 64                 # 256-byte aligned views
        self.numel = total
        self.w = torch.zeros(total, device=device, dtype=torch.float32)
        self.g = torch.zeros(total, device=device, dtype=torch.float32)
        for k in self.names:
            p, o = params[k], self.offsets[k]
            view = self.w[o:o + p.numel()].view(p.shape)
            view.copy_(p.detach().to(device=device, dtype=torch.float32))
            p.data = view
            ctx.W[k] = view
            ctx.G[k] = self.g[o:o + p.numel()].view(p.shape)
        m._dirty, m._device = True, None
        # ---- graph
        dim, mults = self.dim, self.mults
        dims = [dim] + [dim * mm for mm in mults]
        self.in_out = list(zip(dims[:-1], dims[1:]))
        n = len(self.in_out)
        self.init_attn = _TAttn(ctx, "init_temporal_attn", dim, self)
        self.downs, self.ups = [], []
        for i, (di, do) in enumerate(self.in_out):
            p = f"downs.{i}"
            self.downs.append((_Res(ctx, p + ".0", di, 0, do, True), _Res(ctx, p + ".1", do, 0, do, True), _SAttn(ctx, p + ".2", do),
                               _TAttn(ctx, p + ".3", do, self), _Down(ctx, p + ".4") if i < n - 1 else None))
        mid = dims[-1]
        self.mid1 = _Res(ctx, "mid_block1", mid, 0, mid, True)
        self.mid_s = _MAttn(ctx, "mid_spatial_attn", mid)
        self.mid_t = _TAttn(ctx, "mid_temporal_attn", mid, self)
        self.mid2 = _Res(ctx, "mid_block2", mid, 0, mid, True)
        for i, (di, do) in enumerate(reversed(self.in_out)):
            p = f"ups.{i}"
            self.ups.append((_Res(ctx, p + ".0", do, do, di, True), _Res(ctx, p + ".1", di, 0, di, True), _SAttn(ctx, p + ".2", di),
                             _TAttn(ctx, p + ".3", di, self), _Up(ctx, p + ".4") if i < n - 1 else None))
        self.final_res = _Res(ctx, "final_conv.0", dim, dim, dim, False)
        self.final_f = _Pack(ctx.fwd_mode)
        self.final_d = _Pack(ctx.bwd_mode)
        self.stem = C.c_void_p()
        self.cpad = 4 if self.channels <= 4 else 8
        assert self.channels <= 8, "stem weight gradient: at most 8 input channels"
        half = dim // 2
        self.freqs = torch.exp(torch.arange(half, device=device) * -(math.log(10000) / (half - 1))).float()
        self._frames = None
        self._packed_version = None
        self.version = 0                       # bumped by whoever changes `w` (optimizer step, load)
        self.opad = (self.out_dim + 3) // 4 * 4

    # ------------------------------------------------------------------ packing

    def set_loss_scale(self, loss_scale):
        """Power-of-two factor on d loss / d eps (undone inside the optimizer kernel).  It may change between steps (the Trainer's
        dynamic scaling): nothing on the device depends on it except the two scalars below."""
        self.loss_scale = float(loss_scale)
        assert self.loss_scale > 0 and math.frexp(self.loss_scale)[0] == 0.5, "loss_scale must be a power of two"
        # Operand scale of the f16x3 weight-gradient kernel's dy (wgrad3 splits dy * wgrad_dy_scale, saturating at 65504).
        # * backward-data in f16x3 (the Trainer's default, normally under the DYNAMIC loss scale): 2^4, the operand pre-scale of the
        #   backward-data convolutions -- ONE window |loss_scale * d loss / d conv output| <= 4094 for both kernels, so that a halved
        #   loss scale widens wgrad3's window as well.  (r04 used max(2^4, 2^24 / loss_scale): below 2^20 the split operand was
        #   d loss * 2^24 whatever the scale, an overflow at 2^20 -- the scaler's own start -- tripped the sentinel at every
        #   smaller scale too and the run ended at scale 1 with FloatingPointError: ADVICE r04.)
        # * exact backward-data products (x6 / f32: a fixed loss scale, 1 by default): 2^24 / loss_scale, which puts
        #   d eps ~ 2 (eps - noise) / numel ~ 1e-7 at O(1) in front of the split.
        if self.wgrad_mode != "f16x3":
            self.ctx.wgrad_dy_scale = 0.0
        elif self.ctx.bwd_mode == "f16x3":
            self.ctx.wgrad_dy_scale = 16.0
        else:
            self.ctx.wgrad_dy_scale = max(16.0, (2.0 ** 24) / self.loss_scale)

    def _blocks(self):
        yield self.init_attn
        for lv in self.downs + self.ups:
            for b in lv:
                if b is not None:
                    yield b
        yield from (self.mid1, self.mid_s, self.mid_t, self.mid2, self.final_res)

    def repack(self):
        """Re-derive every packed / transposed operand from the flat weights (once per optimizer step)."""
        L, ctx = _lib.lib(), self.ctx
        for blk in self._blocks():
            for c in blk.convs():
                c.repack()
        wf = _as5(ctx.W["final_conv.1.weight"])                          # [out_dim, dim, 1, 1, 1]
        self.final_f.pack(wf, (1, 1, 1, 1, 1, 0, 0, 0))
        wt = torch.zeros(self.dim, self.opad, 1, 1, 1, device=self.device)
        wt[:, :self.out_dim] = wf.transpose(0, 1)
        self.final_d.pack(wt, (1, 1, 1, 1, 1, 0, 0, 0))
        ws = ctx.W["init_conv.weight"]
        _lib.check(L.dpc_stem_pack(_lib.ptr(ws.contiguous()), ws.shape[0], ws.shape[1], ws.shape[2], ctx.fwd_mode.encode(),
                                   C.byref(self.stem), _lib.stream()))
        self._packed_version = self.version

    def _tables(self, frames):
        if self._frames != frames:
            self.bucket = _relative_position_bucket(frames).to(self.device)                       # [F, F] int64
            cos, sin = _rotary_tables(frames, 32)
            self.rot_cos, self.rot_sin = cos.to(self.device).contiguous(), sin.to(self.device).contiguous()
            self.bucket_mask = torch.nn.functional.one_hot(self.bucket, 32).permute(2, 0, 1).float().contiguous()   # [32, F, F]
            self._frames = frames
        emb = self.ctx.W["time_rel_pos_bias.relative_attention_bias.weight"]                      # [32, heads]
        self.pos_bias = emb[self.bucket].permute(2, 0, 1).contiguous()                            # [heads, F, F] (:106-112)
        self.d_pos_bias = torch.zeros_like(self.pos_bias)

    # ------------------------------------------------------------------ forward with tape
    def forward(self, x, time):
        """x [B, F, C, H, W] contiguous fp32 on the device, time int64 [B] -> eps [B, F, out_dim, H, W]; keeps the tape."""
        ctx, L = self.ctx, _lib.lib()
        B, F, Cc, H, W = x.shape
        assert Cc == self.channels and x.is_contiguous() and x.dtype == torch.float32
        if self._packed_version != self.version:
            self.repack()
        self._tables(F)
        dims = (B, F, H, W)
        # time conditioning (:404-409): SinusoidalPosEmb -> Linear -> GELU -> Linear
        ang = time.to(self.device).float()[:, None] * self.freqs[None, :]
        emb = torch.cat((ang.sin(), ang.cos()), dim=-1).contiguous()
        h1 = ctx.linear(emb, "time_mlp.1.weight", "time_mlp.1.bias")
        t = ctx.linear(h1, "time_mlp.3.weight", "time_mlp.3.bias", in_act=2)
        self.ttape = (emb, h1, t)
        # stem + init temporal attention
        x0 = ctx.empty(B * F * H * W, self.dim)
        _lib.check(L.dpc_stem_run(self.stem, _lib.ptr(x), Cc, 0, _lib.ptr(ctx.W["init_conv.bias"]), _lib.ptr(x0), B, F, H, W, _lib.stream()))
        self.in_tape = (x, dims)
        h = self.init_attn.forward(x0, dims)
        r = h
        skips = []
        for b1, b2, sa, ta, down in self.downs:
            h = b1.forward(h, None, t, dims)
            h = b2.forward(h, None, t, dims)
            h = sa.forward(h, dims)
            h = ta.forward(h, dims)
            skips.append(h)
            if down is not None:
                h = down.forward(h, dims)
                dims = (B, F, dims[2] // 2, dims[3] // 2)
        h = self.mid1.forward(h, None, t, dims)
        h = self.mid_s.forward(h, dims)
        h = self.mid_t.forward(h, dims)
        h = self.mid2.forward(h, None, t, dims)
        for b1, b2, sa, ta, up in self.ups:
            h = b1.forward(h, skips.pop(), t, dims)
            h = b2.forward(h, None, t, dims)
            h = sa.forward(h, dims)
            h = ta.forward(h, dims)
            if up is not None:
                h = up.forward(h, dims)
                dims = (B, F, dims[2] * 2, dims[3] * 2)
        y = self.final_res.forward(h, r, None, dims)
        out = torch.empty(B, F, self.out_dim, H, W, device=self.device, dtype=torch.float32)
        self.final_f.run(y, B, F, H, W, bias=ctx.W["final_conv.1.bias"], out=out, out_mode=1)
        self.ftape = (y, dims)
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, dout):
        """dout [B, F, out_dim, H, W] = d loss / d eps (times loss_scale).  Fills the flat gradient buffer `g`."""
        ctx, L = self.ctx, _lib.lib()
        y, dims = self.ftape
        self.ftape = None
        B, F, H, W = dims
        rows = B * F * H * W
        emb, h1, t = self.ttape
        self.ttape = None
        dt = torch.zeros_like(t)
        # final 1x1x1 conv: gradient to channels-last rows, padded to `opad` channels
        dy = ctx.empty(rows, self.opad)
        _lib.check(L.dpc_nchw_to_cl(_lib.ptr(dout.contiguous()), _lib.ptr(dy), B * F, self.out_dim, self.opad, H * W, _lib.stream()))
        dwf = ctx.empty(self.opad, self.dim)
        ctx.wgrad(y, dy, None, (1, 1, 1, 1, 1, 0, 0, 0), B, F, H, W, H, W, dw=dwf)
        ctx.G["final_conv.1.weight"].copy_(dwf[:self.out_dim].view_as(ctx.G["final_conv.1.weight"]))
        dbf = ctx.empty(self.opad)
        ctx.colsum(dy, dbf)
        ctx.G["final_conv.1.bias"].copy_(dbf[:self.out_dim])
        d = self.final_d.run(dy, B, F, H, W, act_scale=ctx.act_scale)
        del dy, y
        d, d_r = self.final_res.backward(d, dt)
        d_skip = []
        for b1, b2, sa, ta, up in reversed(self.ups):
            if up is not None:
                d = up.backward(d)
            d = ta.backward(d)
            d = sa.backward(d)
            d, _ = b2.backward(d, dt)
            d, ds = b1.backward(d, dt)
            d_skip.append(ds)                      # skip gradients in the order level 0, 1, ... (the LAST up level took skip 0)
        d, _ = self.mid2.backward(d, dt)
        d = self.mid_t.backward(d)
        d = self.mid_s.backward(d)
        d, _ = self.mid1.backward(d, dt)
        for b1, b2, sa, ta, down in reversed(self.downs):
            if down is not None:
                d = down.backward(d)
            ctx.add_(d, d_skip.pop())
            d = ta.backward(d)
            d = sa.backward(d)
            d, _ = b2.backward(d, dt)
            d, _ = b1.backward(d, dt)
        ctx.add_(d, d_r)
        del d_r
        d = self.init_attn.backward(d)
        # stem: weight gradient from the channels-last (zero-padded) input, bias gradient
        x, (B, F, H, W) = self.in_tape
        self.in_tape = None
        xc = ctx.empty(rows, self.cpad)
        _lib.check(L.dpc_nchw_to_cl(_lib.ptr(x), _lib.ptr(xc), B * F, self.channels, self.cpad, H * W, _lib.stream()))
        k = self.module.init_kernel_size
        ctx.wgrad(xc, d, "init_conv.weight", (k, k, k, 1, 1, k // 2, k // 2, k // 2), B, F, H, W, H, W, c_valid=self.channels,
                  ctot=self.channels)
        ctx.colsum(d, ctx.G["init_conv.bias"])
        del d, xc
        # time MLP (dt = gradient w.r.t. t, the input of every block's SiLU -> Linear)
        dh1 = torch.empty_like(h1)
        ctx.linear_bwd(dt, h1, "time_mlp.3.weight", "time_mlp.3.bias", 2, dx=dh1)
        ctx.linear_bwd(dh1, emb, "time_mlp.1.weight", "time_mlp.1.bias", 0)
        # relative position bias: Embedding(32, heads) gathered by the bucket table (:106-112); fixed-order masked sums
        ctx.G["time_rel_pos_bias.relative_attention_bias.weight"].copy_(
            (self.bucket_mask[:, None] * self.d_pos_bias[None]).sum(dim=(2, 3)))

    def p_losses(self, x0, t, noise, sqrt_ac, sqrt_1mac, channel_offset=0):
        """GaussianDiffusion.p_losses (diffusion_2d_smoke.py:809-831) + backward: returns the loss (device scalar tensor);
        `g` then holds loss_scale * d loss / d parameter.  x0 [B, F, Ctot, H, W]: the model's channels are
        [channel_offset, channel_offset + channels) (Trainer.train :1018-1019)."""
        L, ctx = _lib.lib(), self.ctx
        B, F, Ctot, H, W = x0.shape
        Cc = self.channels
        state = torch.empty(B, F, Cc, H, W, device=self.device, dtype=torch.float32)
        target = torch.empty_like(state)
        _lib.check(L.dpc_q_sample_smoke(_lib.ptr(x0), Ctot, channel_offset, _lib.ptr(noise), _lib.ptr(t, torch.long), _lib.ptr(sqrt_ac),
                                        _lib.ptr(sqrt_1mac), _lib.ptr(state), _lib.ptr(target), B, F, Cc, H, W, _lib.stream()))
        out = self.forward(state, t)
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
        dout = torch.empty_like(out)
        p, n = ctx.ws(L.dpc_reduce_workspace_bytes())
        _lib.check(L.dpc_mse_loss_grad(_lib.ptr(out), _lib.ptr(target), _lib.ptr(dout), _lib.ptr(loss), out.numel(), self.loss_scale, p, n,
                                       _lib.stream()))
        del out, target
        self.backward(dout)
        return loss

    def grads(self):
        """{name: gradient view} (scaled by loss_scale)."""
        return dict(self.ctx.G)

    def __del__(self):
        h, self.stem = getattr(self, "stem", None), None
        L = getattr(_lib, "_lib", None) if _lib is not None else None
        if h and L is not None:
            L.dpc_stem_free(h)
