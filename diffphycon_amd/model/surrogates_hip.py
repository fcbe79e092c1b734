"""The jellyfish guidance surrogates on libdpc: forward AND input-gradient backward of the boundary updater `Unet`
(/root/reference/diffusion/diffusion_2d_jellyfish.py:276-403) and of `ForceUnet` (:406-481), and the design gradient
`force_fn` (/root/reference/inference/inference_2d_jellyfish.py:85-114) assembled from them -- no autograd graph, no
torch convolution: every tensor-sized operation is a HIP kernel behind include/dpc.h's operator set (csrc/surr.hip).

What stays in torch is [N, C]-sized bookkeeping (N = batch x frames images): the sinusoidal embedding of theta, the
derivative of the time MLP's activations, the regulariser on theta.  Activations are channels-last fp32
[N * H * W, C]; the backward pass walks an explicit tape of the raw (pre-normalisation) conv outputs and statistics.

Arithmetic: the convolutions run the split-operand implicit GEMM in its exact-product mode ("x6": bf16x6, fp32-equivalent
products) -- the design gradient spans 1e-7 .. 1e-4, which the fp16-based default mode would flush; attention cores,
norms and their backward kernels are fp32 / fp32-MFMA with fp64 reductions.

Weights are transformed ONCE at construction (weight standardisation :107-120, flipped / transposed copies for the
backward-data convolutions, per-source slices for the concatenated inputs, the pixel-unshuffle :101-105 turned into a
2 x 2 stride-2 convolution, the 1 / (h w) of LinearAttention :241 folded into to_out)."""
import ctypes as C
import math
import os

import torch

from .. import _lib

# arithmetic of the convolutions: "f16x3" (default; 22-bit operands, 3 MFMAs per product, the 3x3 convs on the LDS halo-tile
# kernel), "x6" (bf16x6: fp32-equivalent products, 6 MFMAs, implicit GEMM only) or "f32"
_MODE = os.environ.get("DPC_SURROGATE_MODE", "f16x3")
_FUSED_GN = os.environ.get("DPC_SURROGATE_FUSED_GN", "1") != "0"       # A/B: 0 keeps the standalone GroupNorm statistics / apply passes


class _Calibration:
    """Range calibration of the backward pass.  The f16x3 operand split represents an element to 2^-22 of ITS OWN magnitude while
    |x * scale| lies in [2^-3, 65504] and to an absolute 2^-25 / scale below (fp16 subnormals of the remainder plane): with the
    tensor's maximum placed at `target` = 16 the floor is 2e-9 of that maximum and the clamp 4 000 x above it.  Forward activations
    sit in the window by construction (every conv input is a GroupNorm / LayerNorm / softmax output); gradients do not -- through
    the two nets the per-tensor maximum spans 1e-9 .. 1e-1 -- so every backward-data convolution carries a per-convolution
    power-of-two operand scale (dpc_conv_run's act_scale, undone in its epilogue; a power of two moves no mantissa bit).

    r04: the scales are fixed ONCE, before the first design-gradient call, on a SYNTHETIC seeded input of one trajectory (the
    activations of both nets are normalised, so the gradient magnitudes depend on the weights and the geometry, hardly on the data:
    profiles/r04_calibration_drift.log) -- a set-up step like the weight packing, with its host reads outside the sampling loop.
    Every rank computes the same scales from the same seeded input: nothing is exchanged, and a trajectory's bits do not depend on how
    the batch is sharded or chunked.  (r03 re-measured max |input| on the live batch every 64 calls: one host sync and one
    all-reduce(MAX) per convolution inside the loop.)  Inside the loop the window is only WATCHED, on the device: every
    DPC_SURROGATE_RANGE_CHECK_EVERY calls (default 64) each backward convolution folds max |input| into a device-resident running peak
    (one streaming pass, no sync); `HipDesignGradient.check_range()` -- called once at the end of `sample()` -- reads the peaks and
    raises if one reached the clamp."""
    target = 16.0
    active = False      # calibrating: measure, read back, set the scale (set-up only)
    watch = False       # inside the loop: fold max |input| into the convolution's device-resident peak
    seen = []           # (max |input|, scale) per calibrated convolution of the last calibration pass
    convs = None        # weak set of the dynamic convolutions (check_range walks it)


def _f(t):
    return t.detach().to(torch.float32).contiguous()


class _Conv:
    """dpc_conv_t owner.  `w` [N, K, kh, kw] on the device.  dynamic: a backward-data convolution (calibrated operand scale)."""

    def __init__(self, w, sh=1, sw=1, ph=None, pw=None, taps=None, mode=None, dynamic=False):
        w = _f(w)
        self.dynamic = dynamic and (mode or _MODE) == "f16x3"
        self.act_scale = 0.0
        if self.dynamic:
            import weakref
            if _Calibration.convs is None:
                _Calibration.convs = weakref.WeakSet()
            _Calibration.convs.add(self)
        self.N, self.K, self.kh, self.kw = w.shape
        self.sh, self.sw = sh, sw
        ph = (self.kh - 1) // 2 if ph is None else ph
        pw = (self.kw - 1) // 2 if pw is None else pw
        t0, t1 = taps if taps is not None else (0, 0)
        h = C.c_void_p()
        _lib.check(_lib.lib().dpc_conv_pack(_lib.ptr(w), self.N, self.K, self.kh, self.kw, sh, sw, ph, pw, t0, t1,
                                            (mode or _MODE).encode(), C.byref(h), _lib.stream()))
        torch.cuda.current_stream().synchronize()          # the pack kernels read `w`, which dies with this frame
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        L = getattr(_lib, "_lib", None) if _lib is not None else None      # (module globals are gone at interpreter exit)
        if h and L is not None:
            L.dpc_conv_free(h)

    def __call__(self, a0, images, Hi, Wi, a1=None, bias=None, resid=None, out=None, Ho=None, Wo=None, ln=None, out_mode=0,
                 par=(0, 0), C0=None, a0_stride=0):
        Ho = Hi if Ho is None else Ho
        Wo = Wi if Wo is None else Wo
        C0 = a0.shape[-1] if C0 is None else C0
        C1 = a1.shape[-1] if a1 is not None else 0
        if self.dynamic and (_Calibration.active or _Calibration.watch):
            m = torch.zeros(1, device=a0.device)
            _lib.check(_lib.lib().dpc_absmax(_lib.ptr(a0), a0.numel(), _lib.ptr(m), _lib.stream()))
            if _Calibration.active:                                  # set-up pass on the synthetic input: host read allowed
                m = float(m.item())
                if not math.isfinite(m):
                    raise RuntimeError("design gradient: a backward tensor of the surrogate nets is not finite (Inf / NaN reached "
                                       "the range calibration of a backward-data convolution)")
                if m > 0:
                    self.act_scale = 2.0 ** max(-100, min(100, round(math.log2(_Calibration.target / m))))
                _Calibration.seen.append((m, self.act_scale))
                self.peak = None
            else:                                                    # in the loop: device-side running maximum, no sync
                self.peak = m if getattr(self, "peak", None) is None else torch.maximum(self.peak, m)
        if out is None:
            rows = images * (Ho * Wo if out_mode != 2 else 4 * Ho * Wo)
            out = torch.empty(rows, self.N, device=a0.device, dtype=torch.float32)
        _lib.check(_lib.lib().dpc_conv_run(self.h, _lib.ptr(a0), _lib.ptr(a1), C0, C1, _lib.ptr(bias), _lib.ptr(resid), _lib.ptr(out),
                                           images, Hi, Wi, Ho, Wo, _lib.ptr(ln[0]) if ln else None, _lib.ptr(ln[1]) if ln else None,
                                           out_mode, par[0], par[1], self.act_scale, a0_stride, _lib.stream()))
        return out


def _conv_gn_fusable(conv, H, W, C0, C1=0):
    """the library's own predicate for the halo-tile path (include/dpc.h): False sends the caller to the standalone GroupNorm passes"""
    return bool(_lib.lib().dpc_conv_gn_fusable(conv.h, H, W, C0, C1))


def _conv_run_gn(conv, a0, images, H, W, a1=None, bias=None, part=None, in_coef=None):
    """dpc_conv_run_gn: the 3x3 convolution with the per-image GroupNorm hooks of the halo kernel (include/dpc.h)."""
    C0 = a0.shape[-1]
    C1 = a1.shape[-1] if a1 is not None else 0
    out = torch.empty(images * H * W, conv.N, device=a0.device, dtype=torch.float32)
    _lib.check(_lib.lib().dpc_conv_run_gn(conv.h, _lib.ptr(a0), _lib.ptr(a1), C0, C1, _lib.ptr(bias), _lib.ptr(out), images, H, W,
                                          _lib.ptr(part), _lib.ptr(in_coef), _lib.stream()))
    return out


def _DConv(w, **kw):
    """A backward-data convolution: its operand scale is calibrated (see _Calibration)."""
    return _Conv(w, dynamic=True, **kw)


def _flipT(w):
    """Weight of the backward-data convolution of a stride-1 'same' conv: [Co, Ci, kh, kw] -> [Ci, Co, kh, kw] flipped."""
    return w.flip(2, 3).transpose(0, 1).contiguous()


def _standardise(w):
    """WeightStandardizedConv2d (:107-120): per-filter zero mean / unit biased variance, eps 1e-5 (fp32)."""
    w = w.double()
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return ((w - mean) * (var + 1e-5).rsqrt()).float()


class _Ctx:
    """Shapes + scratch shared by the blocks of one net."""

    def __init__(self, device, groups):
        self.device, self.groups = device, groups
        self._gn_ws = None

    def gn_ws(self, B, Cc):
        need = _lib.lib().dpc_gn_workspace_bytes(B, Cc)
        if self._gn_ws is None or self._gn_ws.numel() < need:
            self._gn_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._gn_ws

    # ---- thin operator wrappers
    def gn_stats(self, x, B, R, Cc):
        st = torch.empty(B, self.groups, 2, device=self.device)
        ws = self.gn_ws(B, Cc)
        _lib.check(_lib.lib().dpc_gn_stats(_lib.ptr(x), _lib.ptr(st), B, R, Cc, self.groups, C.c_void_p(ws.data_ptr()), ws.numel(),
                                           _lib.stream()))
        return st

    def gn_apply(self, x, st, gamma, beta, ss, B, R, Cc, resid=None, out=None):
        out = torch.empty_like(x) if out is None else out
        _lib.check(_lib.lib().dpc_gn_apply(_lib.ptr(x), _lib.ptr(out), _lib.ptr(resid), _lib.ptr(st), _lib.ptr(gamma), _lib.ptr(beta),
                                           _lib.ptr(ss), B, R, Cc, self.groups, _lib.stream()))
        return out

    def gn_bwd(self, x, dy, st, gamma, beta, ss, B, R, Cc, want_dss):
        dx = torch.empty_like(x)
        dss = torch.empty(B, 2 * Cc, device=self.device) if want_dss else None
        ws = self.gn_ws(B, Cc)
        _lib.check(_lib.lib().dpc_gn_silu_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(st), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(ss),
                                              _lib.ptr(dx), _lib.ptr(dss), B, R, Cc, self.groups, C.c_void_p(ws.data_ptr()), ws.numel(),
                                              _lib.stream()))
        return dx, dss

    def ln_stats(self, x):
        st = torch.empty(x.shape[0], 2, device=self.device)
        _lib.check(_lib.lib().dpc_ln_stats(_lib.ptr(x), _lib.ptr(st), x.shape[0], x.shape[1], _lib.stream()))
        return st

    def ln_apply(self, x, st, g, resid):
        out = torch.empty_like(x)
        _lib.check(_lib.lib().dpc_ln_apply(_lib.ptr(x), _lib.ptr(st), _lib.ptr(g), _lib.ptr(resid), _lib.ptr(out), x.shape[0], x.shape[1],
                                           _lib.stream()))
        return out

    def ln_bwd(self, x, st, g, dy, dx=None):
        acc = 1 if dx is not None else 0
        dx = torch.empty_like(x) if dx is None else dx
        _lib.check(_lib.lib().dpc_ln_bwd(_lib.ptr(x), _lib.ptr(st), _lib.ptr(g), _lib.ptr(dy), _lib.ptr(dx), x.shape[0], x.shape[1], acc,
                                         _lib.stream()))
        return dx

    def linear(self, x, W, b=None, in_act=0, out_act=0):
        out = torch.empty(x.shape[0], W.shape[0], device=self.device)
        _lib.check(_lib.lib().dpc_small_linear(_lib.ptr(x), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), x.shape[0], W.shape[1], W.shape[0],
                                               in_act, out_act, _lib.stream()))
        return out

    def add_(self, y, x):
        _lib.check(_lib.lib().dpc_add_inplace(_lib.ptr(y), _lib.ptr(x), y.numel(), _lib.stream()))
        return y


def _silu_grad(t):
    s = torch.sigmoid(t)
    return s * (1 + t * (1 - s))


def _gelu_grad(h):
    return 0.5 * (1 + torch.erf(h * (1 / math.sqrt(2)))) + h * torch.exp(-0.5 * h * h) * (1 / math.sqrt(2 * math.pi))


class _Res:
    """ResnetBlock (:160-184) on the concatenation of (x0 [C0], x1 [C1])."""

    def __init__(self, ctx, sd, p, C0, C1, Cout, has_time):
        self.ctx, self.C0, self.C1, self.Cout = ctx, C0, C1, Cout
        w1 = _standardise(sd[p + "block1.proj.weight"])
        w2 = _standardise(sd[p + "block2.proj.weight"])
        self.c1, self.c2 = _Conv(w1), _Conv(w2)
        self.b1, self.b2 = _f(sd[p + "block1.proj.bias"]), _f(sd[p + "block2.proj.bias"])
        self.g1, self.be1 = _f(sd[p + "block1.norm.weight"]), _f(sd[p + "block1.norm.bias"])
        self.g2, self.be2 = _f(sd[p + "block2.norm.weight"]), _f(sd[p + "block2.norm.bias"])
        w1t = _flipT(w1)
        self.d2 = _DConv(_flipT(w2))
        self.d1 = [_DConv(w1t[:C0])] + ([_DConv(w1t[C0:])] if C1 else [])
        self.cr = None
        if p + "res_conv.weight" in sd:
            wr = _f(sd[p + "res_conv.weight"])
            self.cr, self.br = _Conv(wr), _f(sd[p + "res_conv.bias"])
            wrt = wr.transpose(0, 1).contiguous()
            self.dr = [_DConv(wrt[:C0])] + ([_DConv(wrt[C0:])] if C1 else [])
        else:
            assert C1 == 0 and C0 == Cout
        self.mlp = None
        if has_time and p + "mlp.1.weight" in sd:
            W = _f(sd[p + "mlp.1.weight"])
            self.mlp = (W, _f(sd[p + "mlp.1.bias"]), W.t().contiguous())

    def forward(self, x0, x1, temb, n, H, W):
        ctx, R, Cc = self.ctx, H * W, self.Cout
        ss = ctx.linear(temb, self.mlp[0], self.mlp[1], in_act=1) if (self.mlp is not None and temb is not None) else None
        if (_FUSED_GN and _conv_gn_fusable(self.c1, H, W, x0.shape[-1], x1.shape[-1] if x1 is not None else 0)
                and _conv_gn_fusable(self.c2, H, W, Cc)):
            # r05: statistics from the conv epilogues, block1's GroupNorm + (scale, shift) + SiLU inside conv2's halo load: the activated
            # tensor a1 never exists (the backward recomputes it from raw1 / st1 as before) -- 3 passes per block instead of 7
            L = _lib.lib()
            ent = L.dpc_conv_gn_entries(H, W)
            part = torch.empty(n * ent * Cc * 2, device=x0.device)
            coef = torch.empty(n * Cc * 7, device=x0.device)
            st1 = torch.empty(n, ctx.groups, 2, device=x0.device)
            st2 = torch.empty(n, ctx.groups, 2, device=x0.device)
            raw1 = _conv_run_gn(self.c1, x0, n, H, W, a1=x1, bias=self.b1, part=part)
            _lib.check(L.dpc_gn_finalize_fused(_lib.ptr(part), n, ent, Cc, ctx.groups, R, _lib.ptr(self.g1), _lib.ptr(self.be1), _lib.ptr(ss),
                                               _lib.ptr(st1), _lib.ptr(coef), _lib.stream()))
            raw2 = _conv_run_gn(self.c2, raw1, n, H, W, bias=self.b2, part=part, in_coef=coef)
            _lib.check(L.dpc_gn_finalize_fused(_lib.ptr(part), n, ent, Cc, ctx.groups, R, None, None, None, _lib.ptr(st2), None, _lib.stream()))
            res = self.cr(x0, n, H, W, a1=x1, bias=self.br) if self.cr is not None else x0
            out = ctx.gn_apply(raw2, st2, self.g2, self.be2, None, n, R, Cc, resid=res)
            self.tape = (raw1, st1, raw2, st2, ss, n, H, W)
            return out
        raw1 = self.c1(x0, n, H, W, a1=x1, bias=self.b1)
        st1 = ctx.gn_stats(raw1, n, R, Cc)
        a1 = ctx.gn_apply(raw1, st1, self.g1, self.be1, ss, n, R, Cc)
        raw2 = self.c2(a1, n, H, W, bias=self.b2)
        st2 = ctx.gn_stats(raw2, n, R, Cc)
        res = self.cr(x0, n, H, W, a1=x1, bias=self.br) if self.cr is not None else x0
        out = ctx.gn_apply(raw2, st2, self.g2, self.be2, None, n, R, Cc, resid=res, out=a1)      # a1 is dead: reuse its storage
        self.tape = (raw1, st1, raw2, st2, ss, n, H, W)
        return out

    def backward(self, dout, extra0=None, need_dx=True, need_dx1=True):
        """dout: gradient of the block output.  Returns (dx0, dx1, dss); extra0 (a gradient that reaches x0 by another path) is
        added into dx0.  need_dx False: stop after the scale/shift gradient (first block after a non-differentiated input)."""
        raw1, st1, raw2, st2, ss, n, H, W = self.tape
        self.tape = None
        ctx, R, Cc = self.ctx, H * W, self.Cout
        d_raw2, _ = ctx.gn_bwd(raw2, dout, st2, self.g2, self.be2, None, n, R, Cc, False)
        d_a1 = self.d2(d_raw2, n, H, W)
        d_raw1, dss = ctx.gn_bwd(raw1, d_a1, st1, self.g1, self.be1, ss, n, R, Cc, ss is not None)
        if not need_dx:
            return None, None, dss
        # (the 3x3 convolutions carry no residual operand: that keeps them on the halo-tile kernel; sums ride on the 1x1 convs
        #  or on one streaming add)
        dx0 = self.d1[0](d_raw1, n, H, W)
        if self.cr is None:
            ctx.add_(dx0, dout)
            if extra0 is not None:
                ctx.add_(dx0, extra0)
            return dx0, None, dss
        self.dr[0](dout, n, H, W, resid=dx0, out=dx0)
        if extra0 is not None:
            ctx.add_(dx0, extra0)
        dx1 = None
        if self.C1 and need_dx1:
            dx1 = self.d1[1](d_raw1, n, H, W)
            self.dr[1](dout, n, H, W, resid=dx1, out=dx1)
        return dx0, dx1, dss

    def dtemb(self, dss):
        """Gradient w.r.t. SiLU(t_emb) of this block's scale/shift projection (:167-171)."""
        return self.ctx.linear(dss, self.mlp[2])


class _LinAttn:
    """Residual(PreNorm(LinearAttention)) (:213-251, :186-193, :206-211)."""

    def __init__(self, ctx, sd, p, Cc, HW, heads=4):
        self.ctx, self.Cc, self.heads, self.HW = ctx, Cc, heads, HW
        self.g1 = _f(sd[p + "fn.norm.g"]).reshape(-1)
        wq = _f(sd[p + "fn.fn.to_qkv.weight"])
        wo = _f(sd[p + "fn.fn.to_out.0.weight"]) * (1.0 / HW)           # v / (h w) :241 (out is linear in v)
        # to_out carries the 1 / (h w): weights of ~1e-6 would sit in fp16's subnormals after the f16x3 split -> bf16x6 for this
        # (small: 128 -> C channels, 1x1) pair
        mo = "x6" if _MODE == "f16x3" else None
        self.cq, self.co = _Conv(wq), _Conv(wo, mode=mo)
        self.bo = _f(sd[p + "fn.fn.to_out.0.bias"])
        self.g2 = _f(sd[p + "fn.fn.to_out.1.g"]).reshape(-1)
        self.dq, self.do = _DConv(wq.transpose(0, 1).contiguous()), _DConv(wo.transpose(0, 1).contiguous(), mode=mo)

    def forward(self, x, n, H, W):
        ctx = self.ctx
        st1 = ctx.ln_stats(x)
        qkv = self.cq(x, n, H, W, ln=(st1, self.g1))
        att = torch.empty(x.shape[0], self.heads * 32, device=x.device)
        tape = torch.empty(_lib.lib().dpc_linear_attention_tape_bytes(n, self.heads), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().dpc_linear_attention_fwd_save(_lib.ptr(qkv), _lib.ptr(att), self.heads, n, H * W, C.c_void_p(tape.data_ptr()),
                                                            tape.numel(), _lib.stream()))
        o = self.co(att, n, H, W, bias=self.bo)
        st2 = ctx.ln_stats(o)
        y = ctx.ln_apply(o, st2, self.g2, x)
        self.tape = (x, st1, qkv, o, st2, tape, n, H, W)
        return y

    def backward(self, dy):
        """dy is consumed (updated in place) and returned as dx."""
        x, st1, qkv, o, st2, tape, n, H, W = self.tape
        self.tape = None
        ctx = self.ctx
        d_o = ctx.ln_bwd(o, st2, self.g2, dy)
        d_att = self.do(d_o, n, H, W)
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().dpc_linear_attention_bwd(_lib.ptr(qkv), _lib.ptr(d_att), _lib.ptr(dqkv), self.heads, n, H * W,
                                                       C.c_void_p(tape.data_ptr()), tape.numel(), _lib.stream()))
        d_xn = self.dq(dqkv, n, H, W)
        return ctx.ln_bwd(x, st1, self.g1, d_xn, dx=dy)


class _Attn:
    """Residual(PreNorm(Attention)) of the bottleneck (:253-275)."""

    def __init__(self, ctx, sd, p, Cc, heads=4):
        self.ctx, self.Cc, self.heads = ctx, Cc, heads
        self.g1 = _f(sd[p + "fn.norm.g"]).reshape(-1)
        wq, wo = _f(sd[p + "fn.fn.to_qkv.weight"]), _f(sd[p + "fn.fn.to_out.weight"])
        self.cq, self.co = _Conv(wq), _Conv(wo)
        self.bo = _f(sd[p + "fn.fn.to_out.bias"])
        self.dq, self.do = _DConv(wq.transpose(0, 1).contiguous()), _DConv(wo.transpose(0, 1).contiguous())

    def forward(self, x, n, H, W):
        ctx, L = self.ctx, H * W
        st1 = ctx.ln_stats(x)
        qkv = self.cq(x, n, H, W, ln=(st1, self.g1))
        att = torch.empty(x.shape[0], self.heads * 32, device=x.device)
        _lib.check(_lib.lib().dpc_attention_core(_lib.ptr(qkv), _lib.ptr(att), self.heads, L, n, 1, L, 0, 1, None, None, None,
                                                 _lib.stream()))
        y = self.co(att, n, H, W, bias=self.bo, resid=x)
        self.tape = (x, st1, qkv, n, H, W)
        return y

    def backward(self, dy):
        x, st1, qkv, n, H, W = self.tape
        self.tape = None
        ctx = self.ctx
        d_att = self.do(dy, n, H, W)
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().dpc_attention_bwd(_lib.ptr(qkv), _lib.ptr(d_att), _lib.ptr(dqkv), self.heads, n, H * W, _lib.stream()))
        d_xn = self.dq(dqkv, n, H, W)
        return ctx.ln_bwd(x, st1, self.g1, d_xn, dx=dy)


class _Down:
    """Downsample (:95-105): pixel-unshuffle + 1x1 conv == a 2 x 2 stride-2 convolution on re-indexed weights."""

    def __init__(self, sd, p, Cin, Cout):
        w = _f(sd[p + "1.weight"]).reshape(Cout, Cin, 2, 2)
        self.c = _Conv(w, sh=2, sw=2, ph=0, pw=0)
        self.b = _f(sd[p + "1.bias"])
        self.d = [[_DConv(w[:, :, a, b].t().contiguous()[:, :, None, None]) for b in range(2)] for a in range(2)]
        self.Cin = Cin

    def forward(self, x, n, H, W):
        self.shape = (n, H, W)
        return self.c(x, n, H, W, bias=self.b, Ho=H // 2, Wo=W // 2), H // 2, W // 2

    def backward(self, dy):
        n, H, W = self.shape
        dx = torch.empty(n * H * W, self.Cin, device=dy.device)
        for a in range(2):
            for b in range(2):
                self.d[a][b](dy, n, H // 2, W // 2, out=dx, out_mode=2, par=(a, b))
        return dx


class _Conv3:
    """plain 3x3 conv with bias (last encoder / decoder level :335, :352)."""

    def __init__(self, sd, p):
        w = _f(sd[p + "weight"])
        self.c, self.b, self.d = _Conv(w), _f(sd[p + "bias"]), _DConv(_flipT(w))

    def forward(self, x, n, H, W):
        self.shape = (n, H, W)
        return self.c(x, n, H, W, bias=self.b), H, W

    def backward(self, dy):
        n, H, W = self.shape
        return self.d(dy, n, H, W)


class _Up:
    """Upsample (:89-93): nearest x2 + 3x3 conv."""

    def __init__(self, sd, p):
        w = _f(sd[p + "1.weight"])
        self.c, self.b, self.d = _Conv(w), _f(sd[p + "1.bias"]), _DConv(_flipT(w))

    def forward(self, x, n, H, W):
        self.shape = (n, H, W)
        Cc = x.shape[1]
        up = torch.empty(n * 4 * H * W, Cc, device=x.device)
        _lib.check(_lib.lib().dpc_upsample2x_cl(_lib.ptr(x), _lib.ptr(up), n, H, W, Cc, _lib.stream()))
        return self.c(up, n, 2 * H, 2 * W, bias=self.b), 2 * H, 2 * W

    def backward(self, dy):
        n, H, W = self.shape
        d_up = self.d(dy, n, 2 * H, 2 * W)
        Cc = d_up.shape[1]
        dx = torch.empty(n * H * W, Cc, device=dy.device)
        _lib.check(_lib.lib().dpc_downsum2x_cl(_lib.ptr(d_up), _lib.ptr(dx), n, H, W, Cc, _lib.stream()))
        return dx


class _Init7:
    """init_conv (:296, :427), 7x7 on 3 / 4 channels.  As 49 taps of a 4-channel implicit GEMM it would waste 7/8 of every
    32-wide reduction chunk; instead the 7 horizontally adjacent pixels x 4 channels = 28 contiguous floats of a width-padded
    channels-last image are ONE reduction row (a0_stride), leaving 7 vertical taps.  The backward-data product runs the same way
    transposed: 7 vertical taps from 64 channels to 7 x 4 horizontal partial sums per pixel, folded along w by one small kernel."""

    def __init__(self, sd, need_bwd):
        w = _f(sd["init_conv.weight"])
        Co, self.Cin = w.shape[0], w.shape[1]
        assert self.Cin <= 4 and tuple(w.shape[2:]) == (7, 7)
        if self.Cin != 4:
            w = torch.cat((w, w.new_zeros(Co, 4 - self.Cin, 7, 7)), dim=1)
        self.c = _Conv(w.permute(0, 3, 1, 2).reshape(Co, 28, 7, 1), ph=3, pw=0)            # [co][b * 4 + c][a]
        self.bias = _f(sd["init_conv.bias"])
        if need_bwd:
            self.d = _DConv(w.flip(2).permute(3, 1, 0, 2).reshape(28, Co, 7, 1), ph=3, pw=0)   # [b * 4 + c][co][6 - a]

    def forward(self, x, n, H, W):
        xp = torch.empty(n * H * (W + 6), 4, device=x.device)
        _lib.check(_lib.lib().dpc_pad_w_cl(_lib.ptr(x), _lib.ptr(xp), n * H, W, 4, 3, _lib.stream()))
        return self.c(xp, n, H, W + 6, bias=self.bias, Ho=H, Wo=W, C0=28, a0_stride=4)

    def backward(self, dy, n, H, W):
        part = self.d(dy, n, H, W)                                                         # [rows][7 * 4]
        dx = torch.empty(n * H * W, 4, device=dy.device)
        _lib.check(_lib.lib().dpc_fold_w_cl(_lib.ptr(part), _lib.ptr(dx), n * H, W, 4, 7, _lib.stream()))
        return dx


def _encoder(ctx, sd, dims, has_time, H):
    """`downs` of both nets (:313-327, :433-447): (block1, block2, attn, down) per level."""
    levels = []
    n_lv = len(dims) - 1
    for i in range(n_lv):
        di, do = dims[i], dims[i + 1]
        p = f"downs.{i}."
        last = i == n_lv - 1
        levels.append((_Res(ctx, sd, p + "0.", di, 0, di, has_time), _Res(ctx, sd, p + "1.", di, 0, di, has_time),
                       _LinAttn(ctx, sd, p + "2.", di, H * H), _Conv3(sd, p + "3.") if last else _Down(sd, p + "3.", di, do)))
        if not last:
            H //= 2
    return levels, H


class HipForceUnet:
    """ForceUnet (:406-481): [N, 4, H, W] -> [N, out_dim]; backward: d force -> d input."""

    def __init__(self, module, image_size):
        sd = {k: v.detach() for k, v in module.state_dict().items()}
        dev = next(module.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("HipForceUnet needs the module on the GPU (libdpc has no CPU path)")
        self.device, self.H = dev, image_size
        dim = sd["init_conv.weight"].shape[0]
        groups = module.downs[0][0].block1.norm.num_groups
        self.ctx = _Ctx(dev, groups)
        self.init = _Init7(sd, need_bwd=True)
        dims = [dim] + [sd[f"downs.{i}.3.weight" if f"downs.{i}.3.weight" in sd else f"downs.{i}.3.1.weight"].shape[0]
                        for i in range(len(module.downs))]
        self.levels, self.Hmid = _encoder(self.ctx, sd, dims, False, image_size)
        mid = dims[-1]
        self.mid1 = _Res(self.ctx, sd, "mid_block1.", mid, 0, mid, False)
        self.mid_attn = _Attn(self.ctx, sd, "mid_attn.", mid)
        self.mid2 = _Res(self.ctx, sd, "mid_block2.", mid, 0, mid, False)
        self.Wf, self.bf = _f(sd["final.weight"]), _f(sd["final.bias"])
        self.mid = mid
        if self.Wf.shape[1] != mid:
            # the reference hard-codes `self.final = nn.Linear(512, out_dim)` (diffusion_2d_jellyfish.py:454) and fails at
            # `self.final(x)` for any width whose bottleneck is not 512 channels (dim != 64): same error, up front
            raise ValueError(f"ForceUnet: final Linear expects {self.Wf.shape[1]} features but the bottleneck has {mid} channels "
                             f"(dim = {dim}, dim_mults x8): the reference's ForceUnet only works for dim = 64")

    def forward_cl(self, x, n):
        """x: channels-last [n * H * W, 4].  Returns force [n, out_dim]."""
        H = self.H
        x = self.init.forward(x, n, H, H)
        for b1, b2, attn, down in self.levels:
            x = b1.forward(x, None, None, n, H, H)
            x = b2.forward(x, None, None, n, H, H)
            x = attn.forward(x, n, H, H)
            x, H, _ = down.forward(x, n, H, H)
        x = self.mid1.forward(x, None, None, n, H, H)
        x = self.mid_attn.forward(x, n, H, H)
        x = self.mid2.forward(x, None, None, n, H, H)
        feat = torch.empty(n, self.mid, device=self.device)
        _lib.check(_lib.lib().dpc_mean_rows(_lib.ptr(x), _lib.ptr(feat), n, H * H, self.mid, _lib.stream()))
        self.n = n
        return self.ctx.linear(feat, self.Wf, self.bf)

    def __call__(self, x_nchw):
        n, Cc, H, W = x_nchw.shape
        x = torch.empty(n * H * W, 4, device=self.device)
        _lib.check(_lib.lib().dpc_nchw_to_cl(_lib.ptr(_f(x_nchw)), _lib.ptr(x), n, Cc, 4, H * W, _lib.stream()))
        return self.forward_cl(x, n)

    def backward(self, dforce):
        """dforce [n, out_dim] -> gradient of the channels-last input [n * H * W, 4]."""
        n, H = self.n, self.Hmid
        dfeat = (_f(dforce) @ self.Wf).contiguous()                      # [n, 512]: per-image vector
        dx = torch.empty(n * H * H, self.mid, device=self.device)
        _lib.check(_lib.lib().dpc_bcast_rows(_lib.ptr(dfeat), _lib.ptr(dx), n, H * H, self.mid, 1.0 / (H * H), _lib.stream()))
        dx, _, _ = self.mid2.backward(dx)
        dx = self.mid_attn.backward(dx)
        dx, _, _ = self.mid1.backward(dx)
        for b1, b2, attn, down in reversed(self.levels):
            dx = down.backward(dx)
            dx = attn.backward(dx)
            dx, _, _ = b2.backward(dx)
            dx, _, _ = b1.backward(dx)
        return self.init.backward(dx, n, self.H, self.H)


class HipUnet:
    """Boundary updater `Unet` (:276-403): (bd_0 [N, 3, H, W], theta [N]) -> bd [N, 3, H, W]; backward: d bd -> d theta."""

    def __init__(self, module, image_size):
        sd = {k: v.detach() for k, v in module.state_dict().items()}
        dev = next(module.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("HipUnet needs the module on the GPU (libdpc has no CPU path)")
        self.device, self.H = dev, image_size
        dim = sd["init_conv.weight"].shape[0]
        self.dim = dim
        groups = module.downs[0][0].block1.norm.num_groups
        self.ctx = _Ctx(dev, groups)
        self.init = _Init7(sd, need_bwd=False)
        n_lv = len(module.downs)
        dims = [dim] + [sd[f"downs.{i}.3.weight" if f"downs.{i}.3.weight" in sd else f"downs.{i}.3.1.weight"].shape[0] for i in range(n_lv)]
        self.levels, self.Hmid = _encoder(self.ctx, sd, dims, True, image_size)
        mid = dims[-1]
        self.mid1 = _Res(self.ctx, sd, "mid_block1.", mid, 0, mid, True)
        self.mid_attn = _Attn(self.ctx, sd, "mid_attn.", mid)
        self.mid2 = _Res(self.ctx, sd, "mid_block2.", mid, 0, mid, True)
        self.ups = []
        H = self.Hmid
        for i in range(n_lv):
            di, do = dims[n_lv - 1 - i], dims[n_lv - i]
            p = f"ups.{i}."
            last = i == n_lv - 1
            self.ups.append((_Res(self.ctx, sd, p + "0.", do, di, do, True), _Res(self.ctx, sd, p + "1.", do, di, do, True),
                             _LinAttn(self.ctx, sd, p + "2.", do, H * H), _Conv3(sd, p + "3.") if last else _Up(sd, p + "3.")))
            if not last:
                H *= 2
        self.final_res = _Res(self.ctx, sd, "final_res_block.", dim, dim, dim, True)
        wf, bf = _f(sd["final_conv.weight"]), _f(sd["final_conv.bias"])
        self.out_dim = wf.shape[0]
        self.final_nchw, self.bf = _Conv(wf), bf
        # force_fn variant: output channel 0 is left for the pressure (zero weights), channels 1.. carry the boundary
        wf4 = torch.cat((wf.new_zeros(1, dim, 1, 1), wf), dim=0)
        self.final_cl, self.bf4 = _Conv(wf4), torch.cat((bf.new_zeros(1), bf))
        self.dfinal = _DConv(wf4.transpose(0, 1).contiguous())
        W1, W2 = _f(sd["time_mlp.1.weight"]), _f(sd["time_mlp.3.weight"])
        self.t1 = (W1, _f(sd["time_mlp.1.bias"]), W1.t().contiguous())
        self.t2 = (W2, _f(sd["time_mlp.3.bias"]), W2.t().contiguous())
        half = dim // 2
        self.freqs = torch.exp(torch.arange(half, device=dev) * -(math.log(10000) / (half - 1)))

    def _trunk(self, x, theta, n):
        """x channels-last [n * H * W, 4] (3 boundary channels + a zero pad).  Returns the final_res_block output."""
        H = self.H
        ctx = self.ctx
        theta = _f(theta).reshape(-1)
        ang = theta[:, None] * self.freqs[None, :]
        emb = torch.cat((ang.sin(), ang.cos()), dim=-1).contiguous()           # SinusoidalPosEmb :122-135
        h1 = ctx.linear(emb, self.t1[0], self.t1[1])
        t = ctx.linear(h1, self.t2[0], self.t2[1], in_act=2)                   # Linear(GELU(h1))
        self.ttape = (ang, h1, t)
        x = self.init.forward(x, n, H, H)
        r = x
        hs = []
        for b1, b2, attn, down in self.levels:
            x = b1.forward(x, None, t, n, H, H)
            hs.append(x)
            x = b2.forward(x, None, t, n, H, H)
            x = attn.forward(x, n, H, H)
            hs.append(x)
            x, H, _ = down.forward(x, n, H, H)
        x = self.mid1.forward(x, None, t, n, H, H)
        x = self.mid_attn.forward(x, n, H, H)
        x = self.mid2.forward(x, None, t, n, H, H)
        for b1, b2, attn, up in self.ups:
            x = b1.forward(x, hs.pop(), t, n, H, H)
            x = b2.forward(x, hs.pop(), t, n, H, H)
            x = attn.forward(x, n, H, H)
            x, H, _ = up.forward(x, n, H, H)
        self.n = n
        return self.final_res.forward(x, r, t, n, H, H)

    def _input_cl(self, bd0):
        n, Cc, H, W = bd0.shape
        x = torch.empty(n * H * W, 4, device=self.device)
        _lib.check(_lib.lib().dpc_nchw_to_cl(_lib.ptr(_f(bd0)), _lib.ptr(x), n, Cc, 4, H * W, _lib.stream()))
        return x, n

    def __call__(self, bd0, theta):
        """The reference's forward: [N, 3, H, W] out (channels-first, written by the final conv's epilogue)."""
        x, n = self._input_cl(bd0)
        y = self._trunk(x, theta, n)
        out = torch.empty(n, self.out_dim, self.H, self.H, device=self.device)
        self.final_nchw(y, n, self.H, self.H, bias=self.bf, out=out, out_mode=1)
        self._drop_tape()
        return out

    def forward_cl4(self, bd0, theta):
        """force_fn's variant: channels-last [n * H * W, 4] with channel 0 = 0 (the caller writes the pressure there)."""
        x, n = self._input_cl(bd0)
        y = self._trunk(x, theta, n)
        return self.final_cl(y, n, self.H, self.H, bias=self.bf4)

    def _drop_tape(self):
        for blk in self._blocks():
            blk.tape = None
        self.ttape = None

    def _blocks(self):
        for lv in self.levels + self.ups:
            yield from lv[:3]
        yield from (self.mid1, self.mid_attn, self.mid2, self.final_res)

    def backward_theta(self, d_out_cl4):
        """Gradient of theta [n] given the gradient of forward_cl4's output (channel 0 is ignored)."""
        n, H = self.n, self.H
        dts = torch.zeros(n, self.t2[0].shape[0], device=self.device)          # d SiLU(t), summed over the blocks

        def acc(blk, dss):
            dts.add_(blk.dtemb(dss))

        dy = self.dfinal(d_out_cl4, n, H, H)
        dx, _, dss = self.final_res.backward(dy, need_dx1=False)               # d r is not needed (bd_0 is not differentiated)
        acc(self.final_res, dss)
        # decoder level j took its skips in pop order: b1 <- attn output of encoder level e = L-1-j, b2 <- b1 output of level e
        L = len(self.levels)
        d_attn_out, d_b1_out = [None] * L, [None] * L
        for j in range(L - 1, -1, -1):
            b1, b2, attn, up = self.ups[j]
            dx = up.backward(dx)
            dx = attn.backward(dx)
            dx, d_b1_out[L - 1 - j], dss = b2.backward(dx)
            acc(b2, dss)
            dx, d_attn_out[L - 1 - j], dss = b1.backward(dx)
            acc(b1, dss)
        dx, _, dss = self.mid2.backward(dx)
        acc(self.mid2, dss)
        dx = self.mid_attn.backward(dx)
        dx, _, dss = self.mid1.backward(dx)
        acc(self.mid1, dss)
        for e in range(L - 1, -1, -1):
            b1, b2, attn, down = self.levels[e]
            dx = down.backward(dx)
            self.ctx.add_(dx, d_attn_out[e])                  # the attn output feeds `down` and the decoder's b1
            dx = attn.backward(dx)
            dx, _, dss = b2.backward(dx, extra0=d_b1_out[e])  # the b1 output feeds b2 and the decoder's b2
            acc(b2, dss)
            dx, _, dss = b1.backward(dx, need_dx=(e != 0))    # level 0's input is init_conv(bd_0): not differentiated
            acc(b1, dss)
        return dts

    def theta_grad(self, dts):
        """Chain d SiLU(t) back through time_mlp (:300-305) and the sinusoidal embedding to theta."""
        ang, h1, t = self.ttape
        self.ttape = None
        dt = dts * _silu_grad(t)
        dg = self.ctx.linear(dt.contiguous(), self.t2[2])
        dh1 = (dg * _gelu_grad(h1)).contiguous()
        demb = self.ctx.linear(dh1, self.t1[2])
        half = self.freqs.shape[0]
        return ((demb[:, :half] * ang.cos() - demb[:, half:] * ang.sin()) * self.freqs[None, :]).sum(dim=1)

    # nn.Module look-alikes used by GaussianDiffusion._init_state (diffusion_2d_jellyfish.py:869-871)
    def to(self, *a, **k):
        return self

    def eval(self):
        return self


class HipDesignGradient:
    """`force_fn` (inference_2d_jellyfish.py:85-114) without autograd: guidance = -mean_t(force_t * (T - t)) + reg_ratio *
    reg_theta(theta) (:49-61) differentiated w.r.t. the pressure channel and the theta map of x.

    x [B, T, Cd, H, W] (Cd = 4: vx, vy, pressure, theta map; 2 with only_vis_pressure), bd_0 [B, T, 3, H, W].
    Returns d guidance / d x, same shape (the reference's cat([grad_state, grad_theta.unsqueeze(2)], dim=2))."""
    analytic = True

    def __init__(self, force_model, bd_updater, args, image_size=None):
        image_size = image_size or args.image_size
        self.force = force_model if isinstance(force_model, HipForceUnet) else HipForceUnet(force_model, image_size)
        self.unet = bd_updater if isinstance(bd_updater, HipUnet) else HipUnet(bd_updater, image_size)
        self.only_p = bool(args.only_vis_pressure)
        self.p_min, self.p_max, self.reg_ratio = float(args.p_min), float(args.p_max), float(args.reg_ratio)
        self.calibrated = _MODE == "f16x3"            # per-convolution operand scales of the backward pass (_Calibration)
        self.check_every = int(os.environ.get("DPC_SURROGATE_RANGE_CHECK_EVERY", "64"))
        self.watch_first = int(os.environ.get("DPC_SURROGATE_RANGE_WATCH_FIRST", "4"))
        self.calls = 0
        self.since_check = 0                          # design-gradient calls since the last check_range()
        self.last_calibration = None                  # [(max |input| on the synthetic input, scale)] of the set-up pass
        self._calibrated_for = None                   # (T, Cd, H, W) the scales were fixed for

    # largest activation of the two nets: the level-0 tensors [B T H W, dim] fp32 (dim = image_size, inference_2d_jellyfish.py
    # load_model).  The kernels address a tensor with 32-bit BYTE offsets inside buffer descriptors: keep every tensor below 2 GB.
    _MAX_TENSOR_BYTES = 3 << 29                 # 1.6 GB (measured: 1.34 GB tensors run, 2.68 GB fault)

    def calibrate(self, T, Cd, H, W, device):
        """Fix the operand scales of the backward-data convolutions on a seeded synthetic trajectory (see _Calibration): uniform
        [-1, 1] states, a constant theta map, a random boundary mask -- the same tensors on every rank.  Set-up, not loop, work."""
        g = torch.Generator().manual_seed(20240229)
        x = torch.rand(1, T, Cd, H, W, generator=g) * 2 - 1
        x[:, :, -1 if self.only_p else 3] = 0.3
        bd = (torch.rand(1, 1, 3, H, W, generator=g) > 0.7).float().expand(-1, T, -1, -1, -1).contiguous()
        self._run(x.to(device), bd.to(device), "calibrate")
        self._calibrated_for = (T, Cd, H, W)

    def check_range(self):
        """One host read per sample() call: did a backward tensor reach the fp16 clamp of its convolution since the last check?"""
        self.since_check = 0
        convs = [c for c in (_Calibration.convs or ()) if getattr(c, "peak", None) is not None and c.act_scale > 0]
        if not convs:
            return None
        # scaled peaks as ONE tensor test: Python's max() drops a NaN unless it comes first (every `nan > x` is False), so a NaN peak
        # of any convolution but the first used to pass
        peaks = torch.cat([c.peak.reshape(-1)[:1] for c in convs]).double().cpu()
        scaled = peaks * torch.tensor([c.act_scale for c in convs], dtype=torch.float64)
        for c in convs:
            c.peak = None
        finite = bool(torch.isfinite(scaled).all())
        worst = float(scaled.max()) if finite else float("nan")
        self.last_headroom = 65504.0 / worst if finite and worst > 0 else (float("inf") if finite else 0.0)   # x by which the largest watched tensor could still grow
        if not finite or worst > 65504.0:
            raise RuntimeError(f"design gradient: a backward tensor left the fp16 window of its convolution (max |x| * scale = {worst:.3g} "
                               "> 65504; scales were fixed on the synthetic calibration input): call calibrate() on representative "
                               "data or set DPC_SURROGATE_MODE=x6")
        return self.last_headroom

    def __call__(self, x, bd_0):
        B, T, Cd_, H, W = x.shape
        if self.calibrated and self._calibrated_for != (T, Cd_, H, W):
            self.calibrate(T, Cd_, H, W, x.device)
        # device-side peak watch of the backward operands (no host read here; check_range() reads once per sample()): the first
        # `watch_first` calls after every check -- the start of a chain is where a new input family would show -- and every
        # `check_every`-th call after that (one call in 64 alone left the other 63 unwatched from the first step on)
        check = "watch" if (self.calibrated and self.check_every > 0 and
                            (self.since_check < self.watch_first or self.calls % self.check_every == 0)) else None
        self.calls += 1
        self.since_check += 1
        per_traj = T * H * W * max(self.force.mid // 8, 64) * 4            # bytes of a level-0 activation per trajectory (dim = mid / 8)
        chunk = max(1, self._MAX_TENSOR_BYTES // per_traj)
        if B <= chunk:
            return self._run(x, bd_0, check)
        # J128 at 16 trajectories per GPU (128 x 128 x 20 frames, dim 128): the level-0 activations would be 2.7 GB -> the batch is
        # processed in chunks of <= `chunk` trajectories (they are independent; the operand scales are fixed, so chunking changes no bit)
        outs = []
        for b0 in range(0, B, chunk):
            outs.append(self._run(x[b0:b0 + chunk].contiguous(), bd_0[b0:b0 + chunk].contiguous(), check))
        return torch.cat(outs, dim=0)

    def _run(self, x, bd_0, check):
        B, T, Cd, H, W = x.shape
        n, HW = B * T, H * W
        c_p, c_th = (0, Cd - 1) if self.only_p else (2, 3)
        x = _f(x)
        L, S = _lib.lib(), _lib.stream
        theta = torch.empty(n, device=x.device)
        _lib.check(L.dpc_channel_mean(_lib.ptr(x), _lib.ptr(theta), n, Cd, c_th, HW, S()))
        inp = self.unet.forward_cl4(bd_0.reshape(n, *bd_0.shape[2:]), theta)
        a = 0.5 * (self.p_max - self.p_min)                                   # unnormalize_state :40-41
        _lib.check(L.dpc_channel_affine_to_cl(_lib.ptr(x), _lib.ptr(inp), n, Cd, c_p, 4, 0, a, a + self.p_min, HW, S()))
        force = self.force.forward_cl(inp, n)                                 # [n, 1]
        self.last_force = force
        w = torch.arange(T, 0, -1, dtype=torch.float32, device=x.device)
        dforce = (-(w / T)).repeat(B).reshape(n, 1).contiguous()
        _Calibration.active, _Calibration.watch, _Calibration.seen = check == "calibrate", check == "watch", []
        try:
            d_inp = self.force.backward(dforce)
            out = torch.zeros_like(x)
            _lib.check(L.dpc_cl_to_nchw(_lib.ptr(d_inp), _lib.ptr(out), n, 1, 4, 0, Cd, c_p, a, HW, S()))
            dts = self.unet.backward_theta(d_inp)
        finally:
            _Calibration.active = _Calibration.watch = False
        if check == "calibrate":
            self.last_calibration = list(_Calibration.seen)
        dtheta = self.unet.theta_grad(dts).reshape(B, T)
        th = theta.reshape(B, T)
        d = th[:, 1:] - th[:, :-1]
        dreg = torch.zeros_like(th)
        dreg[:, 1:] += 2 * d
        dreg[:, :-1] -= 2 * d
        dtheta = (dtheta + self.reg_ratio * dreg).reshape(n).contiguous()
        _lib.check(L.dpc_channel_fill(_lib.ptr(out), _lib.ptr(dtheta), n, Cd, c_th, 1.0 / HW, HW, S()))
        return out
