// Host-side orchestration of the Burgers space-time U-Net (`Unet2D`, model/burgers_1d/unet.py:267-431) on
// channels-last activations [B, H, W, C] (H = 16 padded time rows, W = 128 cells).  Same building blocks as the 3-D
// denoiser (unet3d.hip): the fp32-MFMA implicit GEMM for every convolution (3x3 = 9 taps, 1x1, the 2x2/stride-2 form
// of pixel-unshuffle + 1x1), the gather stem for the 7x7 init conv, GroupNorm/SiLU, channel LayerNorm statistics fused
// into the qkv projection, the linear-attention and softmax-attention cores.  No host synchronisation, no allocation
// inside forward: activations live in the caller's workspace (stack arena).
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "unet_common.h"

struct dpc_unet2d_s {
    dpc_unet2d_cfg cfg;
    std::vector<int> dims;                                          // [dim, dim*m0, dim*m1, ...]
    std::map<std::string, std::unique_ptr<dpc::DevBuf>> raw;
    std::map<std::string, std::unique_ptr<dpc::PackedConv>> conv;
    std::unique_ptr<dpc::DevBuf> stem_wp, stem_ktab;
    int stem_npad = 0, stem_kchunks = 0;
    std::set<std::string> loaded;
    dpc::DevBuf t_freq;
    bool have_tables = false, finalized = false;
    long long ws_key[3] = {0, 0, 0};      // (B, H, W) of the last workspace dry run and its result
    size_t ws_need = 0;
    dpc::Modes modes{2, 2, 2, 2};                                  // captured at create time (common.h: Modes)
    bool fused_attn = true;      // DPC_UNFUSED_ATTN=1 (captured at create) selects the unfused composition of LinearAttention (A/B tests)
    bool fused_gn = true;        // DPC_UNFUSED_GN=1: standalone GroupNorm passes instead of the conv-fused form of the 16 x 128 / 8 x 64 levels
    bool taps_on = false;
    struct Tap { std::unique_ptr<dpc::DevBuf> buf; size_t floats = 0; };
    std::map<std::string, Tap> taps;
};

namespace dpc {

static std::vector<std::string> expected_names2d(const dpc_unet2d_cfg& c, const std::vector<int>& dims) {
    std::vector<std::string> v;
    auto res = [&](const std::string& p, int di, int dout) {
        v.push_back(p + ".mlp.1.weight");
        v.push_back(p + ".mlp.1.bias");
        for (const char* b : {".block1", ".block2"}) {
            v.push_back(p + b + ".proj.weight");
            v.push_back(p + b + ".proj.bias");
            v.push_back(p + b + ".norm.weight");
            v.push_back(p + b + ".norm.bias");
        }
        if (di != dout) { v.push_back(p + ".res_conv.weight"); v.push_back(p + ".res_conv.bias"); }
    };
    auto lattn = [&](const std::string& p) {
        v.push_back(p + ".fn.fn.to_qkv.weight");
        v.push_back(p + ".fn.fn.to_out.0.weight");
        v.push_back(p + ".fn.fn.to_out.0.bias");
        v.push_back(p + ".fn.fn.to_out.1.g");
        v.push_back(p + ".fn.norm.g");
    };
    for (const char* n : {"time_mlp.1.weight", "time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias",
                          "init_conv.weight", "init_conv.bias"})
        v.push_back(n);
    const int nres = c.n_mults;
    for (int i = 0; i < nres; ++i) {
        const std::string p = "downs." + std::to_string(i);
        res(p + ".0", dims[i], dims[i]);
        res(p + ".1", dims[i], dims[i]);
        lattn(p + ".2");
        const std::string d = (i < nres - 1) ? p + ".3.1" : p + ".3";
        v.push_back(d + ".weight");
        v.push_back(d + ".bias");
    }
    const int mid = dims[nres];
    res("mid_block1", mid, mid);
    for (const char* n : {"mid_attn.fn.fn.to_qkv.weight", "mid_attn.fn.fn.to_out.weight", "mid_attn.fn.fn.to_out.bias",
                          "mid_attn.fn.norm.g"})
        v.push_back(n);
    res("mid_block2", mid, mid);
    for (int i = 0; i < nres; ++i) {
        const int di = dims[nres - 1 - i], dout = dims[nres - i];
        const std::string p = "ups." + std::to_string(i);
        res(p + ".0", dout + di, dout);
        res(p + ".1", dout + di, dout);
        lattn(p + ".2");
        const std::string u = (i < nres - 1) ? p + ".3.1" : p + ".3";
        v.push_back(u + ".weight");
        v.push_back(u + ".bias");
    }
    res("final_res_block", c.dim * 2, c.dim);
    v.push_back("final_conv.weight");
    v.push_back("final_conv.bias");
    return v;
}

static bool ends_with2(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

struct Runner2D {
    dpc_unet2d_s* h;
    Arena ar;
    hipStream_t s;
    int mb, H, W;
    float* temb = nullptr;
    int rc = DPC_OK;
    bool dry() const { return ar.dry; }

    const float* raw(const std::string& n) {
        if (dry()) return nullptr;
        auto it = h->raw.find(n);
        if (it == h->raw.end()) { rc = fail(DPC_ERR_STATE, "missing parameter " + n); return nullptr; }
        return it->second->f();
    }
    const PackedConv* conv(const std::string& n) {
        if (dry()) return nullptr;
        auto it = h->conv.find(n);
        if (it == h->conv.end()) { rc = fail(DPC_ERR_STATE, "missing packed weight " + n); return nullptr; }
        return it->second.get();
    }
#define RUN(expr) do { if (!dry() && rc == DPC_OK) { int _r = (expr); if (_r) rc = _r; } } while (0)

    void tap(const std::string& name, const float* x_cl, int C, int Hl, int Wl) {
        if (dry() || !h->taps_on || rc) return;
        auto& t = h->taps[name];
        const size_t n = (size_t)mb * Hl * Wl * C;
        if (!t.buf || t.floats != n) {
            t.buf.reset(new DevBuf());
            if (t.buf->alloc(n * sizeof(float))) { rc = DPC_ERR_HIP; return; }
            t.floats = n;
        }
        RUN(launch_cl_to_cf(x_cl, t.buf->f(), mb, C, (long long)Hl * Wl, 1, s));
    }

    void convolve(const std::string& wname, const std::string& bname, const float* x0, const float* x1, int C0, int C1,
                  const float* resid, float* out, int Hi, int Wi, int Ho, int Wo, const float* ln_stats,
                  const float* ln_gamma, int out_mode) {
        const PackedConv* pc = conv(wname);
        if (pc)
            RUN(run_conv(*pc, x0, x1, C0, C1, bname.empty() ? nullptr : raw(bname), resid, out, mb, 1, Hi, Wi, Ho, Wo,
                         ln_stats, ln_gamma, out_mode, 0, 0, s));
    }

    // Block: conv3x3 -> GN -> (scale,shift) -> SiLU (+resid)   (unet.py:134-155)
    void block(const std::string& p, const float* x0, const float* x1, int C0, int C1, int Cout, float* conv_out,
               float* act_out, const float* resid, const float* scale_shift, int Hl, int Wl) {
        convolve(p + ".proj.weight", p + ".proj.bias", x0, x1, C0, C1, nullptr, conv_out, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
        const size_t m = ar.mark();
        void* ws = ar.alloc(gn_workspace_bytes(mb, Cout));
        RUN(launch_groupnorm_silu(conv_out, act_out, resid, raw(p + ".norm.weight"), raw(p + ".norm.bias"), scale_shift,
                                  mb, (long long)Hl * Wl, Cout, h->cfg.groups, ws, s));
        ar.release(m);
    }

    // Every ResnetBlock's (scale, shift) = mlp(temb) (unet.py:163-166, 176-179) depends on the time embedding only: all of them in one
    // launch, in the order the traversal consumes them (downs i.0, i.1; mid 1, 2; ups i.0, i.1; final_res_block).
    std::vector<float*> ss_pre;
    size_t ss_next = 0;
    void time_projections(const std::vector<int>& dims, int nres) {
        std::vector<std::pair<std::string, int>> blocks;
        for (int i = 0; i < nres; ++i)
            for (int j = 0; j < 2; ++j) blocks.push_back({"downs." + std::to_string(i) + "." + std::to_string(j), dims[i]});
        blocks.push_back({"mid_block1", dims[nres]});
        blocks.push_back({"mid_block2", dims[nres]});
        for (int i = 0; i < nres; ++i)
            for (int j = 0; j < 2; ++j) blocks.push_back({"ups." + std::to_string(i) + "." + std::to_string(j), dims[nres - i]});
        blocks.push_back({"final_res_block", dims[0]});
        ss_pre.clear();
        ss_next = 0;
        if (blocks.size() > 32) return;                    // (deeper nets keep the per-block launches)
        SmallLinearBatch d{};
        for (const auto& bk : blocks) {
            float* ss = ar.allocf((long long)mb * 2 * bk.second);
            ss_pre.push_back(ss);
            d.W[d.count] = raw(bk.first + ".mlp.1.weight");
            d.bias[d.count] = raw(bk.first + ".mlp.1.bias");
            d.out[d.count] = ss;
            d.N[d.count] = 2 * bk.second;
            ++d.count;
        }
        RUN(launch_small_linear_multi(temb, d, mb, h->cfg.dim * 4, 1, 0, s));
    }

    // ResnetBlock (unet.py:157-191).  dst == x0 (in place) is allowed when C1 == 0 and C0 == Cout.
    void resnet(const std::string& p, const float* x0, const float* x1, int C0, int C1, int Cout, float* dst, int Hl, int Wl) {
        const long long P = (long long)mb * Hl * Wl;
        const size_t m = ar.mark();
        float* ss;
        if (ss_next < ss_pre.size()) {                     // computed up front with all the other blocks' (time_projections)
            ss = ss_pre[ss_next++];
        } else {
            ss = ar.allocf((long long)mb * 2 * Cout);
            RUN(launch_small_linear(temb, raw(p + ".mlp.1.weight"), raw(p + ".mlp.1.bias"), ss, mb, h->cfg.dim * 4, 2 * Cout, 1, 0, s));
        }
        const bool same = (C1 == 0 && C0 == Cout);
        const PackedConv* c1 = conv(p + ".block1.proj.weight");
        const PackedConv* c2 = conv(p + ".block2.proj.weight");
        // r05: GroupNorm fused around the two 3x3 convolutions where both run on the halo-tile kernel (H, W multiples of 8: the 16 x 128 and
        // 8 x 64 levels of the Burgers nets): per-image statistics come out of the conv epilogues, block1's normalise + (scale, shift) +
        // SiLU is applied inside block2's halo load (h1 never exists in activated form), only block2's apply (+ residual) stays a
        // streaming pass -- 3 passes over the activation per ResnetBlock instead of 7 (unet.py:134-191).  Shape-only rule.
        const bool fused = h->fused_gn && igemm_mode_default() != 0 && C0 % 4 == 0 && C1 % 4 == 0 &&
                           conv2d_gn_fusable(Cout, igemm_npad(Cout), Hl, Wl);     // (modes + shapes only: the dry run decides alike)
        if (fused) {
            if (!dry() && !rc && !(c1 && c2 && c1->flat3 && c2->flat3)) rc = fail(DPC_ERR_STATE, "unet2d: " + p + ": 3x3 pack for the halo kernel missing");
            const long long ent = conv3f3c_flat_gn_entries(Hl, Wl), R = (long long)Hl * Wl;
            float* raw1 = ar.allocf(P * Cout);
            float* part = ar.allocf((long long)mb * ent * Cout * 2);
            float* stats = ar.allocf((long long)mb * Cout * 2);
            float* coef = ar.allocf((long long)mb * Cout * 7);
            float* raw2 = (same && dst == x0) ? ar.allocf(P * Cout) : dst;        // (not same: the fused res_conv epilogue reads raw2[i] and writes
                                                                                  //  dst[i] element for element -- in place is fine, as in the 3-D net)
            if (c1 && c2) {
                RUN(run_conv(*c1, x0, x1, C0, C1, raw(p + ".block1.proj.bias"), nullptr, raw1, mb, 1, Hl, Wl, Hl, Wl, nullptr, nullptr, 0, 0, 0,
                             s, part, nullptr));
                RUN(launch_gn_finalize_fused(part, mb, 0, Cout, h->cfg.groups, R, raw(p + ".block1.norm.weight"), raw(p + ".block1.norm.bias"),
                                             ss, stats, coef, s, ent));
                RUN(run_conv(*c2, raw1, nullptr, Cout, 0, raw(p + ".block2.proj.bias"), nullptr, raw2, mb, 1, Hl, Wl, Hl, Wl, nullptr, nullptr, 0,
                             0, 0, s, part, coef));
                // block2's GroupNorm + SiLU: a streaming pass with the identity residual, or -- when the block has a res_conv (the up path's
                // concatenated inputs) -- folded into the res_conv's epilogue as in the 3-D net (block2(h) + res_conv(x) without materialising
                // block2's activated output; per-IMAGE coefficients: rows per sample = H W)
                const PackedConv* rcv = same ? nullptr : conv(p + ".res_conv.weight");
                const bool fuse_res = rcv && !dry() && conv_can_fuse_gn_residual(*rcv, R);
                if (fuse_res) {
                    RUN(launch_gn_finalize_fused(part, mb, 0, Cout, h->cfg.groups, R, raw(p + ".block2.norm.weight"), raw(p + ".block2.norm.bias"),
                                                 nullptr, stats, coef, s, ent));
                    RUN(run_conv(*rcv, x0, x1, C0, C1, raw(p + ".res_conv.bias"), nullptr, dst, mb, 1, Hl, Wl, Hl, Wl, nullptr, nullptr, 0, 0, 0, s,
                                 nullptr, nullptr, raw2, coef));
                } else {
                    RUN(launch_gn_finalize_fused(part, mb, 0, Cout, h->cfg.groups, R, nullptr, nullptr, nullptr, stats, nullptr, s, ent));
                    RUN(launch_gn_apply(raw2, dst, same ? x0 : nullptr, stats, raw(p + ".block2.norm.weight"), raw(p + ".block2.norm.bias"),
                                        nullptr, mb, R, Cout, h->cfg.groups, s));
                    if (!same)
                        convolve(p + ".res_conv.weight", p + ".res_conv.bias", x0, x1, C0, C1, dst, dst, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
                }
            }
            ar.release(m);
            return;
        }
        float* h1 = ar.allocf(P * Cout);
        block(p + ".block1", x0, x1, C0, C1, Cout, h1, h1, nullptr, ss, Hl, Wl);
        if (same) {
            float* h2 = (dst == x0) ? ar.allocf(P * Cout) : dst;
            block(p + ".block2", h1, nullptr, Cout, 0, Cout, h2, dst, x0, nullptr, Hl, Wl);     // + x (identity res_conv)
        } else {
            block(p + ".block2", h1, nullptr, Cout, 0, Cout, dst, dst, nullptr, nullptr, Hl, Wl);
            convolve(p + ".res_conv.weight", p + ".res_conv.bias", x0, x1, C0, C1, dst, dst, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
        }
        ar.release(m);
    }

    // Residual(PreNorm(LinearAttention)) in place (unet.py:193-236): LN -> qkv -> softmaxes/context -> to_out conv -> LN -> + x
    void linear_attention(const std::string& p, float* x, int C, int Hl, int Wl) {
        const long long P = (long long)mb * Hl * Wl;
        const int HD = h->cfg.attn_heads * 32;
        if (h->fused_attn && h->modes.attn == 2 && lattn3_supported(C, h->cfg.attn_heads) && h->raw.count(p + ".fn.fn.to_qkv.weight#h3")) {
            // one launch per block, one workgroup per sample (lattn3.hip, OUT_LN form): x is read twice and written once, the
            // qkv / attention / projection tensors never exist in HBM (shape-only rule: C = 64 / 128, 4 heads)
            LattnParams lp{};
            lp.x = x; lp.out = x; lp.gamma = raw(p + ".fn.norm.g"); lp.bout = raw(p + ".fn.fn.to_out.0.bias");
            lp.gamma_out = raw(p + ".fn.fn.to_out.1.g");
            lp.images = mb; lp.N = Hl * Wl;
            const float* q3 = raw(p + ".fn.fn.to_qkv.weight#h3");
            const float* o3 = raw(p + ".fn.fn.to_out.0.weight#h3");
            RUN(launch_lattn3(lp, reinterpret_cast<const unsigned char*>(q3), reinterpret_cast<const unsigned char*>(o3), C, s));
            return;
        }
        const size_t m = ar.mark();
        float* stats = ar.allocf(P * 2);
        float* qkv = ar.allocf(P * 3 * HD);
        float* att = ar.allocf(P * HD);
        float* proj = ar.allocf(P * C);
        void* ws = ar.alloc(linattn_workspace_bytes(mb, h->cfg.attn_heads));
        RUN(launch_ln_stats(x, stats, P, C, s));
        convolve(p + ".fn.fn.to_qkv.weight", "", x, nullptr, C, 0, nullptr, qkv, Hl, Wl, Hl, Wl, stats, raw(p + ".fn.norm.g"), 0);
        RUN(launch_linear_attention(qkv, att, h->cfg.attn_heads, mb, Hl * Wl, ws, s));
        convolve(p + ".fn.fn.to_out.0.weight", p + ".fn.fn.to_out.0.bias", att, nullptr, HD, 0, nullptr, proj, Hl, Wl, Hl, Wl,
                 nullptr, nullptr, 0);
        RUN(launch_ln_stats(proj, stats, P, C, s));
        RUN(launch_ln_apply(proj, stats, raw(p + ".fn.fn.to_out.1.g"), x, x, P, C, s));
        ar.release(m);
    }

    // Residual(PreNorm(Attention)) in place (unet.py:238-272): dense softmax attention over the H*W tokens of an image
    void attention(const std::string& p, float* x, int C, int Hl, int Wl) {
        const long long P = (long long)mb * Hl * Wl;
        const int HD = h->cfg.attn_heads * 32;
        const long long HWl = (long long)Hl * Wl;
        const size_t m = ar.mark();
        float* stats = ar.allocf(P * 2);
        float* qkv = ar.allocf(P * 3 * HD);
        float* att = ar.allocf(P * HD);
        RUN(launch_ln_stats(x, stats, P, C, s));
        convolve(p + ".fn.fn.to_qkv.weight", "", x, nullptr, C, 0, nullptr, qkv, Hl, Wl, Hl, Wl, stats, raw(p + ".fn.norm.g"), 0);
        AttnParams ap{};
        ap.qkv = qkv; ap.out = att; ap.heads = h->cfg.attn_heads;
        ap.L = (int)HWl; ap.n_seq = mb; ap.seq_inner = 1; ap.seq_outer_stride = HWl; ap.seq_inner_stride = 0; ap.token_stride = 1;
        RUN(launch_attention(ap, s));
        convolve(p + ".fn.fn.to_out.weight", p + ".fn.fn.to_out.bias", att, nullptr, HD, 0, x, x, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
        ar.release(m);
    }

    void forward(const float* x_in, const int64_t* t_in, float* out) {
        const dpc_unet2d_cfg& c = h->cfg;
        const int dim = c.dim, nres = c.n_mults;
        const std::vector<int>& dims = h->dims;
        const long long P0 = (long long)mb * H * W;
        // time embedding (unet.py:316-321, 401)
        float* sinemb = ar.allocf((long long)mb * dim);
        float* t1 = ar.allocf((long long)mb * dim * 4);
        temb = ar.allocf((long long)mb * dim * 4);
        // library scratch of this forward (common.h: ScratchScope): SiLU(temb) of the one-launch time projections, then the split-K
        // partials of the deep levels (4 slices x rows x channels: rows x channels halves per level and the reductions long
        // enough to split -- >= 128 iterations of 32 channels -- start at level 2, i.e. <= one level-0 activation; twice that is lent)
        const size_t scratch_bytes = std::max((size_t)mb * dim * 4, (size_t)P0 * dim * 2) * sizeof(float);
        void* scratch_mem = ar.alloc(scratch_bytes);
        ScratchScope scratch_scope(dry() ? nullptr : scratch_mem, scratch_bytes);
        RUN(launch_sinusoidal(t_in, h->t_freq.f(), sinemb, mb, dim / 2, s));
        RUN(launch_small_linear(sinemb, raw("time_mlp.1.weight"), raw("time_mlp.1.bias"), t1, mb, dim, dim * 4, 0, 2, s));
        RUN(launch_small_linear(t1, raw("time_mlp.3.weight"), raw("time_mlp.3.bias"), temb, mb, dim * 4, dim * 4, 0, 0, s));
        if (!dry() && h->taps_on) {
            auto& t = h->taps["time_mlp"];
            const size_t n = (size_t)mb * dim * 4;
            if (!t.buf || t.floats != n) { t.buf.reset(new DevBuf()); if (t.buf->alloc(n * 4)) rc = DPC_ERR_HIP; t.floats = n; }
            if (!rc) RUN((hipMemcpyAsync(t.buf->p, temb, n * 4, hipMemcpyDeviceToDevice, s) == hipSuccess) ? 0 : DPC_ERR_HIP);
        }
        time_projections(dims, nres);
        // init_conv 7x7 (unet.py:333, 398) straight from the reference layout [B, C, H, W]
        float* X0 = ar.allocf(P0 * dim);
        {
            StemParams sp{};
            sp.x = x_in; sp.wp = dry() ? nullptr : h->stem_wp->f(); sp.ktab = dry() ? nullptr : (const int*)h->stem_ktab->p;
            sp.bias = raw("init_conv.bias"); sp.out = X0; sp.BF = mb; sp.F = 1; sp.C = c.channels; sp.H = H; sp.W = W;
            sp.Ctot = c.channels; sp.c_off = 0;
            sp.N = dim; sp.Npad = h->stem_npad; sp.kchunks = h->stem_kchunks; sp.M = P0;
            RUN(launch_stem(sp, s));
        }
        tap("init_conv", X0, dim, H, W);

        std::vector<float*> skips;
        const float* x = X0;
        int Hl = H, Wl = W;
        for (int i = 0; i < nres; ++i) {
            const std::string p = "downs." + std::to_string(i);
            const int di = dims[i], dout = dims[i + 1];
            const long long P = (long long)mb * Hl * Wl;
            float* A = ar.allocf(P * di);
            resnet(p + ".0", x, nullptr, di, 0, di, A, Hl, Wl);
            tap(p + ".0", A, di, Hl, Wl);
            skips.push_back(A);
            float* Bq = ar.allocf(P * di);
            resnet(p + ".1", A, nullptr, di, 0, di, Bq, Hl, Wl);
            tap(p + ".1", Bq, di, Hl, Wl);
            linear_attention(p + ".2", Bq, di, Hl, Wl);
            tap(p + ".2", Bq, di, Hl, Wl);
            skips.push_back(Bq);
            if (i < nres - 1) {
                // Downsample2d = pixel-unshuffle(2) + 1x1 conv == 2x2 stride-2 conv on the weight viewed [N][C][2][2] (:46-50)
                const int Ho = Hl / 2, Wo = Wl / 2;
                float* D = ar.allocf((long long)mb * Ho * Wo * dout);
                convolve(p + ".3.1.weight", p + ".3.1.bias", Bq, nullptr, di, 0, nullptr, D, Hl, Wl, Ho, Wo, nullptr, nullptr, 0);
                tap(p + ".3", D, dout, Ho, Wo);
                x = D; Hl = Ho; Wl = Wo;
            } else {
                float* D = ar.allocf(P * dout);
                convolve(p + ".3.weight", p + ".3.bias", Bq, nullptr, di, 0, nullptr, D, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
                tap(p + ".3", D, dout, Hl, Wl);
                x = D;
            }
        }
        const int mid = dims[nres];
        float* Mx = ar.allocf((long long)mb * Hl * Wl * mid);
        resnet("mid_block1", x, nullptr, mid, 0, mid, Mx, Hl, Wl);
        tap("mid_block1", Mx, mid, Hl, Wl);
        attention("mid_attn", Mx, mid, Hl, Wl);
        tap("mid_attn", Mx, mid, Hl, Wl);
        resnet("mid_block2", Mx, nullptr, mid, 0, mid, Mx, Hl, Wl);
        tap("mid_block2", Mx, mid, Hl, Wl);
        x = Mx;
        for (int i = 0; i < nres; ++i) {
            const std::string p = "ups." + std::to_string(i);
            const int di = dims[nres - 1 - i], dout = dims[nres - i];
            const long long P = (long long)mb * Hl * Wl;
            const float* sk = skips.back(); skips.pop_back();
            float* U = ar.allocf(P * dout);
            resnet(p + ".0", x, sk, dout, di, dout, U, Hl, Wl);                  // cat((x, h.pop()), dim=1) is virtual
            tap(p + ".0", U, dout, Hl, Wl);
            sk = skips.back(); skips.pop_back();
            float* V = ar.allocf(P * dout);
            resnet(p + ".1", U, sk, dout, di, dout, V, Hl, Wl);
            tap(p + ".1", V, dout, Hl, Wl);
            linear_attention(p + ".2", V, dout, Hl, Wl);
            tap(p + ".2", V, dout, Hl, Wl);
            if (i < nres - 1) {
                const int Ho = 2 * Hl, Wo = 2 * Wl;                               // Upsample2d: nearest x2 + 3x3 conv (:40-44)
                float* Up = ar.allocf((long long)mb * Ho * Wo * dout);
                RUN(launch_upsample2x_cl(V, Up, mb, Hl, Wl, dout, s));
                float* Y = ar.allocf((long long)mb * Ho * Wo * di);
                convolve(p + ".3.1.weight", p + ".3.1.bias", Up, nullptr, dout, 0, nullptr, Y, Ho, Wo, Ho, Wo, nullptr, nullptr, 0);
                tap(p + ".3", Y, di, Ho, Wo);
                x = Y; Hl = Ho; Wl = Wo;
            } else {
                float* Y = ar.allocf(P * di);
                convolve(p + ".3.weight", p + ".3.bias", V, nullptr, dout, 0, nullptr, Y, Hl, Wl, Hl, Wl, nullptr, nullptr, 0);
                tap(p + ".3", Y, di, Hl, Wl);
                x = Y;
            }
        }
        // cat((x, r)) -> final_res_block -> 1x1 conv written in the reference layout [B, out_dim, H, W]
        float* Fz = ar.allocf(P0 * dim);
        resnet("final_res_block", x, X0, dim, dim, dim, Fz, Hl, Wl);
        tap("final_res_block", Fz, dim, Hl, Wl);
        convolve("final_conv.weight", "final_conv.bias", Fz, nullptr, dim, 0, nullptr, out, Hl, Wl, Hl, Wl, nullptr, nullptr, 1);
    }
#undef RUN
};

static int micro_batch_of2d(const dpc_unet2d_s* h, int B) {
    int mb = h->cfg.micro_batch;
    if (mb <= 0 || mb > B) mb = B;
    return mb;
}

}  // namespace dpc

using namespace dpc;

extern "C" {

int dpc_unet2d_create(const dpc_unet2d_cfg* cfg, dpc_unet2d_t* out) {
    DPC_REQUIRE(cfg && out, "unet2d_create: null argument");
    DPC_REQUIRE(cfg->attn_dim_head == 32, "unet2d: attn_dim_head must be 32");
    DPC_REQUIRE(cfg->n_mults >= 1 && cfg->n_mults <= 8, "unet2d: 1..8 resolutions");
    DPC_REQUIRE(cfg->dim % 8 == 0 && cfg->dim >= 8, "unet2d: dim must be a multiple of 8");
    DPC_REQUIRE(cfg->channels >= 1 && cfg->channels <= 255, "unet2d: channels");
    DPC_REQUIRE(cfg->groups >= 1, "unet2d: groups");
    auto* h = new dpc_unet2d_s();
    h->cfg = *cfg;
    h->modes = modes_global();
    h->fused_attn = !debug_switch("DPC_UNFUSED_ATTN", 0);
    h->fused_gn = !debug_switch("DPC_UNFUSED_GN", 0);
    if (h->cfg.out_dim <= 0) h->cfg.out_dim = h->cfg.channels;
    h->dims.push_back(cfg->dim);
    for (int i = 0; i < cfg->n_mults; ++i) h->dims.push_back(cfg->dim * cfg->dim_mults[i]);
    *out = h;
    return DPC_OK;
}

void dpc_unet2d_destroy(dpc_unet2d_t h) { delete h; }

int dpc_unet2d_load(dpc_unet2d_t h, const char* name_c, const float* w, const int64_t* shape, int ndim,
                    dpc_stream_t stream) {
    DPC_REQUIRE(h && name_c && w && shape && ndim >= 1 && ndim <= 4, "unet2d_load: bad argument");
    ModeScope mode_scope(h->modes);
    hipStream_t s = (hipStream_t)stream;
    const std::string name(name_c);
    bool known = false;
    for (const auto& e : expected_names2d(h->cfg, h->dims)) if (e == name) { known = true; break; }
    DPC_REQUIRE(known, "unet2d_load: unknown parameter name '" + name + "'");
    long long numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    int rc = DPC_OK;
    if (name == "init_conv.weight") {
        DPC_REQUIRE(ndim == 4 && shape[1] == h->cfg.channels && shape[2] == shape[3] && shape[2] % 2 == 1 && shape[2] <= 15,
                    "init_conv.weight shape");
        const int N = (int)shape[0], C = (int)shape[1], k = (int)shape[2];
        h->stem_npad = (int)align_up(N, 64);
        h->stem_kchunks = igemm_kchunks(k * k * C);
        h->stem_wp.reset(new DevBuf());
        h->stem_ktab.reset(new DevBuf());
        if ((rc = h->stem_wp->alloc((size_t)h->stem_kchunks * h->stem_npad * 32 * sizeof(float)))) return rc;
        if ((rc = h->stem_ktab->alloc((size_t)h->stem_kchunks * 32 * sizeof(int)))) return rc;
        rc = launch_pack_stem(w, h->stem_wp->f(), (int*)h->stem_ktab->p, N, h->stem_npad, C, k, s, 1);
    } else if (ndim == 4 && !ends_with2(name, ".g")) {
        auto pc = std::make_unique<PackedConv>();
        const int N = (int)shape[0];
        int K = (int)shape[1], kh = (int)shape[2], kw = (int)shape[3];
        if (name.rfind("downs.", 0) == 0 && ends_with2(name, ".3.1.weight")) {
            // Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w') + Conv2d(4c, out, 1): the [N][4c] weight IS a [N][c][2][2] kernel
            DPC_REQUIRE(kh == 1 && kw == 1 && K % 4 == 0, "Downsample2d weight must be [N, 4c, 1, 1]");
            K /= 4;
            rc = pack_conv3d(*pc, w, N, K, 1, 2, 2, 2, 2, 0, 0, 0, s);
        } else {
            DPC_REQUIRE(kh == kw && (kh == 1 || kh == 3), "unet2d: 1x1 and 3x3 convolutions only");
            rc = pack_conv3d(*pc, w, N, K, 1, kh, kw, 1, 1, 0, kh / 2, kw / 2, s);
        }
        h->conv[name] = std::move(pc);
        // LinearAttention projections also as the pre-split per-head images of the weight-stationary fused kernel (lattn3.hip)
        const bool is_qkv = ends_with2(name, ".fn.fn.to_qkv.weight"), is_out = ends_with2(name, ".fn.fn.to_out.0.weight");
        if (!rc && (is_qkv || is_out) && kh == 1) {
            const int C = is_out ? N : K, inner = is_out ? K : N / 3;
            if (inner == 128 && lattn3_supported(C, h->cfg.attn_heads) && h->modes.attn == 2) {
                auto b3 = std::make_unique<DevBuf>();
                if ((rc = b3->alloc(is_out ? tattn3_out_bytes(C) : tattn3_qkv_bytes(C)))) return rc;
                rc = launch_pack_tattn3(w, reinterpret_cast<unsigned char*>(b3->p), C, is_out, s);
                h->raw[name + "#h3"] = std::move(b3);
            }
        }
    } else {
        auto b = std::make_unique<DevBuf>();
        if ((rc = b->alloc((size_t)numel * sizeof(float)))) return rc;
        DPC_HIP(hipMemcpyAsync(b->p, w, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
        h->raw[name] = std::move(b);
    }
    if (rc == DPC_OK) h->loaded.insert(name);
    h->finalized = false;
    h->ws_need = 0;
    return rc;
}

int dpc_unet2d_set_tables(dpc_unet2d_t h, const float* freqs, dpc_stream_t stream) {
    DPC_REQUIRE(h && freqs, "unet2d_set_tables: bad argument");
    int rc;
    if ((rc = h->t_freq.alloc((size_t)(h->cfg.dim / 2) * 4))) return rc;
    DPC_HIP(hipMemcpyAsync(h->t_freq.p, freqs, h->t_freq.bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    h->have_tables = true;
    return DPC_OK;
}

int dpc_unet2d_finalize(dpc_unet2d_t h) {
    DPC_REQUIRE(h, "unet2d_finalize: null handle");
    if (!h->have_tables) return fail(DPC_ERR_STATE, "unet2d_finalize: call dpc_unet2d_set_tables first");
    for (const auto& e : expected_names2d(h->cfg, h->dims))
        if (!h->loaded.count(e)) return fail(DPC_ERR_STATE, "unet2d_finalize: parameter not loaded: " + e);
    if (int rc = f16x3_weight_overflow_check("unet2d_finalize")) return rc;
    h->finalized = true;
    return DPC_OK;
}

size_t dpc_unet2d_workspace_bytes(dpc_unet2d_t h, int B, int H, int W) {
    if (!h || B <= 0) return 0;
    if (h->ws_need && h->ws_key[0] == B && h->ws_key[1] == H && h->ws_key[2] == W) return h->ws_need;
    ModeScope mode_scope(h->modes);
    Runner2D r{};
    r.h = h; r.s = nullptr; r.mb = micro_batch_of2d(h, B); r.H = H; r.W = W;
    r.ar.dry = true;
    r.forward(nullptr, nullptr, nullptr);
    h->ws_key[0] = B; h->ws_key[1] = H; h->ws_key[2] = W;
    h->ws_need = r.ar.peak + 256;
    return h->ws_need;
}

int dpc_unet2d_forward(dpc_unet2d_t h, const float* x, const int64_t* t, float* out, int B, int H, int W, void* ws,
                       size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(h && x && t && out, "unet2d_forward: null argument");
    if (!h->finalized) return fail(DPC_ERR_STATE, "unet2d_forward: call dpc_unet2d_finalize first");
    ModeScope mode_scope(h->modes);
    const int levels = h->cfg.n_mults - 1;
    DPC_REQUIRE(H % (1 << levels) == 0 && W % (1 << levels) == 0, "unet2d_forward: H, W must be divisible by 2^(levels-1)");
    if (B == 0) return DPC_OK;
    const int mb = micro_batch_of2d(h, B);
    const size_t need = dpc_unet2d_workspace_bytes(h, B, H, W);
    if (ws_bytes < need || !ws) return fail(DPC_ERR_STATE, "unet2d_forward: workspace too small: need " + std::to_string(need));
    const long long in_per = (long long)h->cfg.channels * H * W, out_per = (long long)h->cfg.out_dim * H * W;
    for (int b0 = 0; b0 < B; b0 += mb) {
        Runner2D r{};
        r.h = h; r.s = (hipStream_t)stream; r.mb = std::min(mb, B - b0); r.H = H; r.W = W;
        r.ar.dry = false;
        r.ar.base = reinterpret_cast<char*>(align_up((size_t)ws, 256));
        r.ar.cap = ws_bytes - (size_t)(r.ar.base - (char*)ws);
        r.forward(x + b0 * in_per, t + b0, out + b0 * out_per);
        if (r.rc) return r.rc;
        if (r.ar.overflow) return fail(DPC_ERR_STATE, "unet2d_forward: arena overflow");
    }
    return DPC_OK;
}

const char* dpc_unet2d_modes(dpc_unet2d_t h) {
    static thread_local std::string buf;
    buf = h ? modes_string(h->modes) : std::string();
    return buf.c_str();
}

int dpc_unet2d_debug_taps(dpc_unet2d_t h, int enable) {
    DPC_REQUIRE(h, "null handle");
    h->taps_on = enable != 0;
    if (!enable) h->taps.clear();
    return DPC_OK;
}

int dpc_unet2d_get_tap(dpc_unet2d_t h, const char* name, float* dst, size_t dst_floats, dpc_stream_t stream) {
    DPC_REQUIRE(h && name && dst, "get_tap: null argument");
    auto it = h->taps.find(name);
    if (it == h->taps.end() || !it->second.buf) return fail(DPC_ERR_STATE, std::string("get_tap: no such tap ") + name);
    DPC_REQUIRE(dst_floats >= it->second.floats, "get_tap: destination too small");
    DPC_HIP(hipMemcpyAsync(dst, it->second.buf->p, it->second.floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DPC_OK;
}

}  // extern "C"
