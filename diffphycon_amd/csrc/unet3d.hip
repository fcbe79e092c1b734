// Host-side orchestration of the space-time U-Net forward on channels-last activations.
// Mirrors Unet3D_with_Conv3D.forward (video_diffusion_pytorch_conv3d.py:486-552) launch for launch; no host
// synchronisation, no allocation: activations live in a caller-provided workspace carved by a stack arena.
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "unet_common.h"

struct dpc_unet3d_s {
    dpc_unet3d_cfg cfg;
    std::vector<int> dims;                                  // [init_dim, dim*m0, dim*m1, ...]
    std::map<std::string, std::unique_ptr<dpc::DevBuf>> raw;        // reference-layout small params
    std::map<std::string, std::unique_ptr<dpc::PackedConv>> conv;   // packed GEMM operands (ups.*.4 -> name#ab)
    std::unique_ptr<dpc::DevBuf> stem_wp, stem_ktab, stem_wp6;     // stem_wp6: pre-split weights of the LDS-halo bf16x6 stem
    int stem_npad = 0, stem_kchunks = 0;
    std::set<std::string> loaded;
    // tables
    int frames = 0;
    dpc::DevBuf t_bias, t_bias32, t_brel, t_cos, t_sin, t_freq;
    bool bias_toeplitz = false;  // the bias table is a function of (key - query) only: t_brel holds its [heads][128] form
    bool finalized = false;
    long long ws_key[4] = {0, 0, 0, 0};   // (B, F, H, W) of the last workspace dry run and its result (a forward needs it twice)
    size_t ws_need = 0;
    dpc::Modes modes{2, 2, 2, 2}; // arithmetic modes captured at create time (common.h: Modes); every call runs under them
    bool fused_attn = true;      // DPC_UNFUSED_ATTN=1 selects the unfused reference composition (A/B tests)
    int attn_mode = 2;           // = modes.attn: f32 (0: fused attention on the fp32 MFMA) | x6 (1: bf16x6) | f16x3 (2, default)
    dpc::RangeCheck range;       // dpc_unet3d_set_range_check: f16x3 activation range check (common.h)
    dpc::DevBuf range_flag;
    dpc::DevBuf oflow;           // always-on f16x3 activation-range sentinel word (common.h: OverflowScope), read by dpc_unet3d_range_status
    bool fused_gn = true;        // DPC_UNFUSED_GN=1: standalone GroupNorm passes (3 per norm) instead of the conv-fused form
    // debug taps
    bool taps_on = false;
    struct Tap { std::unique_ptr<dpc::DevBuf> buf; size_t floats = 0; };
    std::map<std::string, Tap> taps;
};

namespace dpc {

static std::vector<std::string> expected_names(const dpc_unet3d_cfg& c, const std::vector<int>& dims) {
    std::vector<std::string> v;
    auto tattn = [&](const std::string& p) {
        v.push_back(p + ".fn.fn.fn.to_qkv.weight");
        v.push_back(p + ".fn.fn.fn.to_out.weight");
        v.push_back(p + ".fn.norm.gamma");
    };
    auto sattn = [&](const std::string& p) {
        v.push_back(p + ".fn.fn.to_qkv.weight");
        v.push_back(p + ".fn.fn.to_out.weight");
        v.push_back(p + ".fn.fn.to_out.bias");
        v.push_back(p + ".fn.norm.gamma");
    };
    auto res = [&](const std::string& p, int di, int dout, bool temb) {
        if (temb) { v.push_back(p + ".mlp.1.weight"); v.push_back(p + ".mlp.1.bias"); }
        for (const char* b : {".block1", ".block2"}) {
            v.push_back(p + b + ".proj.weight");
            v.push_back(p + b + ".proj.bias");
            v.push_back(p + b + ".norm.weight");
            v.push_back(p + b + ".norm.bias");
        }
        if (di != dout) { v.push_back(p + ".res_conv.weight"); v.push_back(p + ".res_conv.bias"); }
    };
    v.push_back("time_rel_pos_bias.relative_attention_bias.weight");
    v.push_back("init_conv.weight");
    v.push_back("init_conv.bias");
    tattn("init_temporal_attn");
    for (const char* n : {"time_mlp.1.weight", "time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias"}) v.push_back(n);
    const int nres = c.n_mults;
    for (int i = 0; i < nres; ++i) {
        const std::string p = "downs." + std::to_string(i);
        res(p + ".0", dims[i], dims[i + 1], true);
        res(p + ".1", dims[i + 1], dims[i + 1], true);
        sattn(p + ".2");
        tattn(p + ".3");
        if (i < nres - 1) { v.push_back(p + ".4.weight"); v.push_back(p + ".4.bias"); }
    }
    const int mid = dims[nres];
    res("mid_block1", mid, mid, true);
    v.push_back("mid_spatial_attn.fn.fn.fn.to_qkv.weight");
    v.push_back("mid_spatial_attn.fn.fn.fn.to_out.weight");
    v.push_back("mid_spatial_attn.fn.norm.gamma");
    tattn("mid_temporal_attn");
    res("mid_block2", mid, mid, true);
    for (int i = 0; i < nres; ++i) {
        const int di = dims[nres - 1 - i], dout = dims[nres - i];
        const std::string p = "ups." + std::to_string(i);
        res(p + ".0", dout * 2, di, true);
        res(p + ".1", di, di, true);
        sattn(p + ".2");
        tattn(p + ".3");
        if (i < nres - 1) { v.push_back(p + ".4.weight"); v.push_back(p + ".4.bias"); }
    }
    res("final_conv.0", c.dim * 2, c.dim, false);
    v.push_back("final_conv.1.weight");
    v.push_back("final_conv.1.bias");
    return v;
}

static bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// Build a PackedConv from a reference-layout Conv3d weight [N][K][kd][kh][kw].
int pack_conv3d(PackedConv& pc, const float* w, int N, int K, int kd, int kh, int kw, int sh, int sw, int pd, int ph,
                int pw, hipStream_t s) {
    const int ntaps = kd * kh * kw;
    DPC_REQUIRE(ntaps <= 32, "conv: at most 32 taps in the igemm path");
    pc.N = N; pc.K = K; pc.Npad = igemm_npad(N); pc.kchunks = igemm_kchunks(K); pc.ntaps = ntaps;
    pc.sh = sh; pc.sw = sw;
    int off[32];
    int t = 0;
    for (int a = 0; a < kd; ++a)
        for (int b = 0; b < kh; ++b)
            for (int c = 0; c < kw; ++c, ++t) {
                pc.tdf[t] = (signed char)(a - pd);
                pc.tdh[t] = (signed char)(b - ph);
                pc.tdw[t] = (signed char)(c - pw);
                off[t] = t;
            }
    pc.halo = (kd == 3 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && pd == 1 && ph == 1 && pw == 1);
    const int bk = pc.halo ? 16 : 32;
    pc.kchunks = (K + bk - 1) / bk;
    int rc = pc.wp.alloc((size_t)ntaps * pc.kchunks * pc.Npad * bk * sizeof(float));
    if (rc) return rc;
    rc = launch_pack_weights(w, pc.wp.f(), N, pc.Npad, K, ntaps, (long long)K * ntaps, ntaps, off, s, bk);
    if (rc) return rc;
    if (!pc.halo) {
        if (igemm_mode_default() == 0) return DPC_OK;
        if ((rc = pc.wp6g.alloc(igemm6_packed_bytes(pc.Npad, K, ntaps)))) return rc;
        if ((rc = launch_pack_weights_g6(w, pc.wp6g.p, N, pc.Npad, K, ntaps, (long long)K * ntaps, ntaps, off, s))) return rc;
        // the 2-D U-Net's 3x3 convs: 9-tap f16x3 pack for the big-tile halo kernel (taken when H % 8 == 0 and W % 8 == 0)
        pc.flat3 = kd == 1 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && pd == 0 && ph == 1 && pw == 1 && K % 4 == 0 &&
                   conv_mode_default() == 2;
        if (pc.flat3) {
            const int kc16 = (K + 15) / 16;
            if ((rc = pc.wp3.alloc((size_t)9 * kc16 * pc.Npad * 64))) return rc;
            return launch_pack_weights_f3(w, pc.wp3.p, N, pc.Npad, K, s, 9);
        }
        return DPC_OK;
    }
    if (conv_mode_default() == 2) {
        if ((rc = pc.wp3.alloc((size_t)27 * pc.kchunks * pc.Npad * 64))) return rc;
        if ((rc = launch_pack_weights_f3(w, pc.wp3.p, N, pc.Npad, K, s))) return rc;
        if (N % 64 == 0 && N == pc.Npad && K % 32 == 0) {
            if ((rc = pc.wpw.alloc(conv3w_packed_bytes(pc.Npad, K)))) return rc;
            return launch_pack_weights_w3(w, pc.wpw.p, N, pc.Npad, K, s);
        }
        return DPC_OK;
    }
    if ((rc = pc.wp6.alloc((size_t)27 * pc.kchunks * pc.Npad * 96))) return rc;
    return launch_pack_weights_x6(w, pc.wp6.p, N, pc.Npad, K, s);
}

bool conv2d_gn_fusable(int N, int Npad, int H, int W) {
    static const int flat_ok = debug_switch("DPC_CONV2D_HALO", 1);       // (the same switch run_conv reads: the two must not diverge)
    return flat_ok && conv_mode_default() == 2 && conv3f3c_flat_gn_ok(N, Npad, H, W);
}

bool conv_can_fuse_gn_residual(const PackedConv& pc, long long rows_per_sample) {
    static const int ok = debug_switch("DPC_FUSE_GN_RES", 1);
    return ok && !pc.halo && !pc.flat3 && igemm_mode_default() == 2 && pc.wp6g.p && pc.N % 4 == 0 && rows_per_sample % 128 == 0;
}

// One output-parity class (a,b) of ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1), weight [K][N][1][4][4]:
// out[2i+a][2j+b] = sum over the 2x2 taps (dh,kh) x (dw,kw) below of in[i+dh][j+dw] * w[:, :, kh, kw].
int pack_convT_parity(PackedConv& pc, const float* w, int K, int N, int a, int b, hipStream_t s) {
    pc.N = N; pc.K = K; pc.Npad = igemm_npad(N); pc.kchunks = igemm_kchunks(K); pc.ntaps = 4;
    pc.sh = 1; pc.sw = 1;
    const int dh_[2][2] = {{0, -1}, {1, 0}}, kh_[2][2] = {{1, 3}, {0, 2}};
    int off[32];
    int t = 0;
    for (int u = 0; u < 2; ++u)
        for (int v = 0; v < 2; ++v, ++t) {
            pc.tdf[t] = 0;
            pc.tdh[t] = (signed char)dh_[a][u];
            pc.tdw[t] = (signed char)dh_[b][v];
            off[t] = kh_[a][u] * 4 + kh_[b][v];
        }
    int rc = pc.wp.alloc((size_t)4 * pc.kchunks * pc.Npad * 32 * sizeof(float));
    if (rc) return rc;
    if ((rc = launch_pack_weights(w, pc.wp.f(), N, pc.Npad, K, 4, 16, (long long)N * 16, off, s))) return rc;
    if (igemm_mode_default() == 0) return DPC_OK;
    if ((rc = pc.wp6g.alloc(igemm6_packed_bytes(pc.Npad, K, 4)))) return rc;
    return launch_pack_weights_g6(w, pc.wp6g.p, N, pc.Npad, K, 4, 16, (long long)N * 16, off, s);
}

int run_conv(const PackedConv& pc, const float* a0, const float* a1, int C0, int C1, const float* bias,
             const float* resid, float* out, int BF, int F, int Hi, int Wi, int Ho, int Wo, const float* ln_stats,
             const float* ln_gamma, int out_mode, int par_a, int par_b, hipStream_t s, float* gn_part,
             const float* in_coef, const float* gn_raw, const float* gn_coef, float act_scale, int a0_stride) {
    DPC_REQUIRE(C0 + C1 == pc.K, "conv: channel mismatch");
    DPC_REQUIRE(!gn_raw || (!pc.halo && !pc.flat3 && igemm_mode_default() == 2 && pc.wp6g.p && gn_coef),
                "conv: the fused GroupNorm residual needs the f16x3 implicit GEMM");
    static const int flat_ok = debug_switch("DPC_CONV2D_HALO", 1);
    const bool flat_halo = flat_ok && pc.flat3 && !a0_stride && !resid && !ln_stats && out_mode == 0 && Hi == Ho && Wi == Wo && Hi % 8 == 0 &&
                           Wi % 8 == 0 && C0 % 4 == 0 && C1 % 4 == 0;
    DPC_REQUIRE(!(gn_part || in_coef) || (pc.halo && conv_mode_default() >= 1) || (flat_halo && conv2d_gn_fusable(pc.N, pc.Npad, Hi, Wi)),
                "conv: GroupNorm fusion needs the split-operand halo-tile conv path");
    if (pc.halo) {
        DPC_REQUIRE(!resid && !ln_stats && out_mode == 0 && Hi == Ho && Wi == Wo, "conv3h: plain 3x3x3 conv only");
        Conv3hParams q{};
        q.a0 = a0; q.a1 = a1; q.C0 = C0; q.C1 = C1; q.wp = pc.wp.f(); q.bias = bias; q.out = out;
        q.B = BF / F; q.F = F; q.H = Hi; q.W = Wi; q.N = pc.N; q.Npad = pc.Npad; q.kchunks = pc.kchunks;
        q.gn_part = gn_part; q.in_coef = in_coef;
        q.act_scale = act_scale;
        if (conv_mode_default() == 2) {
            q.wp = reinterpret_cast<const float*>(pc.wp3.p);
            q.wpw = pc.wpw.p;
            if (int r = range_check_note(a0, (long long)BF * Hi * Wi, C0, a1, (long long)BF * Hi * Wi, C1, in_coef, (long long)F * Hi * Wi, s))
                return r;
            return launch_conv3f3(q, s);
        }
        if (conv_mode_default() == 1) {
            q.wp = reinterpret_cast<const float*>(pc.wp6.p);
            return launch_conv3x6(q, s);
        }
        return launch_conv3h(q, s);
    }
    if (flat_halo) {
        // (1,3,3) convolution on the halo-tile kernel: frames = the BF images (no coupling), shape-only rule (any batch size)
        Conv3hParams q{};
        q.a0 = a0; q.a1 = a1; q.C0 = C0; q.C1 = C1; q.wp = reinterpret_cast<const float*>(pc.wp3.p); q.bias = bias; q.out = out;
        q.B = 1; q.F = BF; q.H = Hi; q.W = Wi; q.N = pc.N; q.Npad = pc.Npad; q.kchunks = (pc.K + 15) / 16; q.kd = 1;
        q.act_scale = act_scale;
        q.gn_part = gn_part; q.in_coef = in_coef;          // per-IMAGE GroupNorm hooks of conv3f3c's (1,3,3) form (r05)
        if (int r = range_check_note(a0, (long long)BF * Hi * Wi, C0, a1, (long long)BF * Hi * Wi, C1, in_coef, (long long)Hi * Wi, s)) return r;
        return launch_conv3f3(q, s);
    }
    IgemmParams p{};
    p.a0 = a0; p.a1 = a1; p.C0 = C0; p.C1 = C1;
    p.wp = pc.wp.f(); p.bias = bias; p.resid = resid; p.out = out;
    p.ln_stats = ln_stats; p.ln_gamma = ln_gamma;
    p.BF = BF; p.F = F; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.sh = pc.sh; p.sw = pc.sw;
    p.ntaps = pc.ntaps; p.N = pc.N; p.Npad = pc.Npad; p.kchunks = pc.kchunks;
    p.out_mode = out_mode; p.par_a = par_a; p.par_b = par_b;
    for (int i = 0; i < 32; ++i) { p.tdf[i] = pc.tdf[i]; p.tdh[i] = pc.tdh[i]; p.tdw[i] = pc.tdw[i]; }
    p.M = (long long)BF * Ho * Wo;
    p.gn_raw = gn_raw; p.gn_coef = gn_coef; p.gn_rows = (long long)F * Ho * Wo;
    p.act_scale = act_scale; p.a0_stride = a0_stride;
    if (pc.wp6g.p) {
        if (igemm_mode_default() == 2 && !ln_stats)      // (a LayerNorm prologue normalises the operand before the split)
            if (int r = range_check_note(a0, (long long)BF * Hi * Wi, C0, a1, (long long)BF * Hi * Wi, C1, nullptr, 0, s)) return r;
        return launch_igemm6(p, pc.wp6g.p, s);
    }
    return launch_igemm(p, s);
}

// ------------------------------------------------------------------------------------ forward
struct Runner {
    dpc_unet3d_s* h;
    Arena ar;
    hipStream_t s;
    int mb, F, H, W;
    int x_ctot = 0, x_coff = 0;
    float* temb = nullptr;
    int rc = DPC_OK;
    bool dry() const { return ar.dry; }

    const float* raw(const std::string& n) {
        if (dry()) return nullptr;
        auto it = h->raw.find(n);
        if (it == h->raw.end()) { rc = fail(DPC_ERR_STATE, "missing parameter " + n); return nullptr; }
        return it->second->f();
    }
    const float* raw_opt(const std::string& n) {
        if (dry()) return nullptr;
        auto it = h->raw.find(n);
        return it == h->raw.end() ? nullptr : it->second->f();
    }
    const PackedConv* conv(const std::string& n) {
        if (dry()) return nullptr;
        if (RangeCheck* rq = range_check_current()) rq->cur = n;
        auto it = h->conv.find(n);
        if (it == h->conv.end()) { rc = fail(DPC_ERR_STATE, "missing packed weight " + n); return nullptr; }
        return it->second.get();
    }
#define RUN(expr) do { if (!dry() && rc == DPC_OK) { int _r = (expr); if (_r) rc = _r; } } while (0)

    void tap(const std::string& name, const float* x_cl, int C, int Hl, int Wl) {
        if (dry() || !h->taps_on || rc) return;
        auto& t = h->taps[name];
        const size_t n = (size_t)mb * F * Hl * Wl * C;
        if (!t.buf || t.floats != n) {
            t.buf.reset(new DevBuf());
            if (t.buf->alloc(n * sizeof(float))) { rc = DPC_ERR_HIP; return; }
            t.floats = n;
        }
        RUN(launch_cl_to_cf(x_cl, t.buf->f(), mb * F, C, (long long)Hl * Wl, F, s));
    }

    // Block: conv3x3x3 -> GN -> (scale,shift) -> SiLU (+resid)   (…conv3d.py:189-204)
    void block(const std::string& p, const float* x0, const float* x1, int C0, int C1, int Cout, float* conv_out,
               float* act_out, const float* resid, const float* scale_shift, int Hl, int Wl) {
        const PackedConv* pc = conv(p + ".proj.weight");
        if (pc)
            RUN(run_conv(*pc, x0, x1, C0, C1, raw(p + ".proj.bias"), nullptr, conv_out, mb * F, F, Hl, Wl, Hl, Wl,
                         nullptr, nullptr, 0, 0, 0, s));
        const size_t m = ar.mark();
        void* ws = ar.alloc(gn_workspace_bytes(mb, Cout));
        RUN(launch_groupnorm_silu(conv_out, act_out, resid, raw(p + ".norm.weight"), raw(p + ".norm.bias"), scale_shift,
                                  mb, (long long)F * Hl * Wl, Cout, h->cfg.groups, ws, s));
        ar.release(m);
    }

    // ResnetBlock (…conv3d.py:206-230). dst == x0 (in place) is allowed when C1 == 0 and C0 == Cout.
    // Every ResnetBlock's (scale, shift) = mlp(temb) (…conv3d.py:209-212, 222-225) depends on the time embedding only: all of them in
    // one launch, in the order the traversal below consumes them (downs i.0, i.1; mid 1, 2; ups i.0, i.1).
    std::vector<float*> ss_pre;
    size_t ss_next = 0;
    void time_projections(const std::vector<int>& dims, int nres) {
        std::vector<std::pair<std::string, int>> blocks;
        for (int i = 0; i < nres; ++i)
            for (int j = 0; j < 2; ++j) blocks.push_back({"downs." + std::to_string(i) + "." + std::to_string(j), dims[i + 1]});
        blocks.push_back({"mid_block1", dims[nres]});
        blocks.push_back({"mid_block2", dims[nres]});
        for (int i = 0; i < nres; ++i)
            for (int j = 0; j < 2; ++j) blocks.push_back({"ups." + std::to_string(i) + "." + std::to_string(j), dims[nres - 1 - i]});
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

    void resnet(const std::string& p, const float* x0, const float* x1, int C0, int C1, int Cout, float* dst,
                bool has_temb, int Hl, int Wl) {
        const long long P = (long long)mb * F * Hl * Wl;
        const size_t m = ar.mark();
        float* ss = nullptr;
        if (has_temb) {
            if (ss_next < ss_pre.size()) {                 // computed up front with all the other blocks' (time_projections)
                ss = ss_pre[ss_next++];
            } else {
                ss = ar.allocf((long long)mb * 2 * Cout);
                RUN(launch_small_linear(temb, raw(p + ".mlp.1.weight"), raw(p + ".mlp.1.bias"), ss, mb, h->cfg.dim * 4,
                                        2 * Cout, 1, 0, s));
            }
        }
        const bool same = (C1 == 0 && C0 == Cout);
        if (h->fused_gn && conv_mode_default() >= 1) {
            // GroupNorm fused around the two conv3x6 launches: statistics come out of the conv epilogues, block1's
            // normalise + scale/shift + SiLU is applied inside block2's halo load (h1 never exists in HBM in activated
            // form), only block2's normalise + SiLU (+ residual) is a separate streaming pass.
            // GroupNorm partial sums per (sample, channel): 2 per 4x4x8 tile, or what the f16x3 tiling of this shape emits
            const long long tiles = conv_mode_default() == 2 ? conv3f3_gn_entries(F, Hl, Wl, Cout, igemm_npad(Cout)) / 2
                                                             : conv3x6_tiles_per_sample(F, Hl, Wl);
            const long long R = (long long)F * Hl * Wl;
            float* raw1 = ar.allocf(P * Cout);
            float* part = ar.allocf((long long)mb * tiles * 2 * Cout * 2);
            float* stats = ar.allocf((long long)mb * Cout * 2);
            float* coef = ar.allocf((long long)mb * Cout * 7);          // 5-row table + its folded 2-row form (launch_gn_finalize_fused)
            float* raw2 = (same && dst == x0) ? ar.allocf(P * Cout) : dst;      // (allocated in the dry run as well)
            const PackedConv* c1 = conv(p + ".block1.proj.weight");
            const PackedConv* c2 = conv(p + ".block2.proj.weight");
            if (c1 && c2) {
                RUN(run_conv(*c1, x0, x1, C0, C1, raw(p + ".block1.proj.bias"), nullptr, raw1, mb * F, F, Hl, Wl, Hl, Wl, nullptr,
                             nullptr, 0, 0, 0, s, part, nullptr));
                RUN(launch_gn_finalize_fused(part, mb, tiles, Cout, h->cfg.groups, R, raw(p + ".block1.norm.weight"),
                                             raw(p + ".block1.norm.bias"), ss, stats, coef, s));
                RUN(run_conv(*c2, raw1, nullptr, Cout, 0, raw(p + ".block2.proj.bias"), nullptr, raw2, mb * F, F, Hl, Wl, Hl, Wl,
                             nullptr, nullptr, 0, 0, 0, s, part, coef));
                // block2's GroupNorm + SiLU: a streaming pass with the identity residual, or -- when the block has a res_conv --
                // folded into the res_conv epilogue (block2(h) + res_conv(x) without materialising block2's activated output)
                const PackedConv* rcv = same ? nullptr : conv(p + ".res_conv.weight");
                const bool fuse_res = rcv && !dry() && conv_can_fuse_gn_residual(*rcv, R);
                if (fuse_res) {
                    RUN(launch_gn_finalize_fused(part, mb, tiles, Cout, h->cfg.groups, R, raw(p + ".block2.norm.weight"),
                                                 raw(p + ".block2.norm.bias"), nullptr, stats, coef, s));
                    RUN(run_conv(*rcv, x0, x1, C0, C1, raw(p + ".res_conv.bias"), nullptr, dst, mb * F, F, Hl, Wl, Hl, Wl, nullptr,
                                 nullptr, 0, 0, 0, s, nullptr, nullptr, raw2, coef));
                } else {
                    RUN(launch_gn_finalize_fused(part, mb, tiles, Cout, h->cfg.groups, R, nullptr, nullptr, nullptr, stats, nullptr, s));
                    RUN(launch_gn_apply(raw2, dst, same ? x0 : nullptr, stats, raw(p + ".block2.norm.weight"),
                                        raw(p + ".block2.norm.bias"), nullptr, mb, R, Cout, h->cfg.groups, s));
                    if (rcv)
                        RUN(run_conv(*rcv, x0, x1, C0, C1, raw(p + ".res_conv.bias"), dst, dst, mb * F, F, Hl, Wl, Hl, Wl,
                                     nullptr, nullptr, 0, 0, 0, s));
                }
            }
            ar.release(m);
            return;
        }
        float* h1 = ar.allocf(P * Cout);
        block(p + ".block1", x0, x1, C0, C1, Cout, h1, h1, nullptr, ss, Hl, Wl);
        if (same) {
            float* h2 = (dst == x0) ? ar.allocf(P * Cout) : dst;
            block(p + ".block2", h1, nullptr, Cout, 0, Cout, h2, dst, x0, nullptr, Hl, Wl);   // + x (identity res_conv)
        } else {
            block(p + ".block2", h1, nullptr, Cout, 0, Cout, dst, dst, nullptr, nullptr, Hl, Wl);
            const PackedConv* rcv = conv(p + ".res_conv.weight");
            if (rcv)
                RUN(run_conv(*rcv, x0, x1, C0, C1, raw(p + ".res_conv.bias"), dst, dst, mb * F, F, Hl, Wl, Hl, Wl,
                             nullptr, nullptr, 0, 0, 0, s));
        }
        ar.release(m);
    }

    // Residual(PreNorm(SpatialLinearAttention)) in place (…conv3d.py:232-257, 441)
    void spatial_linear(const std::string& p, float* x, int C, int Hl, int Wl) {
        const long long P = (long long)mb * F * Hl * Wl;
        const int HD = h->cfg.attn_heads * 32;
        if (h->fused_attn && h->attn_mode == 2 && lattn3_supported(C, h->cfg.attn_heads)) {
            LattnParams lp{};
            lp.x = x; lp.out = x; lp.gamma = raw(p + ".fn.norm.gamma"); lp.bout = raw(p + ".fn.fn.to_out.bias");
            lp.images = (long long)mb * F; lp.N = Hl * Wl;
            const float* q3 = raw(p + ".fn.fn.to_qkv.weight#h3");
            const float* o3 = raw(p + ".fn.fn.to_out.weight#h3");
            RUN(launch_lattn3(lp, reinterpret_cast<const unsigned char*>(q3), reinterpret_cast<const unsigned char*>(o3), C, s));
            return;
        }
        if (h->fused_attn && lattn_fused_supported(C, h->cfg.attn_heads)) {
            const size_t m0 = ar.mark();
            LattnParams lp{};
            lp.ctx = reinterpret_cast<float*>(ar.alloc(lattn_fused_workspace_bytes((long long)mb * F)));
            lp.x = x; lp.out = x; lp.gamma = raw(p + ".fn.norm.gamma"); lp.wqkv = raw(p + ".fn.fn.to_qkv.weight");
            lp.wout = raw(p + ".fn.fn.to_out.weight"); lp.bout = raw(p + ".fn.fn.to_out.bias");
            lp.images = (long long)mb * F; lp.N = Hl * Wl;
            RUN(launch_lattn_fused(lp, C, s));
            ar.release(m0);
            return;
        }
        const size_t m = ar.mark();
        float* stats = ar.allocf(P * 2);
        float* qkv = ar.allocf(P * 3 * HD);
        float* att = ar.allocf(P * HD);
        void* ws = ar.alloc(linattn_workspace_bytes((long long)mb * F, h->cfg.attn_heads));
        RUN(launch_ln_stats(x, stats, P, C, s));
        const PackedConv* q = conv(p + ".fn.fn.to_qkv.weight");
        const PackedConv* o = conv(p + ".fn.fn.to_out.weight");
        if (q && o) {
            RUN(run_conv(*q, x, nullptr, C, 0, nullptr, nullptr, qkv, mb * F, F, Hl, Wl, Hl, Wl, stats,
                         raw(p + ".fn.norm.gamma"), 0, 0, 0, s));
            RUN(launch_linear_attention(qkv, att, h->cfg.attn_heads, (long long)mb * F, Hl * Wl, ws, s));
            RUN(run_conv(*o, att, nullptr, HD, 0, raw(p + ".fn.fn.to_out.bias"), x, x, mb * F, F, Hl, Wl, Hl, Wl, nullptr,
                         nullptr, 0, 0, 0, s));
        }
        ar.release(m);
    }

    // Residual(PreNorm(Attention)) in place; temporal (sequence over frames per pixel, rotary + rel-pos bias) or
    // spatial (sequence over pixels per frame)   (…conv3d.py:276-352, 382, 449)
    void attention(const std::string& p, float* x, int C, int Hl, int Wl, bool temporal) {
        const long long P = (long long)mb * F * Hl * Wl;
        const int HD = h->cfg.attn_heads * 32;
        const long long HWl = (long long)Hl * Wl;
        const bool use_t3 = temporal && h->fused_attn && h->attn_mode == 2 && tattn3_supported(C, F, h->cfg.attn_heads) &&
                            (F <= 32 || h->bias_toeplitz);
        if (temporal && h->fused_attn && (use_t3 || tattn_fused_supported(C, F, h->cfg.attn_heads))) {
            TattnParams tp{};
            tp.x = x; tp.out = x; tp.gamma = raw(p + ".fn.norm.gamma"); tp.wqkv = raw(p + ".fn.fn.fn.to_qkv.weight");
            tp.wout = raw(p + ".fn.fn.fn.to_out.weight"); tp.rot_cos = h->t_cos.f(); tp.rot_sin = h->t_sin.f();
            tp.bias = h->t_bias.f(); tp.bias32 = h->t_bias32.f(); tp.npix = (long long)mb * HWl; tp.HW = HWl; tp.F = F;
            tp.brel = h->bias_toeplitz ? h->t_brel.f() : nullptr;
            const float* q6 = raw_opt(p + ".fn.fn.fn.to_qkv.weight#x6");
            const float* o6 = raw_opt(p + ".fn.fn.fn.to_out.weight#x6");
            const float* q3 = raw_opt(p + ".fn.fn.fn.to_qkv.weight#h3");
            const float* o3 = raw_opt(p + ".fn.fn.fn.to_out.weight#h3");
            if (use_t3) {
                const size_t m3 = ar.mark();
                void* ws3 = ar.alloc(tattn3_workspace_bytes(C, P));        // (allocated in the dry run as well)
                if (q3 && o3)
                    RUN(launch_tattn3(tp, reinterpret_cast<const unsigned char*>(q3), reinterpret_cast<const unsigned char*>(o3), C,
                                      ws3, s));
                ar.release(m3);
            } else if (h->attn_mode >= 1 && q6 && o6)
                RUN(launch_tattn6(tp, reinterpret_cast<const unsigned char*>(q6), reinterpret_cast<const unsigned char*>(o6), C, s));
            else
                RUN(launch_tattn_fused(tp, C, s));
            return;
        }
        const size_t m = ar.mark();
        float* stats = ar.allocf(P * 2);
        float* qkv = ar.allocf(P * 3 * HD);
        float* att = ar.allocf(P * HD);
        RUN(launch_ln_stats(x, stats, P, C, s));
        const PackedConv* q = conv(p + ".fn.fn.fn.to_qkv.weight");
        const PackedConv* o = conv(p + ".fn.fn.fn.to_out.weight");
        if (q && o) {
            RUN(run_conv(*q, x, nullptr, C, 0, nullptr, nullptr, qkv, mb * F, F, Hl, Wl, Hl, Wl, stats,
                         raw(p + ".fn.norm.gamma"), 0, 0, 0, s));
            AttnParams ap{};
            ap.qkv = qkv; ap.out = att; ap.heads = h->cfg.attn_heads;
            if (temporal) {
                ap.L = F; ap.n_seq = (long long)mb * HWl; ap.seq_inner = HWl; ap.seq_outer_stride = (long long)F * HWl;
                ap.seq_inner_stride = 1; ap.token_stride = HWl;
                ap.rot_cos = h->t_cos.f(); ap.rot_sin = h->t_sin.f(); ap.bias = h->t_bias.f();
            } else {
                ap.L = (int)HWl; ap.n_seq = (long long)mb * F; ap.seq_inner = 1; ap.seq_outer_stride = HWl;
                ap.seq_inner_stride = 0; ap.token_stride = 1;
            }
            RUN(launch_attention(ap, s));
            RUN(run_conv(*o, att, nullptr, HD, 0, nullptr, x, x, mb * F, F, Hl, Wl, Hl, Wl, nullptr, nullptr, 0, 0, 0, s));
        }
        ar.release(m);
    }

    void forward(const float* x_in, const int64_t* t_in, float* out) {
        const dpc_unet3d_cfg& c = h->cfg;
        const int dim = c.dim, nres = c.n_mults;
        const std::vector<int>& dims = h->dims;
        const long long P0 = (long long)mb * F * H * W;
        // time embedding (…conv3d.py:404-409, 509)
        float* sinemb = ar.allocf((long long)mb * dim);
        float* t1 = ar.allocf((long long)mb * dim * 4);
        temb = ar.allocf((long long)mb * dim * 4);
        // library scratch of this forward (common.h: ScratchScope): SiLU(temb) of the one-launch time projections
        const size_t scratch_bytes = (size_t)mb * dim * 4 * sizeof(float);
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
        // stem (…conv3d.py:392, 503)
        float* X0 = ar.allocf(P0 * dim);
        {
            StemParams sp{};
            sp.x = x_in; sp.wp = dry() ? nullptr : h->stem_wp->f(); sp.ktab = dry() ? nullptr : (const int*)h->stem_ktab->p;
            sp.bias = raw("init_conv.bias"); sp.out = X0; sp.BF = mb * F; sp.F = F; sp.C = c.channels; sp.H = H; sp.W = W;
            sp.Ctot = x_ctot; sp.c_off = x_coff;
            sp.N = dim; sp.Npad = h->stem_npad; sp.kchunks = h->stem_kchunks; sp.M = P0;
            if (!dry() && h->stem_wp6 && h->modes.stem == 2) {
                if (RangeCheck* rq = range_check_current()) rq->cur = "init_conv.weight";
                RUN(range_check_note(x_in, (long long)mb * F * x_ctot * H * W / 4, 4, nullptr, 0, 0, nullptr, 0, s));
            }
            if (!dry() && h->stem_wp6) RUN(launch_stem7x6(sp, h->stem_wp6->p, s));
            else RUN(launch_stem(sp, s));
        }
        tap("init_conv", X0, dim, H, W);
        attention("init_temporal_attn", X0, dim, H, W, true);
        tap("init_temporal_attn", X0, dim, H, W);

        std::vector<float*> skips;
        const float* x = X0;
        int Hl = H, Wl = W;
        for (int i = 0; i < nres; ++i) {
            const std::string p = "downs." + std::to_string(i);
            const int di = dims[i], dout = dims[i + 1];
            float* A = ar.allocf((long long)mb * F * Hl * Wl * dout);
            resnet(p + ".0", x, nullptr, di, 0, dout, A, true, Hl, Wl);
            tap(p + ".0", A, dout, Hl, Wl);
            resnet(p + ".1", A, nullptr, dout, 0, dout, A, true, Hl, Wl);
            tap(p + ".1", A, dout, Hl, Wl);
            spatial_linear(p + ".2", A, dout, Hl, Wl);
            tap(p + ".2", A, dout, Hl, Wl);
            attention(p + ".3", A, dout, Hl, Wl, true);
            tap(p + ".3", A, dout, Hl, Wl);
            skips.push_back(A);
            x = A;
            if (i < nres - 1) {
                const int Ho = (Hl + 2 - 4) / 2 + 1, Wo = (Wl + 2 - 4) / 2 + 1;
                float* D = ar.allocf((long long)mb * F * Ho * Wo * dout);
                const PackedConv* pc = conv(p + ".4.weight");
                if (pc)
                    RUN(run_conv(*pc, A, nullptr, dout, 0, raw(p + ".4.bias"), nullptr, D, mb * F, F, Hl, Wl, Ho, Wo,
                                 nullptr, nullptr, 0, 0, 0, s));
                tap(p + ".4", D, dout, Ho, Wo);
                x = D; Hl = Ho; Wl = Wo;
            }
        }
        const int mid = dims[nres];
        float* Mx = ar.allocf((long long)mb * F * Hl * Wl * mid);
        resnet("mid_block1", x, nullptr, mid, 0, mid, Mx, true, Hl, Wl);
        tap("mid_block1", Mx, mid, Hl, Wl);
        attention("mid_spatial_attn", Mx, mid, Hl, Wl, false);
        tap("mid_spatial_attn", Mx, mid, Hl, Wl);
        attention("mid_temporal_attn", Mx, mid, Hl, Wl, true);
        tap("mid_temporal_attn", Mx, mid, Hl, Wl);
        resnet("mid_block2", Mx, nullptr, mid, 0, mid, Mx, true, Hl, Wl);
        tap("mid_block2", Mx, mid, Hl, Wl);
        x = Mx;
        int xc = mid;
        for (int i = 0; i < nres; ++i) {
            const std::string p = "ups." + std::to_string(i);
            const int di = dims[nres - 1 - i], dout = dims[nres - i];
            const float* sk = skips.back(); skips.pop_back();
            float* U = ar.allocf((long long)mb * F * Hl * Wl * di);
            resnet(p + ".0", x, sk, xc, dout, di, U, true, Hl, Wl);          // cat((x, h.pop()), dim=1) is virtual
            tap(p + ".0", U, di, Hl, Wl);
            resnet(p + ".1", U, nullptr, di, 0, di, U, true, Hl, Wl);
            tap(p + ".1", U, di, Hl, Wl);
            spatial_linear(p + ".2", U, di, Hl, Wl);
            tap(p + ".2", U, di, Hl, Wl);
            attention(p + ".3", U, di, Hl, Wl, true);
            tap(p + ".3", U, di, Hl, Wl);
            x = U; xc = di;
            if (i < nres - 1) {
                const int Ho = 2 * Hl, Wo = 2 * Wl;
                float* V = ar.allocf((long long)mb * F * Ho * Wo * di);
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        const PackedConv* pc = conv(p + ".4.weight#" + std::to_string(a) + std::to_string(b));
                        if (pc)
                            RUN(run_conv(*pc, U, nullptr, di, 0, raw(p + ".4.bias"), nullptr, V, mb * F, F, Hl, Wl, Hl, Wl,
                                         nullptr, nullptr, 2, a, b, s));
                    }
                tap(p + ".4", V, di, Ho, Wo);
                x = V; Hl = Ho; Wl = Wo;
            }
        }
        // final: cat((x, r)) -> ResnetBlock(2dim -> dim, no time emb) -> 1x1x1 conv, written in the reference layout
        float* Fz = ar.allocf(P0 * dim);
        resnet("final_conv.0", x, X0, xc, dim, dim, Fz, false, Hl, Wl);
        tap("final_conv.0", Fz, dim, Hl, Wl);
        const PackedConv* pc = conv("final_conv.1.weight");
        if (pc && conv1x1_rows_supported(dim, pc->N))
            RUN(launch_conv1x1_rows(Fz, raw("final_conv.1.weight"), raw("final_conv.1.bias"), out, (long long)mb * F * Hl * Wl,
                                    (long long)Hl * Wl, dim, pc->N, s));
        else if (pc)
            RUN(run_conv(*pc, Fz, nullptr, dim, 0, raw("final_conv.1.bias"), nullptr, out, mb * F, F, Hl, Wl, Hl, Wl, nullptr,
                         nullptr, 1, 0, 0, s));
    }
#undef RUN
};

static int micro_batch_of(const dpc_unet3d_s* h, int B) {
    int mb = h->cfg.micro_batch;
    if (mb <= 0 || mb > B) mb = B;
    return mb;
}

}  // namespace dpc

using namespace dpc;

extern "C" {

int dpc_unet3d_create(const dpc_unet3d_cfg* cfg, dpc_unet3d_t* out) {
    DPC_REQUIRE(cfg && out, "unet3d_create: null argument");
    DPC_REQUIRE(cfg->attn_dim_head == 32, "unet3d: attn_dim_head must be 32");
    DPC_REQUIRE(cfg->n_mults >= 1 && cfg->n_mults <= 8, "unet3d: 1..8 resolutions");
    DPC_REQUIRE(cfg->dim % 8 == 0 && cfg->dim >= 8, "unet3d: dim must be a multiple of 8");
    DPC_REQUIRE(cfg->init_kernel % 2 == 1 && cfg->init_kernel <= 15, "unet3d: odd init kernel");
    DPC_REQUIRE(cfg->channels >= 1 && cfg->channels <= 255, "unet3d: channels");
    auto* h = new dpc_unet3d_s();
    h->cfg = *cfg;
    h->modes = modes_global();
    h->attn_mode = h->modes.attn;
    h->fused_attn = !debug_switch("DPC_UNFUSED_ATTN", 0);
    h->fused_gn = !debug_switch("DPC_UNFUSED_GN", 0);
    if (h->cfg.out_dim <= 0) h->cfg.out_dim = h->cfg.channels;
    h->dims.push_back(cfg->dim);
    for (int i = 0; i < cfg->n_mults; ++i) h->dims.push_back(cfg->dim * cfg->dim_mults[i]);
    *out = h;
    return DPC_OK;
}

void dpc_unet3d_destroy(dpc_unet3d_t h) { delete h; }

int dpc_unet3d_load(dpc_unet3d_t h, const char* name_c, const float* w, const int64_t* shape, int ndim,
                    dpc_stream_t stream) {
    DPC_REQUIRE(h && name_c && w && shape && ndim >= 1 && ndim <= 5, "unet3d_load: bad argument");
    ModeScope mode_scope(h->modes);
    hipStream_t s = (hipStream_t)stream;
    const std::string name(name_c);
    const auto exp = expected_names(h->cfg, h->dims);
    bool known = false;
    for (const auto& e : exp) if (e == name) { known = true; break; }
    DPC_REQUIRE(known, "unet3d_load: unknown parameter name '" + name + "'");
    long long numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= shape[i];
    int rc = DPC_OK;
    if (name == "init_conv.weight") {
        DPC_REQUIRE(ndim == 5 && shape[1] == h->cfg.channels && shape[2] == h->cfg.init_kernel, "init_conv.weight shape");
        const int N = (int)shape[0], C = (int)shape[1], k = (int)shape[2];
        h->stem_npad = (int)align_up(N, 64);
        h->stem_kchunks = igemm_kchunks(k * k * k * C);
        h->stem_wp.reset(new DevBuf());
        h->stem_ktab.reset(new DevBuf());
        if ((rc = h->stem_wp->alloc((size_t)h->stem_kchunks * h->stem_npad * 32 * sizeof(float)))) return rc;
        if ((rc = h->stem_ktab->alloc((size_t)h->stem_kchunks * 32 * sizeof(int)))) return rc;
        rc = launch_pack_stem(w, h->stem_wp->f(), (int*)h->stem_ktab->p, N, h->stem_npad, C, k, s);
        const bool stem_f32 = h->modes.stem == 0;
        h->stem_wp6.reset();
        if (!rc && !stem_f32 && stem7x6_supported(C, k)) {
            h->stem_wp6.reset(new DevBuf());
            if ((rc = h->stem_wp6->alloc(stem7x6_packed_bytes(h->stem_npad)))) return rc;
            rc = launch_pack_stem7x6(w, h->stem_wp6->p, N, h->stem_npad, C, s);
        }
    } else if (name.rfind("ups.", 0) == 0 && ends_with(name, ".4.weight")) {
        DPC_REQUIRE(ndim == 5 && shape[2] == 1 && shape[3] == 4 && shape[4] == 4, "ConvTranspose3d weight must be [Cin,Cout,1,4,4]");
        for (int a = 0; a < 2 && !rc; ++a)
            for (int b = 0; b < 2 && !rc; ++b) {
                auto pc = std::make_unique<PackedConv>();
                rc = pack_convT_parity(*pc, w, (int)shape[0], (int)shape[1], a, b, s);
                h->conv[name + "#" + std::to_string(a) + std::to_string(b)] = std::move(pc);
            }
    } else if (ndim == 5 && !ends_with(name, "gamma")) {
        const int kd = (int)shape[2], kh = (int)shape[3], kw = (int)shape[4];
        int sh = 1, sw = 1, pd = kd / 2, ph = kh / 2, pw = kw / 2;
        if (kd == 1 && kh == 4 && kw == 4) { sh = 2; sw = 2; pd = 0; ph = 1; pw = 1; }   // Downsample (:162-163)
        auto pc = std::make_unique<PackedConv>();
        rc = pack_conv3d(*pc, w, (int)shape[0], (int)shape[1], kd, kh, kw, sh, sw, pd, ph, pw, s);
        h->conv[name] = std::move(pc);
        if (!rc && name == "final_conv.1.weight") {          // reference layout too: the row-streaming final 1x1x1 conv (small.hip)
            auto b = std::make_unique<DevBuf>();
            if ((rc = b->alloc((size_t)numel * sizeof(float)))) return rc;
            DPC_HIP(hipMemcpyAsync(b->p, w, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
            h->raw[name] = std::move(b);
        }
    } else if (ends_with(name, "to_qkv.weight") || ends_with(name, "to_out.weight")) {
        auto pc = std::make_unique<PackedConv>();
        rc = pack_conv3d(*pc, w, (int)shape[0], (int)shape[1], 1, 1, 1, 1, 1, 0, 0, 0, s);   // Linear / Conv2d 1x1: [N][K]
        h->conv[name] = std::move(pc);
        auto b = std::make_unique<DevBuf>();                 // reference layout too (fused attention kernel)
        if (!rc && (rc = b->alloc((size_t)numel * sizeof(float)))) return rc;
        DPC_HIP(hipMemcpyAsync(b->p, w, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice, s));
        h->raw[name] = std::move(b);
        // pre-split per-head images for the bf16x6 fused attention kernels (inner dim 4 heads x 32)
        const bool is_out = ends_with(name, "to_out.weight");
        const int C = is_out ? (int)shape[0] : (int)shape[1];
        const int inner = is_out ? (int)shape[1] : (int)shape[0] / 3;
        if (!rc && inner == 128 && (C == 64 || C == 128)) {
            auto b6 = std::make_unique<DevBuf>();
            if ((rc = b6->alloc(is_out ? attn6_out_bytes(C) : attn6_qkv_bytes(C)))) return rc;
            rc = launch_pack_attn6(w, reinterpret_cast<unsigned char*>(b6->p), C, is_out, s);
            h->raw[name + "#x6"] = std::move(b6);
        }
        if (!rc && inner == 128 && (C == 64 || C == 128)) {     // weight-stationary f16x3 attention kernels (tattn3.hip, lattn3.hip)
            auto b3 = std::make_unique<DevBuf>();
            if ((rc = b3->alloc(is_out ? tattn3_out_bytes(C) : tattn3_qkv_bytes(C)))) return rc;
            rc = launch_pack_tattn3(w, reinterpret_cast<unsigned char*>(b3->p), C, is_out, s);
            h->raw[name + "#h3"] = std::move(b3);
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

int dpc_unet3d_set_tables(dpc_unet3d_t h, int frames, const float* bias, const float* rc_, const float* rs_,
                          const float* freqs, dpc_stream_t stream) {
    DPC_REQUIRE(h && frames >= 1 && bias && rc_ && rs_ && freqs, "unet3d_set_tables: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int heads = h->cfg.attn_heads;
    int rc;
    if ((rc = h->t_bias.alloc((size_t)heads * frames * frames * 4))) return rc;
    if ((rc = h->t_cos.alloc((size_t)frames * 32 * 4))) return rc;
    if ((rc = h->t_sin.alloc((size_t)frames * 32 * 4))) return rc;
    if ((rc = h->t_freq.alloc((size_t)(h->cfg.dim / 2) * 4))) return rc;
    DPC_HIP(hipMemcpyAsync(h->t_bias.p, bias, h->t_bias.bytes, hipMemcpyDeviceToDevice, s));
    DPC_HIP(hipMemcpyAsync(h->t_cos.p, rc_, h->t_cos.bytes, hipMemcpyDeviceToDevice, s));
    DPC_HIP(hipMemcpyAsync(h->t_sin.p, rs_, h->t_sin.bytes, hipMemcpyDeviceToDevice, s));
    DPC_HIP(hipMemcpyAsync(h->t_freq.p, freqs, h->t_freq.bytes, hipMemcpyDeviceToDevice, s));
    if (heads == 4 && frames <= 32) {
        if ((rc = h->t_bias32.alloc((size_t)4 * 32 * 32 * 4))) return rc;
        DPC_HIP(hipMemsetAsync(h->t_bias32.p, 0, h->t_bias32.bytes, s));
        for (int hd = 0; hd < 4; ++hd)
            DPC_HIP(hipMemcpy2DAsync(h->t_bias32.f() + hd * 1024, 32 * 4, bias + (size_t)hd * frames * frames, (size_t)frames * 4,
                                     (size_t)frames * 4, frames, hipMemcpyDeviceToDevice, s));
    }
    // Toeplitz form of the bias for the long-sequence fused attention (tattn3.hip): the reference's table is
    // emb[bucket(j - i)][h] (...conv3d.py:106-112); verify that on the host (set-up time, one small copy) instead of assuming it
    h->bias_toeplitz = false;
    if (heads == 4 && frames <= 64) {
        std::vector<float> hb((size_t)heads * frames * frames), tz((size_t)4 * 128, 0.f);
        DPC_HIP(hipMemcpyAsync(hb.data(), bias, hb.size() * 4, hipMemcpyDeviceToHost, s));
        DPC_HIP(hipStreamSynchronize(s));
        bool ok = true;
        for (int hd = 0; hd < heads; ++hd) {
            const float* b = hb.data() + (size_t)hd * frames * frames;
            for (int r = -(frames - 1); r <= frames - 1; ++r) tz[hd * 128 + r + 63] = r >= 0 ? b[r] : b[(size_t)(-r) * frames];
            for (int i = 0; i < frames && ok; ++i)
                for (int j = 0; j < frames; ++j)
                    if (b[(size_t)i * frames + j] != tz[hd * 128 + (j - i) + 63]) { ok = false; break; }
        }
        if (ok) {
            if ((rc = h->t_brel.alloc(tz.size() * 4))) return rc;
            DPC_HIP(hipMemcpyAsync(h->t_brel.p, tz.data(), tz.size() * 4, hipMemcpyHostToDevice, s));
            DPC_HIP(hipStreamSynchronize(s));
            h->bias_toeplitz = true;
        }
    }
    h->frames = frames;
    return DPC_OK;
}

int dpc_unet3d_finalize(dpc_unet3d_t h) {
    DPC_REQUIRE(h, "unet3d_finalize: null handle");
    for (const auto& e : expected_names(h->cfg, h->dims))
        if (!h->loaded.count(e)) return fail(DPC_ERR_STATE, "unet3d_finalize: parameter not loaded: " + e);
    if (int rc = f16x3_weight_overflow_check("unet3d_finalize")) return rc;
    h->finalized = true;
    return DPC_OK;
}

size_t dpc_unet3d_workspace_bytes(dpc_unet3d_t h, int B, int F, int H, int W) {
    if (!h || B <= 0) return 0;
    if (h->ws_need && h->ws_key[0] == B && h->ws_key[1] == F && h->ws_key[2] == H && h->ws_key[3] == W) return h->ws_need;
    ModeScope mode_scope(h->modes);
    Runner r{};
    r.h = h; r.s = nullptr; r.mb = micro_batch_of(h, B); r.F = F; r.H = H; r.W = W;
    r.ar.dry = true;
    r.forward(nullptr, nullptr, nullptr);
    h->ws_key[0] = B; h->ws_key[1] = F; h->ws_key[2] = H; h->ws_key[3] = W;
    h->ws_need = r.ar.peak + 256;
    return h->ws_need;
}

int dpc_unet3d_forward(dpc_unet3d_t h, const float* x, int x_channels_total, int x_channel_offset, const int64_t* t,
                       float* out, int B, int F, int H, int W, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(h && x && t && out, "unet3d_forward: null argument");
    if (!h->finalized) return fail(DPC_ERR_STATE, "unet3d_forward: call dpc_unet3d_finalize first");
    ModeScope mode_scope(h->modes);
    if (h->frames != F) return fail(DPC_ERR_STATE, "unet3d_forward: tables were set for a different frame count");
    const int levels = h->cfg.n_mults - 1;
    DPC_REQUIRE(H % (1 << levels) == 0 && W % (1 << levels) == 0, "unet3d_forward: H, W must be divisible by 2^(levels-1)");
    if (B == 0) return DPC_OK;
    const int mb = micro_batch_of(h, B);
    const size_t need = dpc_unet3d_workspace_bytes(h, B, F, H, W);
    if (ws_bytes < need || !ws) return fail(DPC_ERR_STATE, "unet3d_forward: workspace too small: need " + std::to_string(need));
    if (x_channels_total <= 0) { x_channels_total = h->cfg.channels; x_channel_offset = 0; }
    DPC_REQUIRE(x_channel_offset >= 0 && x_channel_offset + h->cfg.channels <= x_channels_total, "unet3d_forward: bad channel view");
    const long long in_per = (long long)F * x_channels_total * H * W, out_per = (long long)F * h->cfg.out_dim * H * W;
    DPC_REQUIRE(!h->range.on || ((long long)H * W) % 4 == 0, "unet3d_forward: the range check needs H*W % 4 == 0");
    RangeCheckScope range_scope(h->range.on ? &h->range : nullptr);
    const bool f16x3_any = h->modes.conv == 2 || h->modes.igemm == 2;          // the range only exists in the f16x3 mode
    if (f16x3_any && !h->oflow.p) {
        if (int rc = h->oflow.alloc(4)) return rc;
        DPC_HIP(hipMemsetAsync(h->oflow.p, 0, 4, (hipStream_t)stream));
    }
    OverflowScope oflow_scope(f16x3_any ? reinterpret_cast<int*>(h->oflow.p) : nullptr);
    if (h->range.on) {
        h->range.names.clear();
        DPC_HIP(hipMemsetAsync(h->range_flag.p, 0x7f, 4, (hipStream_t)stream));
    }
    for (int b0 = 0; b0 < B; b0 += mb) {
        Runner r{};
        r.h = h; r.s = (hipStream_t)stream; r.mb = std::min(mb, B - b0); r.F = F; r.H = H; r.W = W;
        r.x_ctot = x_channels_total; r.x_coff = x_channel_offset;
        r.ar.dry = false;
        r.ar.base = reinterpret_cast<char*>(align_up((size_t)ws, 256));
        r.ar.cap = ws_bytes - (size_t)(r.ar.base - (char*)ws);
        r.forward(x + b0 * in_per, t + b0, out + b0 * out_per);
        if (r.rc) return r.rc;
        if (r.ar.overflow) return fail(DPC_ERR_STATE, "unet3d_forward: arena overflow");
    }
    if (h->range.on) {                       // opt-in debugging aid: one host sync per forward
        int first = 0;
        DPC_HIP(hipMemcpyAsync(&first, h->range_flag.p, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
        DPC_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (first != 0x7f7f7f7f) {
            const std::string nm = (first >= 1 && first <= (int)h->range.names.size()) ? h->range.names[first - 1] : "?";
            return fail(DPC_ERR_STATE, "unet3d_forward: f16x3 activation range exceeded (|x| > 4094 or non-finite) at the input of '" +
                                           nm + "': this arithmetic mode would clamp it; create the model with arithmetic mode x6 or f32");
        }
    }
    return DPC_OK;
}

int dpc_unet3d_range_status(dpc_unet3d_t h, int reset, dpc_stream_t stream) {
    DPC_REQUIRE(h, "null handle");
    if (!h->oflow.p) return DPC_OK;
    int v = 0;
    DPC_HIP(hipMemcpyAsync(&v, h->oflow.p, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    DPC_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (v && reset) DPC_HIP(hipMemsetAsync(h->oflow.p, 0, 4, (hipStream_t)stream));
    if (v)
        return fail(DPC_ERR_STATE, "unet3d: an activation on the residual stream left the range the f16x3 arithmetic represents (|x| > 4094 "
                                   "or non-finite) during a forward since the last status check: results computed from it are clamped / "
                                   "invalid; create the model with arithmetic mode x6 or f32 (dpc_unet3d_set_range_check names the op)");
    return DPC_OK;
}

int dpc_unet3d_set_range_check(dpc_unet3d_t h, int enable) {
    DPC_REQUIRE(h, "null handle");
    if (enable && !h->range_flag.p)
        if (int rc = h->range_flag.alloc(4)) return rc;
    h->range.flag = reinterpret_cast<int*>(h->range_flag.p);
    h->range.on = enable != 0;
    return DPC_OK;
}

const char* dpc_unet3d_modes(dpc_unet3d_t h) {
    static thread_local std::string buf;
    buf = h ? modes_string(h->modes) : std::string();
    return buf.c_str();
}

int dpc_unet3d_debug_taps(dpc_unet3d_t h, int enable) {
    DPC_REQUIRE(h, "null handle");
    h->taps_on = enable != 0;
    if (!enable) h->taps.clear();
    return DPC_OK;
}

int dpc_unet3d_get_tap(dpc_unet3d_t h, const char* name, float* dst, size_t dst_floats, dpc_stream_t stream) {
    DPC_REQUIRE(h && name && dst, "get_tap: null argument");
    auto it = h->taps.find(name);
    if (it == h->taps.end() || !it->second.buf) return fail(DPC_ERR_STATE, std::string("get_tap: no such tap ") + name);
    DPC_REQUIRE(dst_floats >= it->second.floats, "get_tap: destination too small");
    DPC_HIP(hipMemcpyAsync(dst, it->second.buf->p, it->second.floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DPC_OK;
}

}  // extern "C"
