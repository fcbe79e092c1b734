// Opt-in per-kernel-class timing with HIP events recorded on the launch stream (bench.py's roofline leg).
// Disabled by default: a disabled ProfScope costs one branch and records nothing.
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace dpc {

static const char* kClassNames[PROF_NCLASS] = {
    "igemm_bn64", "igemm_bn128", "stem_gather", "groupnorm_silu", "ln_stats", "attention_core",
    "linear_attention", "ddpm_update", "small_ops", "burgers_fd", "philox_normal", "smoke_eval", "conv3h_bn64",
    "conv3h_bn128", "temporal_attention_fused",
    "linear_attention_fused", "conv3x6_bn64", "conv3x6_bn128", "conv_wgrad", "attention_bwd", "train_misc", "conv3_wgrad_f16x3"};

struct ProfState {
    bool on = false;
    unsigned long long mask = ~0ull;         // classes that are instrumented
    struct Rec { int cls; hipEvent_t a, b; double flops, bytes; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_next = 0;
    std::mutex mu;
};
static ProfState g_prof;

static hipEvent_t prof_event() {
    if (g_prof.pool_next == g_prof.pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_prof.pool.push_back(e);
    }
    return g_prof.pool[g_prof.pool_next++];
}

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s) : idx_(-1), s_(s) {
    if (!g_prof.on || !((g_prof.mask >> cls) & 1ull)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.recs.size() >= (1u << 20)) return;
    hipEvent_t a = prof_event(), b = prof_event();
    if (!a || !b) return;
    (void)hipEventRecord(a, s);
    g_prof.recs.push_back({cls, a, b, flops, bytes});
    idx_ = (long long)g_prof.recs.size() - 1;
}

ProfScope::~ProfScope() {
    if (idx_ < 0) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    (void)hipEventRecord(g_prof.recs[(size_t)idx_].b, s_);
}

}  // namespace dpc

using namespace dpc;

extern "C" {

int dpc_profile_begin(void) { return dpc_profile_begin_classes(nullptr); }

int dpc_profile_begin_classes(const char* class_names) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    unsigned long long mask = 0;
    if (class_names && class_names[0]) {
        const std::string all = std::string(",") + class_names + ",";
        for (int i = 0; i < PROF_NCLASS; ++i)
            if (all.find(std::string(",") + kClassNames[i] + ",") != std::string::npos) mask |= 1ull << i;
        DPC_REQUIRE(mask != 0, "profile_begin_classes: no known class name in '" + std::string(class_names) + "'");
    } else {
        mask = ~0ull;
    }
    g_prof.recs.clear();
    g_prof.pool_next = 0;
    g_prof.mask = mask;
    g_prof.on = true;
    return DPC_OK;
}

int dpc_profile_end(dpc_profile_row* rows, int max_rows, int* n_rows) {
    DPC_REQUIRE(rows && n_rows && max_rows >= 1, "profile_end: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    dpc_profile_row acc[PROF_NCLASS];
    for (int i = 0; i < PROF_NCLASS; ++i) {
        acc[i].name = kClassNames[i];
        acc[i].launches = 0;
        acc[i].total_ms = acc[i].flops = acc[i].bytes = 0.0;
    }
    for (auto& r : g_prof.recs) {
        DPC_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        DPC_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        acc[r.cls].launches += 1;
        acc[r.cls].total_ms += ms;
        acc[r.cls].flops += r.flops;
        acc[r.cls].bytes += r.bytes;
    }
    int n = 0;
    for (int i = 0; i < PROF_NCLASS && n < max_rows; ++i)
        if (acc[i].launches) rows[n++] = acc[i];
    *n_rows = n;
    g_prof.recs.clear();
    g_prof.pool_next = 0;
    return DPC_OK;
}

}  // extern "C"
