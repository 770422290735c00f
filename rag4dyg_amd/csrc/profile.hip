// Per-launch HIP-event timing for bench.py's roofline object (include/r4d.h "Measurement hooks").
#include <vector>
#include "common.h"

namespace r4d {

bool g_prof_on = false;

struct Rec { hipEvent_t a, b; int cls; double work; };
static std::vector<Rec> g_recs;        // live records
static std::vector<Rec> g_free;        // recycled event pairs
static Rec g_open;

void prof_begin_impl(int cls, double work, hipStream_t s) {
    Rec r;
    if (!g_free.empty()) { r = g_free.back(); g_free.pop_back(); }
    else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
    r.cls = cls; r.work = work;
    (void)hipEventRecord(r.a, s);
    g_open = r;
}
void prof_end_impl(hipStream_t s) {
    (void)hipEventRecord(g_open.b, s);
    g_recs.push_back(g_open);
}

static const char* kNames[PK_COUNT] = {
    "gemm_f32_128x128_nn", "gemm_f32_128x128_nt", "gemm_f32_128x64_nn",
    "gemm_f32_128x64_nt", "gemm_f32_64x64_nn", "gemm_f32_64x64_nt", "gemm_f32_kc_128x128x32", "gemm_f32_kc_128x128x16",
    "gemm_f32_kc_128x64x16", "gemm_f32_kc_64x64x32", "gemm_s3_128x256x32", "gemm_s3_128x128x32", "gemm_s3tn_128x256x32", "gemm_h2_128x256x32", "gemm_h2_128x128x32", "gemm_skinny", "gemm_skinny_epilogue", "embed_layernorm",
    "layernorm", "causal_softmax", "decode_attention", "greedy_advance", "attn_fused", "lnf_partial", "meanpool_reduce", "normalize_rows", "pool_scan", "topk", "merge_topk", "argsort", "jaccard", "jaccard_prep"};

unsigned long long g_branch_hits[BR_COUNT];
static const char* kBranchNames[BR_COUNT] = {
#define X(id, name) name,
    R4D_BRANCH_LIST(X)
#undef X
};

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_dispatch_num_branches(void) { return BR_COUNT; }
const char* r4d_dispatch_branch_name(int32_t i) { return (i >= 0 && i < BR_COUNT) ? kBranchNames[i] : ""; }
int64_t r4d_dispatch_branch_hits(int32_t i) { return (i >= 0 && i < BR_COUNT) ? (int64_t)g_branch_hits[i] : -1; }
int r4d_dispatch_reset(void) { for (int i = 0; i < BR_COUNT; ++i) g_branch_hits[i] = 0; return R4D_OK; }

int r4d_profile_enable(int32_t on) {
    for (auto& r : g_recs) g_free.push_back(r);
    g_recs.clear();
    g_prof_on = on != 0;
    return R4D_OK;
}
int r4d_profile_num_classes(void) { return PK_COUNT; }
const char* r4d_profile_class_name(int32_t cls) { return (cls >= 0 && cls < PK_COUNT) ? kNames[cls] : ""; }

int r4d_profile_read(int32_t cls, double* total_ms, int64_t* launches, double* work) {
    R4D_REQUIRE(cls >= 0 && cls < PK_COUNT && total_ms && launches && work, "profile_read: bad arguments");
    double ms = 0, w = 0;
    int64_t n = 0;
    for (auto& r : g_recs) {
        if (r.cls != cls) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("profile_read: event sync failed"); return R4D_ERR_HIP; }
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { set_error("profile_read: elapsed failed"); return R4D_ERR_HIP; }
        ms += t; w += r.work; ++n;
    }
    *total_ms = ms; *launches = n; *work = w;
    return R4D_OK;
}

}  // extern "C"
