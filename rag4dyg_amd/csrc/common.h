// Shared helpers for the gfx950 kernels behind include/r4d.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/r4d.h"

namespace r4d {

void set_error(const char* fmt, ...);
// 1 when the translation unit was built with its kernel-ablation macro set (tools/kc_ablate.sh): r4d_build_flags()
int dbgflag_kc(); int dbgflag_att(); int dbgflag_sk(); int dbgflag_jac(); int dbgflag_scan(); int dbgflag_s3(); int dbgflag_h2();

#define R4D_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            r4d::set_error(__VA_ARGS__);           \
            return R4D_ERR_INVALID;                \
        }                                          \
    } while (0)

#define R4D_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            r4d::set_error("%s: %s", #call, hipGetErrorString(e_));                    \
            return R4D_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define R4D_CHECK_LAUNCH(name)                                                         \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            r4d::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return R4D_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

// ------------------------------------------------------------------ launch profiler (profile.hip)
enum ProfClass {
    PK_GEMM_128x128_NN = 0, PK_GEMM_128x128_NT, PK_GEMM_128x64_NN, PK_GEMM_128x64_NT, PK_GEMM_64x64_NN,
    PK_GEMM_64x64_NT, PK_GEMM_KC_128x128x32, PK_GEMM_KC_128x128x16, PK_GEMM_KC_128x64x16, PK_GEMM_KC_64x64x32, PK_GEMM_S3_128x256, PK_GEMM_S3_128x128, PK_GEMM_S3TN, PK_GEMM_H2_128x256, PK_GEMM_H2_128x128, PK_GEMM_SKINNY, PK_GEMM_SKINNY_EPI,
    PK_EMBED_LN, PK_LAYERNORM, PK_SOFTMAX, PK_DECODE_ATTN, PK_GREEDY_ADVANCE,
    PK_ATTN_FUSED, PK_LNF_MEANPOOL, PK_MEANPOOL_REDUCE, PK_NORMALIZE, PK_POOL_SCAN, PK_TOPK, PK_MERGE_TOPK, PK_RANK_COUNT, PK_JACCARD, PK_JACCARD_PREP, PK_COUNT
};
extern bool g_prof_on;
void prof_begin_impl(int cls, double work, hipStream_t s);
void prof_end_impl(hipStream_t s);
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(int cls, double work, hipStream_t st) : s(st), on(g_prof_on) { if (on) prof_begin_impl(cls, work, s); }
    ~ProfScope() { if (on) prof_end_impl(s); }
};

// ------------------------------------------------------------------ dispatcher-branch coverage (profile.hip)
// Every host-side decision that selects a kernel VARIANT (tile shape, template instantiation, split-K form, fallback) counts
// its launches under a name; r4d_dispatch_* enumerates the table so that a test can assert that each branch was exercised
// (tests/test_gpu_ops.py::test_every_dispatcher_branch_is_exercised -- a shape list keyed by config names missed the
// d = 256 decode path once: commit 2e9f5d4).  Names that start with "tuning:" are reachable through environment switches only.
#define R4D_BRANCH_LIST(X)                                                                                              \
    X(KC_128x128x16, "gemm_kc:128x128x16") X(KC_128x64x16, "gemm_kc:128x64x16") X(KC_64x64x32, "gemm_kc:64x64x32")       \
    X(KC_128x128x32, "tuning:gemm_kc:128x128x32") X(KC_ROWSPLIT, "gemm_kc:row-split (two launches)")                      \
    X(S3_128x256, "gemm_s3:128x256x32") X(S3_PERSISTENT, "gemm_s3:128x256x32 persistent (pipeline across tiles)") X(S3_128x128, "gemm_s3:128x128x32") X(S3_F32B, "gemm_s3:both operands split on the fly (scoring GEMM)") X(S3_TN, "gemm_s3tn:128x256x32 (weight gradients, transposing LDS reads)")      \
    X(H2_128x256, "gemm_h2:128x256x32 (f16x2)") X(H2_128x128, "gemm_h2:128x128x32 (f16x2)") X(H2P_128x256, "gemm_h2p:128x256x32 (f16x2, A as lines, LDS-DMA)") X(H2P_128x128, "gemm_h2p:128x128x32 (f16x2, A as lines, LDS-DMA)") \
    X(F32_128x128, "gemm_f32:128x128") X(F32_128x64, "gemm_f32:128x64") X(F32_64x64, "gemm_f32:64x64")                   \
    X(F32_NT, "gemm_f32:B as [N,K]") X(F32_NN, "gemm_f32:B as [K,N]") X(TN_SPLITK, "gemm_tn:split-K") X(TN_SINGLE, "gemm_tn:one slice") \
    X(SK16_NG2, "skinny16:ng2") X(SK16_NG3, "skinny16:ng3") X(SK16_NG2_LN, "skinny16:ng2+layernorm") X(SK16_LN_FOLDED, "skinny16:LayerNorm pre-folded into the weight") X(SK16_NG3_LN, "skinny16:ng3+layernorm") \
    X(SK16_SPLITK, "skinny16:split-K last-arriver") X(SK8_NG2, "skinny8:ng2") X(SK8_NG3, "skinny8:ng3")                    \
    X(SK8_LN, "tuning:skinny8:+layernorm") X(SK8_SPLITK, "skinny8:split-K + epilogue launch") X(SK_PLAIN, "skinny:plain + epilogue launch") \
    X(ATT_KS32, "attention:key-split hd32") X(ATT_KS64, "attention:key-split hd64") X(ATT_CS96, "attention:column-split hd96") \
    X(ATT_CS128, "attention:column-split hd128") X(ATT_CS256, "attention:column-split hd256") X(ATT_H2_KS32, "attention:f16x2 key-split hd32") X(ATT_H2_KS64, "attention:f16x2 key-split hd64") X(ATT_H2_KS96, "attention:f16x2 key-split hd96") X(ATT_H2_128, "attention:f16x2 hd128") X(ATT_H2_256, "attention:f16x2 hd256") X(ATT_H2_KS32_KBLK, "attention:f16x2 key-split hd32, key-blocked K") X(ATT_H2_KS64_KBLK, "attention:f16x2 key-split hd64, key-blocked K") X(ATT_H2_KS96_KBLK, "attention:f16x2 key-split hd96, key-blocked K") X(ATT_H2_128_KBLK, "attention:f16x2 hd128, key-blocked K") X(ATT_H2_256_KBLK, "attention:f16x2 hd256, key-blocked K")                              \
    X(ATT_KS_FORCED, "tuning:attention:key-split at hd96/128/256") X(ATT_3LAUNCH, "attention:three-launch GEMM form")      \
    X(SCAN_1_1, "scan:d32") X(SCAN_2_1, "scan:d64") X(SCAN_4_1, "scan:d128") X(SCAN_4_2, "scan:d256") X(SCAN_4_3, "scan:d384") \
    X(SCAN_8_2, "scan:d512") X(SCAN_4_4, "tuning:scan:d512 4-way") X(SCAN_8_3, "scan:d768") X(SCAN_8_4, "scan:d1024")       \
    X(SCAN_SHORT, "scan:short shard (even rows, two tiles in flight)") X(SCAN_DMA, "scan:short shard, LDS-DMA staged") X(SCAN_RING, "scan:long shard, LDS-DMA ring") X(SCAN_GEMM, "scan:tiled GEMM (Q > 64 or other d)") X(SCAN_TILED, "scan:128x256 tiles in the scan kernels' arithmetic (Q >= 64)") X(SCAN_BF16X3, "scan:bf16x3 operands") X(SCAN_F32, "scan:exact-f32 operands")  \
    X(TOPK_ONE_WG, "topk:one workgroup per row") X(TOPK_TICKET, "topk:cross-workgroup ticket merge")                      \
    X(TOPK_MULTI, "topk:second launch over candidates") X(TOPK_F64, "topk:f64 rows")                                      \
    X(EMBED_LN4, "embed+layernorm:16-byte lanes") X(EMBED_GENERIC, "embed+layernorm:generic") X(LN4_2, "layernorm:ln4<2>") X(LN4_4, "layernorm:ln4<4>") X(LN4_8, "layernorm:ln4<8>") X(LN_GENERIC, "layernorm:generic") \
    X(LNF_8, "lnf_meanpool:<8>") X(LNF_16, "lnf_meanpool:<16>") X(LNF_32, "lnf_meanpool:<32>")                             \
    X(DEC_ATT_32, "decode_attention:<32>") X(DEC_ATT_64, "decode_attention:<64>")                                         \
    X(JAC_LDS, "jaccard:LDS table") X(JAC_MERGE, "jaccard:merge walk (vocab too large for LDS)")                          \
    X(JAC_PREP_DENSE_LDS, "jaccard_prep:dense tokens, LDS histogram") X(JAC_PREP_DENSE_GLOBAL, "jaccard_prep:dense tokens, global histogram") X(JAC_PREP_ORDER, "jaccard_prep:rows longest first") \
    X(ARGSORT_ONE, "argsort:one chunk") X(ARGSORT_MULTI, "argsort:chunk sort + rank scatter")
enum DispatchBranch {
#define X(id, name) BR_##id,
    R4D_BRANCH_LIST(X)
#undef X
    BR_COUNT
};
extern unsigned long long g_branch_hits[BR_COUNT];
#define R4D_BRANCH(id) (++r4d::g_branch_hits[r4d::BR_##id])

// range guard (include/r4d.h: r4d_set_range_flag)
extern unsigned* g_range_flag;         // bits: R4D_RANGE_NONFINITE_HIDDEN, R4D_RANGE_BAD_NORM

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ fp32 MFMA GEMM (gemm_f32.hip)
enum GemmEpilogue { EPI_NONE = 0, EPI_GELU = 1, EPI_RESIDUAL = 2, EPI_SCALE_DIV = 3, EPI_HALF_PLUS = 4,
                    // gemm_s3 only (training): GELU_KEEP writes gelu(v) to C and the pre-activation v to the `resid` buffer;
                    // GELU_GRAD writes v * gelu'(u) with u read from the `resid` buffer
                    EPI_GELU_KEEP = 5, EPI_GELU_GRAD = 6,
                    // gemm_h2 only: C receives the result as uint32 "h2 words" (fp16 hi | fp16 lo' << 16 of value / 4: csrc/h2.h), the
                    // operand format of attention_h2.hip
                    EPI_H2WORDS = 7 };
enum GemmCausal { CAUSAL_NONE = 0, CAUSAL_QK = 1, CAUSAL_PV = 2 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;      // [N] or null
    const float* resid;     // [M,N] (ldr) or null
    int M, N, K;            // K = full contraction length (CAUSAL_PV trims it per row tile)
    int lda, ldb, ldc, ldr;
    int b_trans;            // 0: B is [K,N] (ldb >= N)   1: B is [N,K] (ldb >= K)
    int b_rows;             // valid rows of B (guards loads): K for NN, N for NT
    int a_cols;             // valid columns of A (guards loads); 0 -> K
    int nbatch, nb1;        // batch z = z0 * nb1 + z1
    long long sA0, sA1, sB0, sB1, sC0, sC1;   // element strides per batch index
    int epilogue; float scale_div;
    int causal;
};
int launch_gemm_f32(const GemmArgs& g, hipStream_t stream);
int launch_gemm_f32_kc(const GemmArgs& g, hipStream_t stream);      // B given as [N,K]: k-contiguous kernel
// C[M,N] = A[Kt,M]^T . B[Kt,N] (weight gradients), split-K partials in scratch (gemm_tn_scratch_floats floats)
size_t gemm_tn_scratch_floats(int M, int N, int Kt);
// `colsum_out` (nullable, with `colsum_scratch` of `colsum_scratch_floats` floats): ALSO the column sums of B over its Kt rows
// (the bias gradient), when the bf16x3 kernel takes the call; *colsum_done says whether they were written
int launch_gemm_f32_tn(const float* A, const float* B, float* C, int M, int N, int Kt, int lda, int ldb, float* scratch,
                       hipStream_t stream, float* colsum_out = nullptr, float* colsum_scratch = nullptr,
                       size_t colsum_scratch_floats = 0, bool* colsum_done = nullptr);
// gemm_skinny.hip: M <= 32 rows against a k-contiguous weight [N,K], K % 256 == 0 (decode step); split-K partials in scratch
bool gemm_skinny_supported(int M, int K, int N);
size_t gemm_skinny_scratch_floats(int K, int N);
bool gemm_skinny_fuses_ln(int M, int K, int N);       // y = epilogue(LayerNorm(x) . wT^T + bias) in one launch
// `counters_zeroed`: the caller cleared the split-K ticket counters (gemm_skinny_counters) on this stream since the last
// aborted launch (a completed launch leaves them at zero); otherwise the launcher clears them itself when K is split
void* gemm_skinny_counters(float* scratch, size_t* bytes);
int launch_gemm_skinny(const float* x, const float* wT, const float* bias, const float* resid, int M, int K, int N,
                       int epilogue, float* y, float* scratch, hipStream_t s, const float* ln_w = nullptr,
                       const float* ln_b = nullptr, float ln_eps = 0.f, bool counters_zeroed = false, const float* ln_fold = nullptr);
int launch_fold_layernorm(const float* wT, const float* g, const float* beta, int N, int K, float* wTg, float* lnc, hipStream_t s);

// gemm_s3.hip: C = epilogue(A . W^T + bias) on the bf16 matrix cores at fp32 accuracy (A fp32, split on the fly into three
// bf16 terms; W given as three pre-split bf16 planes [3][N][K]); K % 32 == 0
struct S3Args {
    const float* A; const unsigned short* planes; float* C;
    const float* bias; const float* resid;
    int M, N, K, lda, ldc, ldr, epilogue;
    unsigned* kblk; int kb_hd;        // gemm_h2p, EPI_H2WORDS, N = 3 d: columns [d, 2d) go to the key-blocked K image instead (kb_hd = head_dim; 0 = off)
};
bool gemm_s3_supported(int M, int K, int N);
int launch_gemm_s3(const S3Args& a, hipStream_t stream);
bool gemm_s3_f32b_supported(int M, int K, int N);
int launch_gemm_s3_f32b(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldc, int epilogue, hipStream_t stream);
int launch_gemm_s3_scan_order(const float* q_hat, const float* pool_hat, float* scores, int Q, int N, int d, int ng, hipStream_t stream);
// w element (n, k) at w[k * ld_k + n * ld_n] -> planes [3][N][K] bf16 (hi, mid, lo)
// gemm_h2.hip: the same contract on the fp16 matrix cores, two fp16 terms per operand and THREE products (planes [2][N][K])
bool gemm_h2_supported(int M, int K, int N);
int launch_gemm_h2(const S3Args& a, hipStream_t stream);
int launch_split2_planes(const float* w, int N, int K, long long ld_k, long long ld_n, unsigned short* planes, hipStream_t s);
// gemm_h2p.hip: the same arithmetic with the A operand already split by its producer into f16x2 lines [M][K/32][2][32] fp16 (both
// tiles staged by LDS-DMA); out_lines: the GELU epilogue writes lines [M][N/32][2][32] fp16 into a.C.  a.A is ignored.
extern int g_gemm_h2p;
bool gemm_h2p_supported(int M, int K, int N);
int launch_gemm_h2p(const S3Args& a, const unsigned short* a_lines, bool out_lines, hipStream_t stream);
int launch_split2_lines(const float* x, long long rows, int K, unsigned short* lines, hipStream_t s);
extern int g_gemm_split3;             // Conv1D arithmetic (r4d_set_gemm_split3): 0 exact-f32 MFMA, 1 bf16x3 planes, 2 f16x2 planes (bf16x3 where a layer carries no f16 planes)
// Conv1D dispatch shared by the encoder and the training forward (encoder.hip): skinny weight stream (decode), bf16x3 planes,
// k-contiguous copy, reference layout -- in that order of preference
int conv1d(const float* x, const float* w, const float* wT, const float* bias, const float* resid, int M, int K, int N,
           int epilogue, float* y, hipStream_t s, float* skinny_scratch = nullptr, bool sk_counters_zeroed = false,
           const unsigned short* w3 = nullptr, const unsigned short* w2h = nullptr);
// gemm_s3tn.hip: C[I,J] = X[M,I]^T . dY[M,J] on the bf16 matrix cores at fp32 accuracy, split over the token rows
bool gemm_s3tn_supported(int I, int J, int M, int lda, int ldb);
int gemm_s3tn_slices(int I, int J, int M, int max_slices);
int launch_gemm_s3tn(const float* X, const float* dY, float* out, int I, int J, int M, int lda, int ldb, int S, int* slices_out,
                     hipStream_t stream, float* db_partials = nullptr);
int launch_split3_planes(const float* w, int N, int K, long long ld_k, long long ld_n, unsigned short* planes, hipStream_t s);

// ------------------------------------------------------------------ train_ops.hip
struct DropKey { unsigned seed_lo, seed_hi, step_lo, step_hi; };      // Philox key (seed) and counter words 2-3 (step)
// out = (resid ? resid : 0) + dropout_p(x): element i uses counter (base + i) / 4 of `site`; out may alias x / resid
int launch_dropout(const float* x, const float* resid, long long n, float* out, float p, DropKey key, unsigned site,
                   unsigned long long base, hipStream_t s);

// ------------------------------------------------------------------ topk.hip
// rows x n values -> rows x k best (value, global index), canonical order; `counters_zeroed`: the caller already cleared
// the ticket counters (first rows * 4 bytes of ws) on this stream
template <typename T> size_t topk_ws_bytes(int rows, int n, int k);
template <typename T>
int topk_rows(const T* vals, const long long* idx_in, int rows, int n, long long ld, int k, long long index_offset, T* out_v,
              long long* out_i, void* ws, size_t ws_bytes, bool counters_zeroed, hipStream_t s);

// ------------------------------------------------------------------ encoder_ops.hip
int launch_layernorm(const float* x, const float* w, const float* b, int rows, int d, float eps, float* y,
                     hipStream_t s);
bool layernorm_lines_supported(int d);
int launch_layernorm_lines(const float* x, const float* w, const float* b, int rows, int d, float eps, unsigned short* y_lines, hipStream_t s);
constexpr int ATT_MAXG = 16;
// Up to ATT_MAXG right-padded batches handled by ONE launch of a row kernel (each batch keeps its own T): rows of the
// batches are consecutive in x / y, sequences consecutive in the pooled output.  ids[g] / emb[g]: exactly one is set.
struct RowGroups {
    int n;
    int B[ATT_MAXG], T[ATT_MAXG];
    const int64_t* ids[ATT_MAXG];
    const float* emb[ATT_MAXG];
};
int launch_embed_layernorm_groups(const RowGroups& G, const float* wte, const float* wpe, int vocab, int d,
                                  const float* w, const float* b, float eps, float* x_out, float* y_out, hipStream_t s, bool y_lines = false);
int launch_causal_softmax(float* S, int nbh, int T, int ld, int row_tile, hipStream_t s);
// decode step (one new position per sequence): x = (ids ? wte[id] : emb) + wpe[pos], y = LayerNorm(x)
int launch_embed_pos_layernorm(const int64_t* ids, const float* emb, const int32_t* pos, const float* wte,
                               const float* wpe, int vocab, int n_positions, int t_cap, int B, int d, const float* w,
                               const float* b, float eps, float* x_out, float* y_out, hipStream_t s,
                               void* zero_words = nullptr, size_t zero_bytes = 0);   // also clears zero_bytes at zero_words
// one query per (sequence, head) against the cached keys/values; writes the new K/V row into the cache first
int launch_decode_attention(const float* qkv_new, float* kv_layer, const int32_t* pos, int B, int t_cap, int H, int d,
                            float* out, hipStream_t s);
// device-side greedy loop state (encoder_ops.hip: greedy_advance_kernel; r4d.h: r4d_greedy_state)
struct GreedyState {
    int64_t* next; int32_t* lens; int32_t* pos; int32_t* active; int32_t* gen_len; int32_t* out_tokens;
    const int32_t* params;       // device int32[8]: max_gen, len_limit, n_eos, eos[0..3], 0
    int out_cap, t_cap;
};
// optional second job of the bookkeeping kernel: the decode step's input rows x[b,:] = wte[next[b]] + wpe[pos[b]] and the
// clearing of `n_zero` 32-bit words (the step's split-K ticket counters)
struct GreedyEmbed { const float* wte; const float* wpe; float* x_out; unsigned* zero_words; int vocab, n_positions, d, n_zero; };
int launch_greedy_advance(const float* logits, int B, int V, const GreedyState& st, hipStream_t s, const GreedyEmbed* embed = nullptr);
// Up to ATT_MAXG right-padded batches (each with its own T) in ONE attention launch: the workgroup index walks the sequences
// of all batches; the batch of a sequence is found by a short scan of the prefix table (kernel argument, by value).
struct AttnGroups {
    int n;
    int seq_prefix[ATT_MAXG + 1];     // first global sequence index of each batch
    int T[ATT_MAXG];
    long long row0[ATT_MAXG];         // first token row of each batch in qkv / out
};
// attention_h2.hip: the same attention on the fp16 matrix cores (f16x2 form), qkv as uint32 "h2 words" (csrc/h2.h; written by
// gemm_h2's EPI_H2WORDS epilogue or launch_pack_h2_words); head_dim 128 / 256 (attention_h2_supported)
bool attention_h2_supported(int H, int d);
int launch_attention_h2_groups(const unsigned* qkv_words, int n, const int* Bs, const int* Ts, const long long* row0s, int H, int d,
                               float* out, hipStream_t s, bool out_lines = false, const unsigned* kblk = nullptr);
// the key-blocked K image of h2 words (attention_h2.hip: attn_h2_kernel<HD, true>): [ceil(M / 32)][H][hd / 8][2][32][4] uint32
int launch_pack_kblk_words(const unsigned* qkv_words, long long M, int H, int d, unsigned* kblk, hipStream_t s);
static inline size_t kblk_words(size_t M, int d) { return (M + 31) / 32 * 32 * (size_t)d; }
int launch_pack_h2_words(const float* x, long long n, unsigned* words, hipStream_t s);
int dbgflag_att_h2();
// attention_fused.hip: R4D_OK / error, or +1 when head_dim has no fused instantiation
int launch_attention_fused(const float* qkv, int B, int T, int H, int d, float* out, hipStream_t s);
int launch_attention_fused_groups(const float* qkv, int n, const int* Bs, const int* Ts, const long long* row0s, int H,
                                  int d, float* out, hipStream_t s);
extern int g_attention_variant;
extern int g_attention_fused;        // -1 auto (default), 1 fused, 0 three-launch GEMM form (r4d_set_attention_fused)
constexpr int LNF_ROWS_PER_CHUNK = 16;
size_t lnf_meanpool_scratch_floats(int B, int T, int d);
// x / hidden_out: first row of the first batch; pool_out: first sequence of the first batch; scratch: the sum over the
// batches of lnf_meanpool_scratch_floats
int launch_lnf_meanpool_groups(const RowGroups& G, const float* x, const float* w, const float* b, int d, float eps,
                               float* hidden_out, float* pool_out, float* scratch, hipStream_t s);

}  // namespace r4d
