// Retriever TRAINING step on gfx950: the encoder forward that KEEPS what the backward pass needs, and the
// backward pass itself -- the calculus torch autograd performs for the reference on the tensors of
// models/modeling_gpt2.py:140-235,400-509 when train/train_retriever.py:196 calls loss.backward().
//
//   forward  (r4d_gpt2_train_forward_f32):  per block  x -> ln_1 -> c_attn -> causal attention (probabilities kept) ->
//            c_proj + x -> ln_2 -> c_fc (pre-activation kept) -> gelu_new -> c_proj + x;  ln_f -> mean over T
//   backward (r4d_gpt2_train_backward_f32): d(mean-pooled embeddings) -> gradients of every parameter
// Dense contractions run on the exact-f32 MFMA GEMMs of the forward path: data gradients dX = dY . W^T read the Conv1D
// weight [in,out] as the k-contiguous B operand directly; weight gradients dW = X^T . dY read BOTH operands row by row over
// the contracted token index (the [K,M] x [K,N] form of gemm_f32_kernel, split over the tokens so that a d x d gradient still
// fills the chip, partials summed in a fixed order); attention backward is four batched per-head GEMMs around a row kernel
// (dP = dO . V^T,  dS = P (dP - rowsum(P dP)) / sqrt(hd),  dQ = dS . K,  dK = dS^T . Q,  dV = P^T . dO).
// All batches of a step (anchor, positive, negative and the two augmented views) form ONE launch sequence over their
// concatenated rows, like the inference path.  Dropout (embeddings, attention probabilities, both residual branches) draws
// its masks from a counter-based generator (train_ops.hip), so the backward regenerates them instead of storing them; the
// reference's torch RNG stream cannot be reproduced, the distribution and the calculus are the same.  Checked against the reference's autograd gradients (tests/test_gpu_training.py).
#include <math.h>
#include <string.h>
#include <vector>
#include <stdlib.h>
#include "common.h"

namespace r4d {

// train_ops.hip
size_t ln_bwd_scratch_floats(int rows, int d);
int launch_ln_bwd(const float* x, const float* w, const float* dy, const float* add, int rows, int d, float eps, float* dx,
                  float* dw, float* db, float* scratch, int accumulate, hipStream_t s);
size_t colsum_scratch_floats(long long rows, int n);
int launch_colsum(const float* x, long long rows, int n, int ld, float* out, float* scratch, int accumulate, hipStream_t s);
int launch_gelu_fwd(const float* pre, long long n, float* y, hipStream_t s);
static int g_train_fuse_gelu = -1;    // tuning aid: R4D_TRAIN_FUSE_GELU=0 keeps the two element-wise GELU launches (read once, below)
int launch_gelu_bwd(const float* pre, const float* dy, long long n, float* dx, hipStream_t s);
int launch_softmax_bwd(const float* P, float* dP, int nbh, int T, int ld, float scale_div, hipStream_t s);
int launch_transpose(const float* in, int rows, int cols, long long ld_in, long long stride_in, float* out, long long ld_out,
                     long long stride_out, int nbatch, hipStream_t s);
int launch_embedding_bwd(const float* dx, const int64_t* ids, int B, int T, int d, int vocab, unsigned long long* acc_wte, float* dwpe,
                         int first_group, long long table_rows, hipStream_t s);
int launch_embedding_fix_to_f32(const unsigned long long* acc, long long n, long long table_rows, float* out, hipStream_t s);
int launch_embedding_absmax(const float* dx, long long n, unsigned long long* acc_wte, long long table_elems, hipStream_t s);
int launch_meanpool_bwd(const float* d_pool, long long rows, int T, int d, float* dh, hipStream_t s);

static inline int tpad128(int T) { return (T + 127) / 128 * 128; }
static inline int up4(long long x) { return (int)((x + 3) / 4 * 4); }

struct TrainGroup { const int64_t* ids; int B, T; size_t row0, seq0, p0; };     // p0: offset of the group's P block (floats)

// Workspace layout, identical in the size query, the forward and the backward (bump allocation in a fixed order)
struct TrainLayout {
    size_t M, Ptot, pmax;
    int L, d;
    // per layer (offsets in floats)
    std::vector<size_t> x_in, ln1, qkv, att, x_mid, ln2, pre, f, P;
    size_t x_out, pool_scratch;
    // backward temporaries
    size_t dx, dy, dbig, dqkv, xT, dP, PT, red;
    size_t emb_acc;                                     // [vocab, d] 64-bit fixed-point token-gradient table (two floats per entry)
    size_t total;
};

static TrainLayout layout(const r4d_gpt2_config* cfg, const TrainGroup* gs, int n) {
    TrainLayout t;
    t.L = cfg->n_layer; t.d = cfg->n_embd;
    const size_t d = t.d;
    t.M = 0; t.Ptot = 0; t.pmax = 0;
    size_t pool = 0;
    for (int g = 0; g < n; ++g) {
        t.M += (size_t)gs[g].B * gs[g].T;
        const size_t pf = (size_t)gs[g].B * cfg->n_head * gs[g].T * tpad128(gs[g].T);
        t.Ptot += pf;
        if (pf > t.pmax) t.pmax = pf;
        pool += lnf_meanpool_scratch_floats(gs[g].B, gs[g].T, t.d);
    }
    size_t off = 0;
    auto take = [&](size_t nfloat) { const size_t o = off; off += (nfloat + 63) / 64 * 64; return o; };
    for (int l = 0; l < t.L; ++l) {
        t.x_in.push_back(take(t.M * d)); t.ln1.push_back(take(t.M * d)); t.qkv.push_back(take(t.M * 3 * d));
        t.att.push_back(take(t.M * d)); t.x_mid.push_back(take(t.M * d)); t.ln2.push_back(take(t.M * d));
        t.pre.push_back(take(t.M * 4 * d)); t.f.push_back(take(t.M * 4 * d)); t.P.push_back(take(t.Ptot));
    }
    t.x_out = take(t.M * d);
    t.pool_scratch = take(pool);
    t.dx = take(t.M * d); t.dy = take(t.M * d); t.dbig = take(t.M * 4 * d); t.dqkv = take(t.M * 3 * d);
    {   // split-K partials of the four weight-gradient shapes
        size_t sk = gemm_tn_scratch_floats(4 * t.d, t.d, (int)t.M);
        const size_t o[3] = {gemm_tn_scratch_floats(t.d, 4 * t.d, (int)t.M), gemm_tn_scratch_floats(t.d, t.d, (int)t.M),
                             gemm_tn_scratch_floats(t.d, 3 * t.d, (int)t.M)};
        for (size_t v : o) if (v > sk) sk = v;
        t.xT = take(sk);
    }
    t.dP = take(t.pmax); t.PT = take(t.pmax);
    size_t red = ln_bwd_scratch_floats((int)t.M, t.d);
    const size_t cs = colsum_scratch_floats((long long)t.M, 4 * t.d);
    if (cs > red) red = cs;
    t.red = take(red);
    t.emb_acc = take((size_t)cfg->vocab * d * 2 + 4);   // + two 64-bit words behind the table: poison, max |contribution|
    t.total = off;
    return t;
}

static int check_groups(const r4d_gpt2_config* cfg, int n_groups, const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts,
                        std::vector<TrainGroup>& gs) {
    R4D_REQUIRE(cfg && cfg->n_layer >= 1 && cfg->n_embd % 64 == 0 && cfg->n_embd <= 2048 && cfg->n_head >= 1 &&
                cfg->n_embd % cfg->n_head == 0 && (cfg->n_embd / cfg->n_head) % 16 == 0, "gpt2 train: bad config");
    R4D_REQUIRE(n_groups >= 1 && n_groups <= ATT_MAXG && ids_d && Bs && Ts, "gpt2 train: 1..%d batches per step", ATT_MAXG);
    gs.resize(n_groups);
    size_t row0 = 0, seq0 = 0, p0 = 0;
    for (int g = 0; g < n_groups; ++g) {
        R4D_REQUIRE(ids_d[g] && Bs[g] >= 1 && Ts[g] >= 1 && Ts[g] <= cfg->n_positions && Ts[g] <= 1024,
                    "gpt2 train: bad batch %d (B=%d T=%d)", g, Bs[g], Ts[g]);
        R4D_REQUIRE((long long)Bs[g] * cfg->n_head <= 65535, "gpt2 train: B * n_head = %lld exceeds the batched-GEMM limit",
                    (long long)Bs[g] * cfg->n_head);
        gs[g] = TrainGroup{ids_d[g], Bs[g], Ts[g], row0, seq0, p0};
        row0 += (size_t)Bs[g] * Ts[g]; seq0 += (size_t)Bs[g];
        p0 += (size_t)Bs[g] * cfg->n_head * Ts[g] * tpad128(Ts[g]);
    }
    R4D_REQUIRE(row0 <= 0x7fffffff / (size_t)(16 * cfg->n_embd), "gpt2 train: %zu rows in one step is too many", row0);
    return R4D_OK;
}

// y[M,N] = epilogue(x[M,K] . W[K,N] + bias)   (Conv1D.forward, modeling_utils.py:1267-1271).  `wT` (nullable): the caller's
// CURRENT [N,K] copy of the weight (refreshed after every optimizer step) -> the k-contiguous kernel; else the reference layout
static int fwd_linear(const float* x, const float* w, const float* wT, const float* bias, const float* resid, int M, int K, int N,
                      int epi, float* y, hipStream_t s, const unsigned short* w3 = nullptr, const unsigned short* w2h = nullptr) {
    return conv1d(x, w, wT, bias, resid, M, K, N, epi, y, s, nullptr, false, w3, w2h);     // f16x2 planes (mode 2) > bf16x3 planes > [N,K] copy > reference layout
}
// dx[M,K] = dy[M,N] . W[K,N]^T : W's rows are k(N)-contiguous, i.e. W IS the [N' = K, K' = N] operand of the fast kernel;
// `w3t` (nullable): its bf16x3 planes [3][K][N] -> the bf16 matrix cores at fp32 accuracy
// `gelu_pre` (bf16x3 path only): dx = (dy . W^T) * gelu_new'(gelu_pre), gelu_pre [M,K]
// (The data gradients stay on bf16x3 in EVERY split mode: their A operand is a GRADIENT -- 1e-5 .. 1e-8 per element in a real run -- and
//  the f16x2 form has fp16's exponent range: below 2.4e-4 an element's absolute error stops shrinking (6e-11), i.e. 1e-4 relative at
//  1e-6.  Built and measured in round 5 (46.9 instead of 50.6 ms per step, G8 green at its max-norm bounds), then taken out: a
//  per-tensor power-of-two scale from an absmax pass would be needed to make it safe, and that pass costs what the kernel saves.)
static int bwd_data(const float* dy, const float* w, int M, int K, int N, float* dx, hipStream_t s, const unsigned short* w3t = nullptr,
                    const float* gelu_pre = nullptr) {
    if (w3t && g_gemm_split3 && gemm_s3_supported(M, N, K)) {
        S3Args a;
        memset(&a, 0, sizeof(a));
        a.A = dy; a.planes = w3t; a.C = dx; a.M = M; a.N = K; a.K = N; a.lda = N; a.ldc = K; a.ldr = K;
        a.epilogue = gelu_pre ? EPI_GELU_GRAD : EPI_NONE; a.resid = gelu_pre;
        return launch_gemm_s3(a, s);
    }
    R4D_REQUIRE(!gelu_pre, "bwd_data: the fused GELU derivative needs the bf16x3 planes");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dy; g.B = w; g.C = dx;
    g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = N; g.ldc = K;
    g.b_trans = 1; g.b_rows = K; g.nbatch = 1; g.nb1 = 1; g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    return launch_gemm_f32(g, s);
}
// dW[K,N] = x[M,K]^T . dy[M,N]: both operands read row by row over the contracted token index (split-K partials in `skp`);
// db[N] = column sums of dy
static int bwd_weight(const float* x, const float* dy, int M, int K, int N, float* dW, float* db, float* skp, float* red,
                      hipStream_t s) {
    bool db_done = false;                                             // the bf16x3 kernel sums dy's columns while it stages them
    const int rc = launch_gemm_f32_tn(x, dy, dW, K, N, M, K, N, skp, s, db, red, colsum_scratch_floats(M, N), &db_done);
    if (rc) return rc;
    return (db && !db_done) ? launch_colsum(dy, M, N, N, db, red, 0, s) : R4D_OK;
}

// Attention._attn forward over one batch, probabilities kept in P [B*H, T, ld] with EVERY column right of the diagonal zero
struct DropCtx {                                                  // dropout of one step; p == 0 everywhere -> identity
    float embd_p, attn_p, resid_p;
    DropKey key;
    bool on() const { return embd_p > 0.f || attn_p > 0.f || resid_p > 0.f; }
};
static int drop_ctx(const r4d_train_dropout* dp, DropCtx& c) {
    c = DropCtx{0.f, 0.f, 0.f, DropKey{0, 0, 0, 0}};
    if (!dp) return R4D_OK;
    R4D_REQUIRE(dp->embd_p >= 0.f && dp->embd_p < 1.f && dp->attn_p >= 0.f && dp->attn_p < 1.f && dp->resid_p >= 0.f && dp->resid_p < 1.f,
                "gpt2 train: dropout probabilities must be in [0, 1)");
    c.embd_p = dp->embd_p; c.attn_p = dp->attn_p; c.resid_p = dp->resid_p;
    c.key = DropKey{(unsigned)dp->seed, (unsigned)(dp->seed >> 32), (unsigned)dp->step, (unsigned)(dp->step >> 32)};
    return R4D_OK;
}

// `Pdrop` (with attn_p > 0): scratch for the dropped-out probabilities the P.V product reads; P keeps the softmax output
static int attn_fwd(const float* qkv, int B, int T, int H, int d, float* P, float* out, hipStream_t s, float attn_p = 0.f,
                    DropKey key = DropKey{0, 0, 0, 0}, unsigned site = 0, unsigned long long pbase = 0, float* Pdrop = nullptr) {
    const int hd = d / H, ld = tpad128(T);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = qkv; g.B = qkv + d; g.C = P;
    g.M = T; g.N = T; g.K = hd; g.lda = 3 * d; g.ldb = 3 * d; g.ldc = ld;
    g.b_trans = 1; g.b_rows = T; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = (long long)T * 3 * d; g.sA1 = hd; g.sB0 = g.sA0; g.sB1 = hd;
    g.sC0 = (long long)H * T * ld; g.sC1 = (long long)T * ld;
    g.epilogue = EPI_SCALE_DIV; g.scale_div = (float)sqrt((double)hd); g.causal = CAUSAL_QK;
    int rc = launch_gemm_f32(g, s);
    if (rc) return rc;
    if ((rc = launch_causal_softmax(P, B * H, T, ld, ld, s))) return rc;          // row_tile = ld: zero-fill the whole row
    const float* Pv = P;
    if (attn_p > 0.f) {                                                            // attn_dropout(w), modeling_gpt2.py:153
        if ((rc = launch_dropout(P, nullptr, (long long)B * H * T * ld, Pdrop, attn_p, key, site, pbase, s))) return rc;
        Pv = Pdrop;
    }
    memset(&g, 0, sizeof(g));
    g.A = Pv; g.B = qkv + 2 * d; g.C = out;
    g.M = T; g.N = hd; g.K = T; g.lda = ld; g.ldb = 3 * d; g.ldc = d;
    g.b_trans = 0; g.b_rows = T; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = (long long)H * T * ld; g.sA1 = (long long)T * ld;
    g.sB0 = (long long)T * 3 * d; g.sB1 = hd; g.sC0 = (long long)T * d; g.sC1 = hd;
    g.a_cols = ld; g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_PV;
    return launch_gemm_f32(g, s);
}

// Attention backward over one batch: dao [B,T,d] (merged heads) -> dqkv [B,T,3d]
static int attn_bwd(const float* qkv, const float* P, const float* dao, int B, int T, int H, int d, float* dqkv, float* dP,
                    float* PT, hipStream_t s, float attn_p = 0.f, DropKey key = DropKey{0, 0, 0, 0}, unsigned site = 0,
                    unsigned long long pbase = 0) {
    const int hd = d / H, ld = tpad128(T), Tp = up4(T);
    const long long sP0 = (long long)H * T * ld, sP1 = (long long)T * ld;
    const long long sQ0 = (long long)T * 3 * d, sO0 = (long long)T * d;
    GemmArgs g;
    int rc;
    // dP = dO . V^T
    memset(&g, 0, sizeof(g));
    g.A = dao; g.B = qkv + 2 * d; g.C = dP;
    g.M = T; g.N = T; g.K = hd; g.lda = d; g.ldb = 3 * d; g.ldc = ld;
    g.b_trans = 1; g.b_rows = T; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = sO0; g.sA1 = hd; g.sB0 = sQ0; g.sB1 = hd; g.sC0 = sP0; g.sC1 = sP1;
    g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    if ((rc = launch_gemm_f32(g, s))) return rc;
    if (attn_p > 0.f &&                                          // through attn_dropout: d(softmax out) = mask * dP / (1 - p)
        (rc = launch_dropout(dP, nullptr, (long long)B * H * T * ld, dP, attn_p, key, site, pbase, s))) return rc;
    // dS (in place), logits were divided by sqrt(hd) before the softmax
    if ((rc = launch_softmax_bwd(P, dP, B * H, T, ld, (float)sqrt((double)hd), s))) return rc;
    // dQ = dS . K
    memset(&g, 0, sizeof(g));
    g.A = dP; g.B = qkv + d; g.C = dqkv;
    g.M = T; g.N = hd; g.K = Tp; g.lda = ld; g.ldb = 3 * d; g.ldc = 3 * d;
    g.b_trans = 0; g.b_rows = T; g.a_cols = Tp; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = sP0; g.sA1 = sP1; g.sB0 = sQ0; g.sB1 = hd; g.sC0 = sQ0; g.sC1 = hd;
    g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    if ((rc = launch_gemm_f32(g, s))) return rc;
    // dK = dS^T . Q
    if ((rc = launch_transpose(dP, T, T, ld, sP1, PT, ld, sP1, B * H, s))) return rc;
    memset(&g, 0, sizeof(g));
    g.A = PT; g.B = qkv; g.C = dqkv + d;
    g.M = T; g.N = hd; g.K = Tp; g.lda = ld; g.ldb = 3 * d; g.ldc = 3 * d;
    g.b_trans = 0; g.b_rows = T; g.a_cols = Tp; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = sP0; g.sA1 = sP1; g.sB0 = sQ0; g.sB1 = hd; g.sC0 = sQ0; g.sC1 = hd;
    g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    if ((rc = launch_gemm_f32(g, s))) return rc;
    // dV = P^T . dO  (the probabilities the forward multiplied V with: after dropout; dP's buffer is free by now)
    const float* Pv = P;
    if (attn_p > 0.f) {
        if ((rc = launch_dropout(P, nullptr, (long long)B * H * T * ld, dP, attn_p, key, site, pbase, s))) return rc;
        Pv = dP;
    }
    if ((rc = launch_transpose(Pv, T, T, ld, sP1, PT, ld, sP1, B * H, s))) return rc;
    memset(&g, 0, sizeof(g));
    g.A = PT; g.B = dao; g.C = dqkv + 2 * d;
    g.M = T; g.N = hd; g.K = Tp; g.lda = ld; g.ldb = d; g.ldc = 3 * d;
    g.b_trans = 0; g.b_rows = T; g.a_cols = Tp; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = sP0; g.sA1 = sP1; g.sB0 = sO0; g.sB1 = hd; g.sC0 = sQ0; g.sC1 = hd;
    g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    return launch_gemm_f32(g, s);
}

static RowGroups row_groups_of(const std::vector<TrainGroup>& gs) {
    RowGroups R;
    R.n = (int)gs.size();
    for (int j = 0; j < ATT_MAXG; ++j) {
        const bool in = j < R.n;
        R.B[j] = in ? gs[j].B : 0; R.T[j] = in ? gs[j].T : 0;
        R.ids[j] = in ? gs[j].ids : nullptr; R.emb[j] = nullptr;
    }
    return R;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

size_t r4d_weight_grad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features) {
    if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
    return (gemm_tn_scratch_floats(in_features, out_features, rows) + colsum_scratch_floats(rows, out_features) + 64) * sizeof(float);
}

int r4d_weight_grad_f32(const float* x_d, const float* dy_d, int32_t rows, int32_t in_features, int32_t out_features, float* dw_d,
                        float* db_d, void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(x_d && dy_d && dw_d && rows > 0 && in_features > 0 && out_features > 0, "weight_grad: bad arguments");
    R4D_REQUIRE(workspace_d && workspace_bytes >= r4d_weight_grad_workspace_bytes(rows, in_features, out_features),
                "weight_grad: workspace too small");
    float* skp = (float*)workspace_d;
    float* red = skp + (gemm_tn_scratch_floats(in_features, out_features, rows) + 63) / 64 * 64;
    return bwd_weight(x_d, dy_d, rows, in_features, out_features, dw_d, db_d, skp, red, (hipStream_t)stream);
}

size_t r4d_gpt2_train_workspace_bytes(const r4d_gpt2_config* cfg, int32_t n_groups, const int32_t* Bs, const int32_t* Ts) {
    if (!cfg || n_groups <= 0 || n_groups > ATT_MAXG || !Bs || !Ts) return 0;
    std::vector<TrainGroup> gs((size_t)n_groups);
    for (int g = 0; g < n_groups; ++g) {
        if (Bs[g] <= 0 || Ts[g] <= 0) return 0;
        gs[g] = TrainGroup{nullptr, Bs[g], Ts[g], 0, 0, 0};
    }
    return layout(cfg, gs.data(), n_groups).total * sizeof(float) + 256;
}

int r4d_gpt2_train_forward_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                               const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts, float* out_meanpool_d,
                               const r4d_train_dropout* dropout, void* workspace_d, size_t workspace_bytes, void* stream) {
    if (g_train_fuse_gelu < 0) { const char* e = getenv("R4D_TRAIN_FUSE_GELU"); g_train_fuse_gelu = e ? atoi(e) : 1; }
    hipStream_t s = (hipStream_t)stream;
    std::vector<TrainGroup> gs;
    int rc = check_groups(cfg, n_groups, ids_d, Bs, Ts, gs);
    if (rc) return rc;
    DropCtx dc;
    if ((rc = drop_ctx(dropout, dc))) return rc;
    R4D_REQUIRE(w && w->wte && w->wpe && w->ln_f_w && w->ln_f_b && w->layers && out_meanpool_d, "gpt2 train: null pointer");
    const TrainLayout t = layout(cfg, gs.data(), n_groups);
    if (!workspace_d || workspace_bytes < t.total * sizeof(float)) {
        set_error("gpt2 train: workspace %zu bytes < required %zu", workspace_bytes, t.total * sizeof(float));
        return R4D_ERR_WORKSPACE;
    }
    float* ws = (float*)workspace_d;
    const int d = cfg->n_embd, H = cfg->n_head, M = (int)t.M;
    const RowGroups R = row_groups_of(gs);
    for (int l = 0; l < cfg->n_layer; ++l) {
        const r4d_gpt2_layer& Lw = w->layers[l];
        R4D_REQUIRE(Lw.ln_1_w && Lw.c_attn_w && Lw.attn_proj_w && Lw.ln_2_w && Lw.c_fc_w && Lw.mlp_proj_w, "gpt2 train: null weight in layer %d", l);
        float *x_in = ws + t.x_in[l], *ln1 = ws + t.ln1[l], *qkv = ws + t.qkv[l], *att = ws + t.att[l];
        float *x_mid = ws + t.x_mid[l], *ln2 = ws + t.ln2[l], *pre = ws + t.pre[l], *f = ws + t.f[l];
        if (l == 0)
            rc = launch_embed_layernorm_groups(R, w->wte, w->wpe, cfg->vocab, d, Lw.ln_1_w, Lw.ln_1_b, cfg->ln_eps, x_in, ln1, s);
        else
            rc = launch_layernorm(x_in, Lw.ln_1_w, Lw.ln_1_b, M, d, cfg->ln_eps, ln1, s);
        if (rc) return rc;
        if (l == 0 && dc.embd_p > 0.f) {                             // self.drop(inputs_embeds + position_embeds), :427
            if ((rc = launch_dropout(x_in, nullptr, (long long)M * d, x_in, dc.embd_p, dc.key, R4D_DROPOUT_SITE_EMBD, 0, s))) return rc;
            if ((rc = launch_layernorm(x_in, Lw.ln_1_w, Lw.ln_1_b, M, d, cfg->ln_eps, ln1, s))) return rc;
        }
        if ((rc = fwd_linear(ln1, Lw.c_attn_w, Lw.c_attn_wT, Lw.c_attn_b, nullptr, M, d, 3 * d, EPI_NONE, qkv, s, Lw.c_attn_w3, Lw.c_attn_h2))) return rc;
        for (const TrainGroup& G : gs)
            if ((rc = attn_fwd(qkv + G.row0 * 3 * d, G.B, G.T, H, d, ws + t.P[l] + G.p0, att + G.row0 * d, s, dc.attn_p, dc.key,
                               4u * l + 0u, G.p0, ws + t.dP))) return rc;
        float* branch = ws + t.dy;                                   // a backward temporary, free during the forward
        if (dc.resid_p > 0.f) {                                      // x + resid_dropout(c_proj(a)), :194,229
            if ((rc = fwd_linear(att, Lw.attn_proj_w, Lw.attn_proj_wT, Lw.attn_proj_b, nullptr, M, d, d, EPI_NONE, branch, s, Lw.attn_proj_w3, Lw.attn_proj_h2))) return rc;
            if ((rc = launch_dropout(branch, x_in, (long long)M * d, x_mid, dc.resid_p, dc.key, 4u * l + 1u, 0, s))) return rc;
        } else if ((rc = fwd_linear(att, Lw.attn_proj_w, Lw.attn_proj_wT, Lw.attn_proj_b, x_in, M, d, d, EPI_RESIDUAL, x_mid, s, Lw.attn_proj_w3, Lw.attn_proj_h2))) return rc;
        if ((rc = launch_layernorm(x_mid, Lw.ln_2_w, Lw.ln_2_b, M, d, cfg->ln_eps, ln2, s))) return rc;
        if (g_train_fuse_gelu && Lw.c_fc_w3 && g_gemm_split3 && gemm_s3_supported(M, d, 4 * d)) {
            // one launch: f = gelu_new(v) and the pre-activation v (kept for the backward pass) both leave the GEMM's epilogue
            if ((rc = fwd_linear(ln2, Lw.c_fc_w, Lw.c_fc_wT, Lw.c_fc_b, pre, M, d, 4 * d, EPI_GELU_KEEP, f, s, Lw.c_fc_w3, Lw.c_fc_h2))) return rc;
        } else {
            if ((rc = fwd_linear(ln2, Lw.c_fc_w, Lw.c_fc_wT, Lw.c_fc_b, nullptr, M, d, 4 * d, EPI_NONE, pre, s, Lw.c_fc_w3, Lw.c_fc_h2))) return rc;
            if ((rc = launch_gelu_fwd(pre, (long long)M * 4 * d, f, s))) return rc;
        }
        float* x_next = l + 1 < cfg->n_layer ? ws + t.x_in[l + 1] : ws + t.x_out;
        if (dc.resid_p > 0.f) {                                      // x + dropout(c_proj(act(c_fc(x)))), :212,233
            if ((rc = fwd_linear(f, Lw.mlp_proj_w, Lw.mlp_proj_wT, Lw.mlp_proj_b, nullptr, M, 4 * d, d, EPI_NONE, branch, s, Lw.mlp_proj_w3, Lw.mlp_proj_h2))) return rc;
            if ((rc = launch_dropout(branch, x_mid, (long long)M * d, x_next, dc.resid_p, dc.key, 4u * l + 2u, 0, s))) return rc;
        } else if ((rc = fwd_linear(f, Lw.mlp_proj_w, Lw.mlp_proj_wT, Lw.mlp_proj_b, x_mid, M, 4 * d, d, EPI_RESIDUAL, x_next, s, Lw.mlp_proj_w3, Lw.mlp_proj_h2))) return rc;
    }
    return launch_lnf_meanpool_groups(R, ws + t.x_out, w->ln_f_w, w->ln_f_b, d, cfg->ln_eps, nullptr, out_meanpool_d,
                                      ws + t.pool_scratch, s);
}

int r4d_gpt2_train_backward_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_gpt2_grads* gr,
                                int32_t n_groups, const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts,
                                const float* d_meanpool_d, const r4d_train_dropout* dropout, void* workspace_d,
                                size_t workspace_bytes, void* stream) {
    if (g_train_fuse_gelu < 0) { const char* e = getenv("R4D_TRAIN_FUSE_GELU"); g_train_fuse_gelu = e ? atoi(e) : 1; }
    hipStream_t s = (hipStream_t)stream;
    std::vector<TrainGroup> gs;
    int rc = check_groups(cfg, n_groups, ids_d, Bs, Ts, gs);
    if (rc) return rc;
    DropCtx dc;
    if ((rc = drop_ctx(dropout, dc))) return rc;
    R4D_REQUIRE(w && w->layers && gr && gr->layers && gr->wte && gr->wpe && gr->ln_f_w && gr->ln_f_b && d_meanpool_d,
                "gpt2 train backward: null pointer");
    const TrainLayout t = layout(cfg, gs.data(), n_groups);
    if (!workspace_d || workspace_bytes < t.total * sizeof(float)) {
        set_error("gpt2 train backward: workspace %zu bytes < required %zu", workspace_bytes, t.total * sizeof(float));
        return R4D_ERR_WORKSPACE;
    }
    float* ws = (float*)workspace_d;
    const int d = cfg->n_embd, H = cfg->n_head, M = (int)t.M, L = cfg->n_layer;
    float *dx = ws + t.dx, *dy = ws + t.dy, *dbig = ws + t.dbig, *dqkv = ws + t.dqkv, *xT = ws + t.xT, *red = ws + t.red;
    // mean over T -> ln_f
    for (const TrainGroup& G : gs)
        if ((rc = launch_meanpool_bwd(d_meanpool_d + G.seq0 * d, (long long)G.B * G.T, G.T, d, dy + G.row0 * d, s))) return rc;
    if ((rc = launch_ln_bwd(ws + t.x_out, w->ln_f_w, dy, nullptr, M, d, cfg->ln_eps, dx, gr->ln_f_w, gr->ln_f_b, red, 0, s))) return rc;
    for (int l = L - 1; l >= 0; --l) {
        const r4d_gpt2_layer& Lw = w->layers[l];
        const r4d_gpt2_layer_grads& Lg = gr->layers[l];
        R4D_REQUIRE(Lg.ln_1_w && Lg.ln_1_b && Lg.c_attn_w && Lg.c_attn_b && Lg.attn_proj_w && Lg.attn_proj_b && Lg.ln_2_w &&
                    Lg.ln_2_b && Lg.c_fc_w && Lg.c_fc_b && Lg.mlp_proj_w && Lg.mlp_proj_b, "gpt2 train backward: null gradient in layer %d", l);
        float *x_in = ws + t.x_in[l], *ln1 = ws + t.ln1[l], *qkv = ws + t.qkv[l], *att = ws + t.att[l];
        float *x_mid = ws + t.x_mid[l], *ln2 = ws + t.ln2[l], *pre = ws + t.pre[l], *f = ws + t.f[l];
        // ---- MLP: x_out = x_mid + gelu(ln_2(x_mid) Wfc + bfc) Wp + bp ;  dx holds d(x_out)
        const float* dbr = dx;                                       // gradient of the branch output: through its dropout mask
        if (dc.resid_p > 0.f) {
            if ((rc = launch_dropout(dx, nullptr, (long long)M * d, dy, dc.resid_p, dc.key, 4u * l + 2u, 0, s))) return rc;
            dbr = dy;
        }
        if ((rc = bwd_weight(f, dbr, M, 4 * d, d, Lg.mlp_proj_w, Lg.mlp_proj_b, xT, red, s))) return rc;
        if (g_train_fuse_gelu && Lw.mlp_proj_w3t && g_gemm_split3 && gemm_s3_supported(M, d, 4 * d)) {
            // d(pre) = (d(branch) . Wp^T) * gelu_new'(pre): the derivative is applied in the GEMM's epilogue
            if ((rc = bwd_data(dbr, Lw.mlp_proj_w, M, 4 * d, d, dbig, s, Lw.mlp_proj_w3t, pre))) return rc;
        } else {
            if ((rc = bwd_data(dbr, Lw.mlp_proj_w, M, 4 * d, d, dbig, s, Lw.mlp_proj_w3t))) return rc;                    // d(f)
            if ((rc = launch_gelu_bwd(pre, dbig, (long long)M * 4 * d, dbig, s))) return rc;              // d(pre), in place
        }
        if ((rc = bwd_weight(ln2, dbig, M, d, 4 * d, Lg.c_fc_w, Lg.c_fc_b, xT, red, s))) return rc;
        if ((rc = bwd_data(dbig, Lw.c_fc_w, M, d, 4 * d, dy, s, Lw.c_fc_w3t))) return rc;                              // d(ln_2 out)
        if ((rc = launch_ln_bwd(x_mid, Lw.ln_2_w, dy, dx, M, d, cfg->ln_eps, dx, Lg.ln_2_w, Lg.ln_2_b, red, 0, s))) return rc;   // dx = d(x_mid)
        // ---- attention: x_mid = x_in + attn(ln_1(x_in)) Wo + bo
        dbr = dx;
        if (dc.resid_p > 0.f) {                                      // dbig (M x 4d) is free here
            if ((rc = launch_dropout(dx, nullptr, (long long)M * d, dbig, dc.resid_p, dc.key, 4u * l + 1u, 0, s))) return rc;
            dbr = dbig;
        }
        if ((rc = bwd_weight(att, dbr, M, d, d, Lg.attn_proj_w, Lg.attn_proj_b, xT, red, s))) return rc;
        if ((rc = bwd_data(dbr, Lw.attn_proj_w, M, d, d, dy, s, Lw.attn_proj_w3t))) return rc;                              // d(att), merged heads
        for (const TrainGroup& G : gs)
            if ((rc = attn_bwd(qkv + G.row0 * 3 * d, ws + t.P[l] + G.p0, dy + G.row0 * d, G.B, G.T, H, d, dqkv + G.row0 * 3 * d,
                               ws + t.dP, ws + t.PT, s, dc.attn_p, dc.key, 4u * l + 0u, G.p0))) return rc;
        if ((rc = bwd_weight(ln1, dqkv, M, d, 3 * d, Lg.c_attn_w, Lg.c_attn_b, xT, red, s))) return rc;
        if ((rc = bwd_data(dqkv, Lw.c_attn_w, M, d, 3 * d, dy, s, Lw.c_attn_w3t))) return rc;                            // d(ln_1 out)
        if ((rc = launch_ln_bwd(x_in, Lw.ln_1_w, dy, dx, M, d, cfg->ln_eps, dx, Lg.ln_1_w, Lg.ln_1_b, red, 0, s))) return rc;    // dx = d(x_in)
    }
    // embeddings: x_in[0] = drop(wte[ids] + wpe[0..T-1])
    if (dc.embd_p > 0.f && (rc = launch_dropout(dx, nullptr, (long long)M * d, dx, dc.embd_p, dc.key, R4D_DROPOUT_SITE_EMBD, 0, s)))
        return rc;
    // deterministic sums (train_ops.hip): tokens through a 64-bit fixed-point table, positions as ordered column sums
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws + t.emb_acc);
    R4D_HIP(hipMemsetAsync(acc, 0, ((size_t)cfg->vocab * d + 2) * sizeof(unsigned long long), s));       // table + poison word + max word
    R4D_HIP(hipMemsetAsync(gr->wpe, 0, (size_t)cfg->n_positions * d * sizeof(float), s));
    if ((rc = launch_embedding_absmax(dx, (long long)M * d, acc, (long long)cfg->vocab * d, s))) return rc;      // the scale follows the data
    for (const TrainGroup& G : gs)
        if ((rc = launch_embedding_bwd(dx + G.row0 * d, G.ids, G.B, G.T, d, cfg->vocab, acc, gr->wpe, 0, (long long)M, s))) return rc;
    return launch_embedding_fix_to_f32(acc, (long long)cfg->vocab * d, (long long)M, gr->wte, s);
}

}  // extern "C"
