// C-ABI entry points of the GPT-2 encoder: orchestration of the gfx950 kernels, no device allocation,
// everything enqueued on the caller's stream (graph-capturable).  See include/r4d.h for the contract.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace r4d {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_gemm_split3 = 1;                // r4d_set_gemm_split3: 0 exact f32, 1 bf16x3, 2 f16x2

int conv1d(const float* x, const float* w, const float* wT, const float* bias, const float* resid, int M, int K,
           int N, int epilogue, float* y, hipStream_t s, float* skinny_scratch, bool sk_counters_zeroed,
           const unsigned short* w3, const unsigned short* w2h) {
    if (skinny_scratch && wT && gemm_skinny_supported(M, K, N))        // decode step: a weight stream, not a tiled GEMM
        return launch_gemm_skinny(x, wT, bias, resid, M, K, N, epilogue, y, skinny_scratch, s, nullptr, nullptr, 0.f,
                                  sk_counters_zeroed);
    // f16x2 planes present and selected: fp16 matrix cores, three products per fp32 product, fp32 accuracy (gemm_h2.hip).
    // NOT gated on M, like the bf16x3 branch below
    if (w2h && g_gemm_split3 == 2 && (epilogue <= EPI_RESIDUAL || epilogue == EPI_GELU_KEEP || epilogue == EPI_H2WORDS) && gemm_h2_supported(M, K, N)) {
        S3Args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.planes = w2h; a.C = y; a.bias = bias; a.resid = resid;
        a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
        return launch_gemm_h2(a, s);
    }
    // bf16x3 planes present: bf16 matrix cores at fp32 accuracy.  NOT gated on M: a row's result must not depend on how many
    // other rows share the call (fused multi-batch encode == one call per batch, bit for bit)
    if (w3 && g_gemm_split3 && gemm_s3_supported(M, K, N)) {
        S3Args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.planes = w3; a.C = y; a.bias = bias; a.resid = resid;
        a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
        return launch_gemm_s3(a, s);
    }
    R4D_REQUIRE(epilogue != EPI_H2WORDS, "conv1d: the h2-word epilogue exists in the f16x2 GEMM only");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.C = y; g.bias = bias; g.resid = resid;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.ldr = N;
    if (wT) { g.B = wT; g.ldb = K; g.b_trans = 1; g.b_rows = N; }     // k-contiguous copy of the weight: fast kernel
    else { g.B = w; g.ldb = N; g.b_trans = 0; g.b_rows = K; }         // reference layout [in,out]
    g.nbatch = 1; g.nb1 = 1;
    g.epilogue = epilogue; g.scale_div = 1.f; g.causal = CAUSAL_NONE;
    return launch_gemm_f32(g, s);
}

// Conv1D whose input rows are f16x2 LINES written by their producer (gemm_h2p.hip): f16x2 mode only; out_lines: the GELU epilogue
// writes the result as lines too (c_fc -> mlp.c_proj)
// kblk / kb_hd (c_attn with the h2-word epilogue only): the K third goes to the key-blocked image of attention_h2.hip
static int conv1d_lines(const unsigned short* x_lines, const unsigned short* w2h, const float* bias, const float* resid, int M, int K, int N,
                        int epilogue, void* y, bool out_lines, hipStream_t s, unsigned* kblk = nullptr, int kb_hd = 0) {
    S3Args a;
    memset(&a, 0, sizeof(a));
    a.planes = w2h; a.C = (float*)y; a.bias = bias; a.resid = resid;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
    a.kblk = kblk; a.kb_hd = kb_hd;
    return launch_gemm_h2p(a, x_lines, out_lines, s);
}

static inline int tpad(int T) { return (T + 127) / 128 * 128; }

// Attention._attn + split_heads/merge_heads (modeling_gpt2.py:140-175) as three launches over the packed
// c_attn output: batched per-head Q.K^T (tiles above the diagonal skipped, logits DIVIDED by sqrt(hd) as
// in :143), in-place causal softmax, batched P.V (K-loop trimmed to the causal range) writing straight
// into the merged-head layout.  The B*H*T*T score block (29 MB at the worst UCI_13 batch) stays in
// the 256 MB Infinity Cache between the launches.
int g_attention_fused = -1;           // -1 auto (by head_dim), 0 three launches, 1 fused wherever instantiated
int g_attention_h2 = 1;               // r4d_set_attention_h2: in f16x2 mode, head_dim 128 / 256 on the fp16 matrix cores (attention_h2.hip)
static int g_attention_kblk = -1;     // tuning aid, read once: R4D_ATT_KBLK=0 keeps the row-major K words (the key-blocked image needs the LDS-DMA c_attn)

static int attention(const float* qkv, int B, int T, int H, int d, float* a_out, float* scores, hipStream_t s) {
    // Measured on MI355X (tools/attn_bench.py, B=128, T=128..300): the fused kernels are 1.15-2.7x faster than the
    // three-launch form at every instantiated head_dim (key-split kernel for hd 32 / 64, column-split for 96 / 128 / 256),
    // so auto == fused; head dims without an instantiation fall through to the GEMM form.
    const bool fused = g_attention_fused != 0;
    if (fused) {
        const int rc_f = launch_attention_fused(qkv, B, T, H, d, a_out, s);
        if (rc_f <= 0) return rc_f;                      // +1: no fused instantiation for this head_dim
    }
    R4D_BRANCH(ATT_3LAUNCH);
    const int hd = d / H, ld = tpad(T);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = qkv; g.B = qkv + d; g.C = scores;
    g.M = T; g.N = T; g.K = hd; g.lda = 3 * d; g.ldb = 3 * d; g.ldc = ld;
    g.b_trans = 1; g.b_rows = T; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = (long long)T * 3 * d; g.sA1 = hd; g.sB0 = g.sA0; g.sB1 = hd;
    g.sC0 = (long long)H * T * ld; g.sC1 = (long long)T * ld;
    g.epilogue = EPI_SCALE_DIV; g.scale_div = (float)sqrt((double)hd); g.causal = CAUSAL_QK;
    int rc = launch_gemm_f32(g, s);
    if (rc) return rc;
    rc = launch_causal_softmax(scores, B * H, T, ld, 128, s);
    if (rc) return rc;
    memset(&g, 0, sizeof(g));
    g.A = scores; g.B = qkv + 2 * d; g.C = a_out;
    g.M = T; g.N = hd; g.K = T; g.lda = ld; g.ldb = 3 * d; g.ldc = d;
    g.b_trans = 0; g.b_rows = T; g.nbatch = B * H; g.nb1 = H;
    g.sA0 = (long long)H * T * ld; g.sA1 = (long long)T * ld;
    g.sB0 = (long long)T * 3 * d; g.sB1 = hd; g.sC0 = (long long)T * d; g.sC1 = hd;
    g.a_cols = ld;
    g.epilogue = EPI_NONE; g.scale_div = 1.f; g.causal = CAUSAL_PV;
    return launch_gemm_f32(g, s);
}

struct Group {                 // one right-padded reference batch inside a fused launch sequence
    const int64_t* ids; const float* emb; int B, T; size_t row0, seq0;
};

static RowGroups row_groups(const Group* groups, int g0, int n_groups) {
    RowGroups R;
    R.n = n_groups - g0 < ATT_MAXG ? n_groups - g0 : ATT_MAXG;
    for (int j = 0; j < ATT_MAXG; ++j) {
        const bool in = j < R.n;
        R.B[j] = in ? groups[g0 + j].B : 0; R.T[j] = in ? groups[g0 + j].T : 0;
        R.ids[j] = in ? groups[g0 + j].ids : nullptr; R.emb[j] = in ? groups[g0 + j].emb : nullptr;
    }
    return R;
}

struct Workspace {
    float *x, *ln, *qkv, *att, *fc, *scores, *pool;
    unsigned* kblk;       // key-blocked K words of the f16x2 attention (attention_h2.hip): ceil32(M) x d
    size_t bytes;
};
static Workspace carve(void* base, size_t M, size_t score_floats, size_t pool_floats, int d) {
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? (float*)((char*)base + off) : nullptr;
        off += align_up(nfloat * sizeof(float), 256);
        return p;
    };
    Workspace w;
    w.x = take(M * d); w.ln = take(M * d); w.qkv = take(M * 3 * d); w.att = take(M * d); w.fc = take(M * 4 * d);
    w.scores = take(score_floats);
    w.pool = take(pool_floats);
    w.kblk = reinterpret_cast<unsigned*>(take(kblk_words(M, d)));
    w.bytes = off;
    return w;
}
static size_t score_floats(int B, int H, int T) { return (size_t)B * H * T * tpad(T); }

static int check_cfg(const r4d_gpt2_config* c) {
    R4D_REQUIRE(c != nullptr, "gpt2: null config");
    R4D_REQUIRE(c->n_layer >= 1 && c->n_head >= 1 && c->n_embd >= 64, "gpt2: bad config L=%d H=%d d=%d", c->n_layer,
                c->n_head, c->n_embd);
    R4D_REQUIRE(c->n_embd % 64 == 0 && c->n_embd <= 2048, "gpt2: n_embd=%d must be a multiple of 64, <= 2048", c->n_embd);
    R4D_REQUIRE(c->n_embd % c->n_head == 0 && (c->n_embd / c->n_head) % 16 == 0,
                "gpt2: head_dim=%d/%d must be a multiple of 16", c->n_embd, c->n_head);
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_abi_version(void) { return R4D_ABI_VERSION; }
int r4d_build_flags(void) {
    return dbgflag_kc() | (dbgflag_att() << 1) | (dbgflag_sk() << 2) | (dbgflag_jac() << 3) | (dbgflag_scan() << 4) | (dbgflag_s3() << 5) | (dbgflag_h2() << 6) | (dbgflag_att_h2() << 7);
}
int r4d_fold_layernorm_f32(const float* wT_d, const float* ln_w_d, const float* ln_b_d, int32_t N, int32_t K, float* wTg_d,
                           float* lnc_d, void* stream) {
    return launch_fold_layernorm(wT_d, ln_w_d, ln_b_d, N, K, wTg_d, lnc_d, (hipStream_t)stream);
}

int r4d_set_gemm_split3(int32_t mode) { g_gemm_split3 = mode == 2 ? 2 : (mode != 0); return R4D_OK; }
int r4d_get_gemm_split3(void) { return g_gemm_split3; }
int r4d_set_range_flag(uint32_t* flag_d) { g_range_flag = flag_d; return R4D_OK; }
int r4d_set_attention_kblk(int32_t on) {
    const int prev = g_attention_kblk < 0 ? 1 : g_attention_kblk;
    g_attention_kblk = on != 0;
    return prev;
}
int r4d_set_attention_h2(int32_t on) {
    const int prev = g_attention_h2;
    g_attention_h2 = on != 0;
    return prev;
}

int r4d_set_attention_fused(int32_t mode) {
    // mode 2: fused with the key-split kernel forced at head_dim 96/128/256 (A/B tuning); 1: fused (column-split there)
    g_attention_variant = (mode == 2) ? 1 : 0;
    g_attention_fused = mode < 0 ? -1 : (mode != 0);
    return R4D_OK;
}
const char* r4d_last_error(void) { return g_err; }

size_t r4d_gpt2_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B, int32_t T) {
    if (!cfg || B <= 0 || T <= 0) return 0;
    return carve(nullptr, (size_t)B * T, score_floats(B, cfg->n_head, T), lnf_meanpool_scratch_floats(B, T, cfg->n_embd),
                 cfg->n_embd).bytes;
}

size_t r4d_gpt2_groups_workspace_bytes(const r4d_gpt2_config* cfg, int32_t n_groups, const int32_t* Bs,
                                       const int32_t* Ts) {
    if (!cfg || n_groups <= 0 || !Bs || !Ts) return 0;
    size_t M = 0, sc = 0, pl = 0;
    for (int g = 0; g < n_groups; ++g) {
        if (Bs[g] <= 0 || Ts[g] <= 0) return 0;
        M += (size_t)Bs[g] * Ts[g];
        const size_t f = score_floats(Bs[g], cfg->n_head, Ts[g]);
        if (f > sc) sc = f;
        pl += lnf_meanpool_scratch_floats(Bs[g], Ts[g], cfg->n_embd);      // all batches pooled by one launch
    }
    return carve(nullptr, M, sc, pl, cfg->n_embd).bytes;
}

}  // extern "C"

namespace r4d {

// The whole encoder over G right-padded batches.  Row-wise work (LayerNorm, the four Conv1D GEMMs per block)
// runs ONCE over the concatenated rows of all batches -- a mean-pooled embedding depends on its own batch's
// padding only through the attention / position / pooling steps, which stay per batch -- so a launch sees
// G times more tiles and the GEMM tail (a CU holds ~3 tiles of a single 32 x T batch) amortises.
static int encode_impl(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const Group* groups, int n_groups,
                       float* out_hidden_d, float* out_meanpool_d, float* out_layers_d, float* out_qkv_d,
                       void* workspace_d, size_t workspace_bytes, hipStream_t s) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    R4D_REQUIRE(w && w->wte && w->wpe && w->ln_f_w && w->ln_f_b && w->layers, "gpt2: null weights");
    R4D_REQUIRE(out_hidden_d || out_meanpool_d, "gpt2: no output requested");
    const int d = cfg->n_embd, H = cfg->n_head;
    size_t Mtot = 0, sc = 0, pl = 0;
    for (int g = 0; g < n_groups; ++g) {
        const Group& G = groups[g];
        R4D_REQUIRE((G.ids != nullptr) != (G.emb != nullptr),
                    "gpt2: specify exactly one of input_ids and inputs_embeds");     // modeling_gpt2.py:400-401,409
        R4D_REQUIRE(G.B >= 1 && G.T >= 1, "gpt2: empty batch B=%d T=%d", G.B, G.T);
        R4D_REQUIRE(G.T <= cfg->n_positions && G.T <= 1024, "gpt2: T=%d exceeds n_positions=%d", G.T, cfg->n_positions);
        Mtot += (size_t)G.B * G.T;
        const size_t f = score_floats(G.B, H, G.T);
        if (f > sc) sc = f;
        pl += lnf_meanpool_scratch_floats(G.B, G.T, d);
    }
    R4D_REQUIRE(Mtot <= 0x7fffffff / (size_t)(16 * d), "gpt2: %zu rows in one call is too many", Mtot);   // [M,4d] f32 byte offsets fit 31 bits
    const int M = (int)Mtot;
    Workspace ws = carve(workspace_d, Mtot, sc, pl, d);
    if (!workspace_d || workspace_bytes < ws.bytes) {
        set_error("gpt2: workspace %zu bytes < required %zu", workspace_bytes, ws.bytes);
        return R4D_ERR_WORKSPACE;
    }
    for (int l = 0; l < cfg->n_layer; ++l) {
        const r4d_gpt2_layer& L = w->layers[l];
        R4D_REQUIRE(L.ln_1_w && L.c_attn_w && L.attn_proj_w && L.ln_2_w && L.c_fc_w && L.mlp_proj_w,
                    "gpt2: null weight in layer %d", l);
        // f16x2 mode, round 5: the LayerNorms write their rows as f16x2 LINES, the c_fc GELU epilogue too, and c_attn / c_fc /
        // mlp.c_proj take them through LDS-DMA (gemm_h2p.hip) -- the values of the register-staged gemm_h2 path, bit for bit
        // (r4d_set_gemm_h2p(0) keeps that one); the buffers keep their size (4 bytes per element either way)
        const bool lines = g_gemm_split3 == 2 && g_gemm_h2p && L.c_attn_h2 && L.c_fc_h2 && L.mlp_proj_h2 && layernorm_lines_supported(d) &&
                           gemm_h2p_supported(M, d, 3 * d) && gemm_h2p_supported(M, d, 4 * d) && gemm_h2p_supported(M, 4 * d, d);
        unsigned short* ln_lines = reinterpret_cast<unsigned short*>(ws.ln);
        unsigned short* fc_lines = reinterpret_cast<unsigned short*>(ws.fc);
        if (l == 0) {
            for (int g0 = 0; g0 < n_groups; g0 += ATT_MAXG) {          // ATT_MAXG batches per launch
                const RowGroups R = row_groups(groups, g0, n_groups);
                const size_t r0 = groups[g0].row0;
                rc = launch_embed_layernorm_groups(R, w->wte, w->wpe, cfg->vocab, d, L.ln_1_w, L.ln_1_b, cfg->ln_eps,
                                                   ws.x + r0 * d, ws.ln + r0 * d, s, lines);
                if (rc) return rc;
            }
        } else if ((rc = lines ? launch_layernorm_lines(ws.x, L.ln_1_w, L.ln_1_b, M, d, cfg->ln_eps, ln_lines, s)
                               : launch_layernorm(ws.x, L.ln_1_w, L.ln_1_b, M, d, cfg->ln_eps, ws.ln, s))) {
            return rc;
        }
        if (out_layers_d &&
            hipMemcpyAsync(out_layers_d + (size_t)l * M * d, ws.x, (size_t)M * d * sizeof(float),
                           hipMemcpyDeviceToDevice, s) != hipSuccess) {
            set_error("gpt2: layer copy failed");
            return R4D_ERR_HIP;
        }
        float* qkv = out_qkv_d ? out_qkv_d + (size_t)l * M * 3 * d : ws.qkv;
        // f16x2 mode, head_dim 128 / 256, qkv not handed out: c_attn writes h2 words (csrc/h2.h) and the attention runs on
        // the fp16 matrix cores (attention_h2.hip); every other case: fp32 qkv and the exact-f32 kernels below
        bool words = !out_qkv_d && g_attention_h2 && g_attention_fused != 0 && g_gemm_split3 == 2 && L.c_attn_h2 &&
                     attention_h2_supported(H, d) && gemm_h2_supported(M, d, 3 * d);
        for (int g0 = 0; g0 < n_groups && words; g0 += ATT_MAXG) {
            int nseq = 0;
            for (int j = g0; j < n_groups && j < g0 + ATT_MAXG; ++j) nseq += groups[j].B;
            if (nseq > 65535) words = false;
        }
        // ... and its K third as the key-blocked image the head_dim 128 / 256 kernel loads by whole cache lines (attention_h2.hip)
        if (g_attention_kblk < 0) { const char* e = getenv("R4D_ATT_KBLK"); g_attention_kblk = e ? atoi(e) != 0 : 1; }
        const bool kblk = words && lines && g_attention_kblk && (d / H) % 32 == 0;      // every head_dim attention_h2.hip serves
        if (kblk && (M & 31) &&        // the last block's rows past M are never written: keep them finite (they are masked keys at most)
            hipMemsetAsync(ws.kblk + (size_t)(M / 32) * 32 * d, 0, (size_t)32 * d * sizeof(unsigned), s) != hipSuccess) {
            set_error("gpt2: memset of the key-block tail failed");
            return R4D_ERR_HIP;
        }
        if (lines) rc = conv1d_lines(ln_lines, L.c_attn_h2, L.c_attn_b, nullptr, M, d, 3 * d, words ? EPI_H2WORDS : EPI_NONE, qkv, false, s,
                                     kblk ? ws.kblk : nullptr, kblk ? d / H : 0);
        else rc = conv1d(ws.ln, L.c_attn_w, L.c_attn_wT, L.c_attn_b, nullptr, M, d, 3 * d, words ? EPI_H2WORDS : EPI_NONE, qkv, s, nullptr, false, L.c_attn_w3, L.c_attn_h2);
        if (rc) return rc;
        // ... and with them the attention output: attn_h2_kernel writes its merged-head rows as lines for attn.c_proj
        const bool att_lines = words && lines && L.attn_proj_h2 && gemm_h2p_supported(M, d, d);
        bool fused_done = false;
        if (words) {
            fused_done = true;
            for (int g0 = 0; g0 < n_groups; g0 += ATT_MAXG) {
                int Bs[ATT_MAXG], Ts[ATT_MAXG];
                long long r0[ATT_MAXG];
                const int n = n_groups - g0 < ATT_MAXG ? n_groups - g0 : ATT_MAXG;
                for (int j = 0; j < n; ++j) { Bs[j] = groups[g0 + j].B; Ts[j] = groups[g0 + j].T; r0[j] = (long long)groups[g0 + j].row0; }
                if ((rc = launch_attention_h2_groups(reinterpret_cast<const unsigned*>(qkv), n, Bs, Ts, r0, H, d, ws.att, s, att_lines,
                                                     kblk ? ws.kblk : nullptr))) return rc;
            }
        } else if (g_attention_fused != 0) {             // all batches of the call in ceil(n/16) fused launches
            fused_done = true;
            for (int g0 = 0; g0 < n_groups && fused_done; g0 += ATT_MAXG) {
                int Bs[ATT_MAXG], Ts[ATT_MAXG];
                long long r0[ATT_MAXG];
                const int n = n_groups - g0 < ATT_MAXG ? n_groups - g0 : ATT_MAXG;
                int nseq = 0;
                for (int j = 0; j < n; ++j) {
                    Bs[j] = groups[g0 + j].B; Ts[j] = groups[g0 + j].T; r0[j] = (long long)groups[g0 + j].row0;
                    nseq += Bs[j];
                }
                if (nseq > 65535) { fused_done = false; break; }
                rc = launch_attention_fused_groups(qkv, n, Bs, Ts, r0, H, d, ws.att, s);
                if (rc < 0) return rc;
                if (rc > 0) fused_done = false;          // no fused instantiation for this head_dim
            }
        }
        if (!fused_done) {
            for (int g = 0; g < n_groups; ++g) {
                const Group& G = groups[g];
                if ((rc = attention(qkv + G.row0 * 3 * d, G.B, G.T, H, d, ws.att + G.row0 * d, ws.scores, s))) return rc;
            }
        }
        if (att_lines) rc = conv1d_lines(reinterpret_cast<const unsigned short*>(ws.att), L.attn_proj_h2, L.attn_proj_b, ws.x, M, d, d, EPI_RESIDUAL, ws.x, false, s);
        else rc = conv1d(ws.att, L.attn_proj_w, L.attn_proj_wT, L.attn_proj_b, ws.x, M, d, d, EPI_RESIDUAL, ws.x, s, nullptr, false, L.attn_proj_w3, L.attn_proj_h2);
        if (rc) return rc;
        if (lines) {
            if ((rc = launch_layernorm_lines(ws.x, L.ln_2_w, L.ln_2_b, M, d, cfg->ln_eps, ln_lines, s))) return rc;
            if ((rc = conv1d_lines(ln_lines, L.c_fc_h2, L.c_fc_b, nullptr, M, d, 4 * d, EPI_GELU, fc_lines, true, s))) return rc;
            if ((rc = conv1d_lines(fc_lines, L.mlp_proj_h2, L.mlp_proj_b, ws.x, M, 4 * d, d, EPI_RESIDUAL, ws.x, false, s))) return rc;
            continue;
        }
        if ((rc = launch_layernorm(ws.x, L.ln_2_w, L.ln_2_b, M, d, cfg->ln_eps, ws.ln, s))) return rc;
        if ((rc = conv1d(ws.ln, L.c_fc_w, L.c_fc_wT, L.c_fc_b, nullptr, M, d, 4 * d, EPI_GELU, ws.fc, s, nullptr, false, L.c_fc_w3, L.c_fc_h2))) return rc;
        if ((rc = conv1d(ws.fc, L.mlp_proj_w, L.mlp_proj_wT, L.mlp_proj_b, ws.x, M, 4 * d, d, EPI_RESIDUAL, ws.x, s, nullptr, false, L.mlp_proj_w3, L.mlp_proj_h2))) return rc;
    }
    size_t part0 = 0;                                                   // scratch offset of the launch's first batch
    for (int g0 = 0; g0 < n_groups; g0 += ATT_MAXG) {
        const RowGroups R = row_groups(groups, g0, n_groups);
        const Group& G = groups[g0];
        rc = launch_lnf_meanpool_groups(R, ws.x + G.row0 * d, w->ln_f_w, w->ln_f_b, d, cfg->ln_eps,
                                        out_hidden_d ? out_hidden_d + G.row0 * d : nullptr,
                                        out_meanpool_d ? out_meanpool_d + G.seq0 * d : nullptr, ws.pool + part0, s);
        if (rc) return rc;
        for (int j = 0; j < R.n; ++j) part0 += lnf_meanpool_scratch_floats(R.B[j], R.T[j], d);
    }
    return R4D_OK;
}

}  // namespace r4d

extern "C" {

int r4d_gpt2_encode_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const int64_t* ids_d,
                        const float* inputs_embeds_d, int32_t B, int32_t T, float* out_hidden_d,
                        float* out_meanpool_d, float* out_layers_d, float* out_qkv_d, void* workspace_d,
                        size_t workspace_bytes, void* stream) {
    Group G = {ids_d, inputs_embeds_d, B, T, 0, 0};
    return encode_impl(cfg, w, &G, 1, out_hidden_d, out_meanpool_d, out_layers_d, out_qkv_d, workspace_d,
                       workspace_bytes, (hipStream_t)stream);
}

int r4d_gpt2_encode_groups_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                               const int64_t* const* ids_d, const int32_t* Bs, const int32_t* Ts,
                               float* out_meanpool_d, void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(n_groups >= 1 && n_groups <= 4096 && ids_d && Bs && Ts, "gpt2 groups: bad arguments");
    R4D_REQUIRE(out_meanpool_d, "gpt2 groups: out_meanpool_d is null");
    std::vector<Group> gs((size_t)n_groups);
    size_t row0 = 0, seq0 = 0;
    for (int g = 0; g < n_groups; ++g) {
        gs[g] = Group{ids_d[g], nullptr, Bs[g], Ts[g], row0, seq0};
        if (Bs[g] > 0 && Ts[g] > 0) { row0 += (size_t)Bs[g] * Ts[g]; seq0 += (size_t)Bs[g]; }
    }
    return encode_impl(cfg, w, gs.data(), n_groups, nullptr, out_meanpool_d, nullptr, nullptr, workspace_d,
                       workspace_bytes, (hipStream_t)stream);
}

int r4d_gpt2_encode_groups_ex_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, int32_t n_groups,
                                  const int64_t* const* ids_d, const float* const* embeds_d, const int32_t* Bs,
                                  const int32_t* Ts, float* out_hidden_d, float* out_meanpool_d, float* out_qkv_d,
                                  void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(n_groups >= 1 && n_groups <= 4096 && Bs && Ts, "gpt2 groups: bad arguments");
    R4D_REQUIRE((ids_d != nullptr) != (embeds_d != nullptr), "gpt2 groups: specify exactly one of ids_d and embeds_d");
    std::vector<Group> gs((size_t)n_groups);
    size_t row0 = 0, seq0 = 0;
    for (int g = 0; g < n_groups; ++g) {
        gs[g] = Group{ids_d ? ids_d[g] : nullptr, embeds_d ? embeds_d[g] : nullptr, Bs[g], Ts[g], row0, seq0};
        if (Bs[g] > 0 && Ts[g] > 0) { row0 += (size_t)Bs[g] * Ts[g]; seq0 += (size_t)Bs[g]; }
    }
    return encode_impl(cfg, w, gs.data(), n_groups, out_hidden_d, out_meanpool_d, nullptr, out_qkv_d, workspace_d,
                       workspace_bytes, (hipStream_t)stream);
}

// decode workspace: the per-row buffers of the encoder + the split-K scratch of the skinny GEMM (largest projection:
// K = 4d, N = d and K = d, N = 4d both give (4d / 256) * 32 * 4d... the max of the four shapes is taken)
static size_t decode_skinny_floats(int d) {
    size_t m = 0;
    const int shapes[4][2] = {{d, 3 * d}, {d, d}, {d, 4 * d}, {4 * d, d}};
    for (auto& kn : shapes)
        if (gemm_skinny_supported(1, kn[0], kn[1])) { const size_t f = gemm_skinny_scratch_floats(kn[0], kn[1]); if (f > m) m = f; }
    return m;
}

size_t r4d_gpt2_decode_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B) {
    if (!cfg || B <= 0) return 0;
    return carve(nullptr, (size_t)B, 0, decode_skinny_floats(cfg->n_embd), cfg->n_embd).bytes;
}

// `x_ready`: the un-normalised input rows are already in the workspace's x buffer and the ticket counters are clear (the
// greedy step's bookkeeping kernel did both): no embedding launch, layer 0 reads its LayerNorm like every other layer
static int decode_step_impl(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const int64_t* ids_d,
                            const float* inputs_embeds_d, const int32_t* pos_d, float* kv_cache_d, int32_t B,
                            int32_t t_cap, float* out_hidden_d, void* workspace_d, size_t workspace_bytes,
                            void* stream, bool x_ready) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    R4D_REQUIRE(w && w->wte && w->wpe && w->ln_f_w && w->ln_f_b && w->layers, "gpt2 decode: null weights");
    R4D_REQUIRE(x_ready || (ids_d != nullptr) != (inputs_embeds_d != nullptr), "gpt2 decode: specify exactly one of ids and inputs_embeds");
    R4D_REQUIRE(pos_d && kv_cache_d && out_hidden_d, "gpt2 decode: null pointer");
    R4D_REQUIRE(B >= 1 && t_cap >= 1, "gpt2 decode: B=%d t_cap=%d", B, t_cap);
    const int d = cfg->n_embd, H = cfg->n_head;
    const size_t sk_floats = decode_skinny_floats(d);
    Workspace ws = carve(workspace_d, (size_t)B, 0, sk_floats, d);
    if (!workspace_d || workspace_bytes < ws.bytes) {
        set_error("gpt2 decode: workspace %zu bytes < required %zu", workspace_bytes, ws.bytes);
        return R4D_ERR_WORKSPACE;
    }
    float* sk = (sk_floats && B <= 32) ? ws.pool : nullptr;         // M <= 32: the projections run as weight streams
    const size_t layer_stride = (size_t)B * t_cap * 2 * d;
    size_t cbytes = 0;                                              // split-K ticket counters of the projections: cleared once per
    void* cnt = sk ? gemm_skinny_counters(sk, &cbytes) : nullptr;   // step, by the step's first kernel
    for (int l = 0; l < cfg->n_layer; ++l) {
        const r4d_gpt2_layer& L = w->layers[l];
        R4D_REQUIRE(L.ln_1_w && L.c_attn_w && L.attn_proj_w && L.ln_2_w && L.c_fc_w && L.mlp_proj_w,
                    "gpt2 decode: null weight in layer %d", l);
        // M <= 32 and d in {512, 768}: LayerNorm runs inside the projection that reads it (gemm_skinny8_kernel) -- x stays
        // the un-normalised residual stream and two launches per layer disappear
        const bool fuse_ln = sk && L.c_attn_wT && L.c_fc_wT && gemm_skinny_fuses_ln(B, d, 3 * d);
        if (l == 0 && !x_ready)
            rc = launch_embed_pos_layernorm(ids_d, inputs_embeds_d, pos_d, w->wte, w->wpe, cfg->vocab, cfg->n_positions,
                                            t_cap, B, d, L.ln_1_w, L.ln_1_b, cfg->ln_eps, ws.x, ws.ln, s, cnt, cbytes);
        else if (!fuse_ln)
            rc = launch_layernorm(ws.x, L.ln_1_w, L.ln_1_b, B, d, cfg->ln_eps, ws.ln, s);
        if (rc) return rc;
        if (fuse_ln && (l > 0 || x_ready))
            rc = (L.c_attn_wTg && L.c_attn_lnc)            // LayerNorm pre-folded into a decode-only copy of the weight
                     ? launch_gemm_skinny(ws.x, L.c_attn_wTg, L.c_attn_b, nullptr, B, d, 3 * d, EPI_NONE, ws.qkv, sk, s, nullptr, nullptr,
                                          cfg->ln_eps, true, L.c_attn_lnc)
                     : launch_gemm_skinny(ws.x, L.c_attn_wT, L.c_attn_b, nullptr, B, d, 3 * d, EPI_NONE, ws.qkv, sk, s, L.ln_1_w,
                                          L.ln_1_b, cfg->ln_eps, true);
        else
            rc = conv1d(ws.ln, L.c_attn_w, L.c_attn_wT, L.c_attn_b, nullptr, B, d, 3 * d, EPI_NONE, ws.qkv, s, sk);
        if (rc) return rc;
        if ((rc = launch_decode_attention(ws.qkv, kv_cache_d + (size_t)l * layer_stride, pos_d, B, t_cap, H, d, ws.att, s)))
            return rc;
        if ((rc = conv1d(ws.att, L.attn_proj_w, L.attn_proj_wT, L.attn_proj_b, ws.x, B, d, d, EPI_RESIDUAL, ws.x, s, sk, true))) return rc;
        if (fuse_ln) {
            rc = (L.c_fc_wTg && L.c_fc_lnc)
                     ? launch_gemm_skinny(ws.x, L.c_fc_wTg, L.c_fc_b, nullptr, B, d, 4 * d, EPI_GELU, ws.fc, sk, s, nullptr, nullptr,
                                          cfg->ln_eps, true, L.c_fc_lnc)
                     : launch_gemm_skinny(ws.x, L.c_fc_wT, L.c_fc_b, nullptr, B, d, 4 * d, EPI_GELU, ws.fc, sk, s, L.ln_2_w, L.ln_2_b,
                                          cfg->ln_eps, true);
        } else {
            if ((rc = launch_layernorm(ws.x, L.ln_2_w, L.ln_2_b, B, d, cfg->ln_eps, ws.ln, s))) return rc;
            rc = conv1d(ws.ln, L.c_fc_w, L.c_fc_wT, L.c_fc_b, nullptr, B, d, 4 * d, EPI_GELU, ws.fc, s, sk);
        }
        if (rc) return rc;
        if ((rc = conv1d(ws.fc, L.mlp_proj_w, L.mlp_proj_wT, L.mlp_proj_b, ws.x, B, 4 * d, d, EPI_RESIDUAL, ws.x, s, sk, true))) return rc;
    }
    return launch_layernorm(ws.x, w->ln_f_w, w->ln_f_b, B, d, cfg->ln_eps, out_hidden_d, s);
}

int r4d_gpt2_decode_step_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const int64_t* ids_d,
                             const float* inputs_embeds_d, const int32_t* pos_d, float* kv_cache_d, int32_t B,
                             int32_t t_cap, float* out_hidden_d, void* workspace_d, size_t workspace_bytes,
                             void* stream) {
    return decode_step_impl(cfg, w, ids_d, inputs_embeds_d, pos_d, kv_cache_d, B, t_cap, out_hidden_d, workspace_d, workspace_bytes,
                            stream, false);
}

static size_t greedy_pool_floats(const r4d_gpt2_config* cfg) {
    const size_t a = decode_skinny_floats(cfg->n_embd);
    const size_t b = gemm_skinny_supported(1, cfg->n_embd, cfg->vocab) ? gemm_skinny_scratch_floats(cfg->n_embd, cfg->vocab) : 0;
    return a > b ? a : b;
}

size_t r4d_gpt2_greedy_workspace_bytes(const r4d_gpt2_config* cfg, int32_t B) {
    if (!cfg || B <= 0) return 0;
    return carve(nullptr, (size_t)B, 0, greedy_pool_floats(cfg), cfg->n_embd).bytes;
}

int r4d_gpt2_greedy_step_f32(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_greedy_state* st,
                             float* kv_cache_d, int32_t B, int32_t t_cap, void* workspace_d, size_t workspace_bytes,
                             void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    R4D_REQUIRE(w && w->wte && st && st->last_d && st->logits_d, "gpt2 greedy: null pointer");
    R4D_REQUIRE(B >= 1 && t_cap >= 1, "gpt2 greedy: B=%d t_cap=%d", B, t_cap);
    const int d = cfg->n_embd, V = cfg->vocab;
    Workspace ws = carve(workspace_d, (size_t)B, 0, greedy_pool_floats(cfg), d);
    if (!workspace_d || workspace_bytes < ws.bytes) {
        set_error("gpt2 greedy: workspace %zu bytes < required %zu", workspace_bytes, ws.bytes);
        return R4D_ERR_WORKSPACE;
    }
    const float* head = w->lm_head ? w->lm_head : w->wte;            // untied checkpoints carry their own lm_head.weight
    if (B <= 32 && gemm_skinny_supported(B, d, V))                   // lm_head on B rows: a weight stream like the projections
        rc = launch_gemm_skinny(st->last_d, head, nullptr, nullptr, B, d, V, EPI_NONE, st->logits_d, ws.pool, s);
    else
        rc = r4d_lm_logits_f32(st->last_d, head, B, V, d, st->logits_d, stream);
    if (rc) return rc;
    GreedyState g;
    g.next = st->next_d; g.lens = st->lens_d; g.pos = st->pos_d; g.active = st->active_d; g.gen_len = st->gen_len_d;
    g.out_tokens = st->out_tokens_d; g.params = st->params_d; g.out_cap = st->out_cap; g.t_cap = t_cap;
    // the bookkeeping kernel also writes the chosen token's input row into the step's x buffer and clears the ticket counters
    GreedyEmbed ge;
    ge.wte = w->wte; ge.wpe = w->wpe; ge.x_out = ws.x; ge.vocab = cfg->vocab; ge.n_positions = cfg->n_positions; ge.d = d;
    size_t cbytes = 0;
    void* cnt = ws.pool ? gemm_skinny_counters(ws.pool, &cbytes) : nullptr;
    ge.zero_words = (unsigned*)cnt; ge.n_zero = (int)(cbytes / 4);
    if ((rc = launch_greedy_advance(st->logits_d, B, V, g, s, &ge))) return rc;
    return decode_step_impl(cfg, w, st->next_d, nullptr, st->pos_d, kv_cache_d, B, t_cap, st->last_d, workspace_d, workspace_bytes,
                            stream, true);
}

struct r4d_decode_graph { hipGraph_t graph; hipGraphExec_t exec; };

int r4d_gpt2_greedy_graph_create(const r4d_gpt2_config* cfg, const r4d_gpt2_weights* w, const r4d_greedy_state* st,
                                 float* kv_cache_d, int32_t B, int32_t t_cap, void* workspace_d, size_t workspace_bytes,
                                 r4d_decode_graph** out_graph) {
    R4D_REQUIRE(out_graph, "gpt2 greedy graph: null out_graph");
    *out_graph = nullptr;
    R4D_REQUIRE(!g_prof_on, "gpt2 greedy graph: switch the launch profiler off before capturing");
    hipStream_t cap = nullptr;                                       // the caller's stream may be the null stream: not capturable
    R4D_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(cap);
        set_error("gpt2 greedy graph: begin capture: %s", hipGetErrorString(e));
        return R4D_ERR_HIP;
    }
    const int rc = r4d_gpt2_greedy_step_f32(cfg, w, st, kv_cache_d, B, t_cap, workspace_d, workspace_bytes, cap);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(cap, &graph);                            // always end the capture, also after an error
    (void)hipStreamDestroy(cap);
    if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess || !graph) {
        set_error("gpt2 greedy graph: end capture: %s", hipGetErrorString(e));
        return R4D_ERR_HIP;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        set_error("gpt2 greedy graph: instantiate: %s", hipGetErrorString(e));
        return R4D_ERR_HIP;
    }
    *out_graph = new r4d_decode_graph{graph, exec};
    return R4D_OK;
}

int r4d_decode_graph_launch(r4d_decode_graph* graph, int32_t n_steps, void* stream) {
    R4D_REQUIRE(graph && graph->exec && n_steps >= 0, "decode graph: null graph or n_steps=%d", n_steps);
    for (int i = 0; i < n_steps; ++i) R4D_HIP(hipGraphLaunch(graph->exec, (hipStream_t)stream));
    return R4D_OK;
}

void r4d_decode_graph_destroy(r4d_decode_graph* graph) {
    if (!graph) return;
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
}

int r4d_lm_logits_f32(const float* hidden_d, const float* wte_d, int32_t M, int32_t V, int32_t d, float* logits_d,
                      void* stream) {
    R4D_REQUIRE(hidden_d && wte_d && logits_d, "lm_logits: null pointer");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = hidden_d; g.B = wte_d; g.C = logits_d;
    g.M = M; g.N = V; g.K = d; g.lda = d; g.ldb = d; g.ldc = V;
    g.b_trans = 1; g.b_rows = V; g.nbatch = 1; g.nb1 = 1; g.epilogue = EPI_NONE; g.scale_div = 1.f;
    return launch_gemm_f32(g, (hipStream_t)stream);
}

int r4d_layernorm_f32(const float* x_d, const float* w_d, const float* b_d, int32_t rows, int32_t d, float eps,
                      float* y_d, void* stream) {
    R4D_REQUIRE(x_d && w_d && b_d && y_d, "layernorm: null pointer");
    return launch_layernorm(x_d, w_d, b_d, rows, d, eps, y_d, (hipStream_t)stream);
}

int r4d_layernorm_lines_f32(const float* x_d, const float* w_d, const float* b_d, int32_t rows, int32_t d, float eps,
                            uint16_t* y_lines_d, void* stream) {
    R4D_REQUIRE(x_d && w_d && b_d && y_lines_d, "layernorm_lines: null pointer");
    return launch_layernorm_lines(x_d, w_d, b_d, rows, d, eps, y_lines_d, (hipStream_t)stream);
}

int r4d_conv1d_f32(const float* x_d, const float* w_d, const float* w_t_d, const float* bias_d, const float* residual_d,
                   int32_t M, int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream) {
    R4D_REQUIRE(x_d && w_d && y_d, "conv1d: null pointer");
    R4D_REQUIRE(epilogue >= 0 && epilogue <= 2, "conv1d: epilogue %d not in {0,1,2}", epilogue);
    R4D_REQUIRE(epilogue != EPI_RESIDUAL || residual_d, "conv1d: residual epilogue needs residual_d");
    return conv1d(x_d, w_d, w_t_d, bias_d, residual_d, M, K, N, epilogue, y_d, (hipStream_t)stream);
}

size_t r4d_attention_workspace_bytes(int32_t B, int32_t n_head, int32_t T) {
    if (B <= 0 || n_head <= 0 || T <= 0) return 0;
    return align_up((size_t)B * n_head * T * tpad(T) * sizeof(float), 256);
}

int r4d_attention_f32(const float* qkv_d, int32_t B, int32_t T, int32_t n_head, int32_t d, float* a_d,
                      void* scores_ws_d, size_t ws_bytes, void* stream) {
    R4D_REQUIRE(qkv_d && a_d && scores_ws_d, "attention: null pointer");
    R4D_REQUIRE(n_head >= 1 && d % n_head == 0 && (d / n_head) % 16 == 0, "attention: head_dim must be a multiple of 16");
    R4D_REQUIRE(T >= 1 && T <= 1024, "attention: T=%d out of range", T);
    if (ws_bytes < r4d_attention_workspace_bytes(B, n_head, T)) {
        set_error("attention: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    return attention(qkv_d, B, T, n_head, d, a_d, (float*)scores_ws_d, (hipStream_t)stream);
}

}  // extern "C"
