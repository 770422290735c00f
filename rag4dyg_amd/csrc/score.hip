// Retrieval scoring on gfx950: row normalisation, query-vs-pool cosine scan (fp32 MFMA), device top-k
// with the canonical (score desc, index asc) order, multi-shard merge and full-row stable ranking.
// Reference: train/train_retriever.py:433-438 (normalise, matmul, (x+1)/2) and :357-358,461-467 (argsort).
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "common.h"

namespace r4d {

// ------------------------------------------------------------------------------------ normalise rows
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, int n, int d,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = x + (long long)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += p[c] * p[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float nrm = sqrtf(s);                       // x / x.norm(dim=1, keepdim=True): no eps
    for (int c = lane; c < d; c += 64) out[(long long)row * d + c] = p[c] / nrm;
}

// ------------------------------------------------------------------------------------ pool scan (Q <= 64)
// S[q, n] = (q_hat[q] . pool_hat[n] + 1) / 2 for one block of <= 32 queries against the whole pool shard:
// train_retriever.py:437-438 at the reference's query batch (32).  At Q_b = 32 the scan is HBM-READ bound
// (16 flop per pool byte), so the kernel is built around streaming pool_hat exactly once:
//   * the 32 normalised queries sit in LDS k-major (Qs[k][q], d*132 B) for the whole kernel; an MFMA A operand
//     is one conflict-free ds_read_b32 (row stride 33 floats; 12 % of the LDS read rate at full MFMA issue);
//   * every wavefront owns whole 32-row pool tiles and the full d, so there is no cross-wave reduction and no
//     barrier in the loop; lane (j, h) streams row j of its tile with 16-byte global loads, the lane pair of a row
//     consuming each 128-byte line at once (k permuted identically on both operands), 16 KB per wave in flight;
//   * exact-f32 MFMA 32x32x2 accumulates S[32 q x 32 rows]; the epilogue applies (x+1)/2 and writes 128-byte
//     row segments of the score matrix, from which topk_seg_kernel selects.
typedef float f32x16s __attribute__((ext_vector_type(16)));


// Q tile layout in LDS: ROW-major [32 queries][d + 4] (k contiguous).  A lane's MFMA A operands for four consecutive
// MFMAs are then ONE ds_read_b128 (16-lane groups hit 16 distinct 16-byte slots: stride 4 banks), matching the four k
// of its 16-byte pool load; the k-major layout needed a ds_read2_b32 + lgkmcnt(0) wait in front of every MFMA pair.
template <int NW>                                    // wavefronts per workgroup (they share the Q tile)
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void pool_scan_kernel(const float* __restrict__ qhat, const float* __restrict__ pool,
                                                        int Q, int N, int d, float* __restrict__ scores) {
    extern __shared__ float Qs[];                       // [32][d + 4]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.y * 32;
    const int ldq = d + 4;
    for (int q = wid; q < 32; q += NW) {                // one query row per wave and trip, 16-byte pieces
        const bool ok = q0 + q < Q;
        const float4* src = reinterpret_cast<const float4*>(qhat + (long long)min(q0 + q, Q - 1) * d);
        for (int k4 = lane; k4 < d / 4; k4 += 64) {
            float4 v = src[k4];
            if (!ok) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
            *reinterpret_cast<float4*>(Qs + q * ldq + 4 * k4) = v;
        }
    }
    __syncthreads();
    const int ntiles = (N + 31) / 32;
    // k order: lane half h of row j owns the 64-byte halves [32g + 16h, 32g + 16h + 16) of the row, i.e. the lane
    // pair (j,0),(j,1) consumes each 128-byte line of the row at once (four 16-byte loads per lane per line);
    // component c of load u of group g is k = 32g + 16h + 4u + c on BOTH operands.  d % 32 == 0 here
    // (r4d_score_topk_f32 falls back to the GEMM otherwise).
    const int ngroups = d / 32;
    const float* qfrag = Qs + li * ldq + 16 * lh;       // + 32g + 4u
    for (int t = blockIdx.x * NW + wid; t < ntiles; t += gridDim.x * NW) {
        const int row = t * 32 + li;
        const float4* __restrict__ prow =
            reinterpret_cast<const float4*>(pool + (long long)min(row, N - 1) * d) + 4 * lh;   // clamped: always valid
        f32x16s acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // NB groups (NB x 64 B per lane) in flight; the slot of group g is refilled with group g + NB right after its
        // MFMAs (program order: no register copies).  NB = 3 and no Q look-ahead keep the kernel under 128 registers:
        // four waves per SIMD (two 8-wave workgroups per CU) hide more latency than a deeper per-wave queue.
        constexpr int NB = NW == 8 ? 3 : 4;
        float4 b[NB][4];
#pragma unroll
        for (int s_ = 0; s_ < NB; ++s_)
#pragma unroll
            for (int u = 0; u < 4; ++u) b[s_][u] = prow[8 * min(s_, ngroups - 1) + u];
        for (int g0 = 0; g0 < ngroups; g0 += NB) {
#pragma unroll
            for (int s_ = 0; s_ < NB; ++s_) {
                const int g = g0 + s_;
                if (g < ngroups) {                      // wave-uniform
                    float4 qa[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) qa[u] = *reinterpret_cast<const float4*>(qfrag + 32 * g + 4 * u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u].x, b[s_][u].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u].y, b[s_][u].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u].z, b[s_][u].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[u].w, b[s_][u].w, acc, 0, 0, 0);
                    }
                    const int gl = min(g + NB, ngroups - 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[s_][u] = prow[8 * gl + u];
                }
            }
        }
        if (row < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (q < Q) scores[(long long)q * N + row] = (acc[r] + 1.0f) / 2.0f;
            }
        }
    }
}

static int launch_pool_scan(const float* qhat, const float* pool, int Q, int N, int d, float* scores, hipStream_t s) {
    const size_t lds = (size_t)32 * (d + 4) * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute((const void*)pool_scan_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    140 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)pool_scan_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    140 * 1024) != hipSuccess) {
                set_error("pool_scan: cannot raise dynamic LDS limit");
                return R4D_ERR_HIP;
            }
            raised = true;
        }
    }
    const int ntiles = cdiv(N, 32);
    const int blocks_per_cu = lds > 80 * 1024 ? 1 : 2;
    // algorithmic bytes (SURVEY 8d B_score): pool read once + queries + score rows out
    ProfScope prof(PK_POOL_SCAN, 4.0 * N * d * cdiv(Q, 32) + 4.0 * Q * d + 4.0 * (double)Q * N, s);
    static int nw = -1;
    if (nw < 0) { const char* e = getenv("R4D_SCAN_WAVES"); nw = e ? atoi(e) : 8; }   // tuning aid: 4 or 8
    if (nw == 8 && ntiles >= 8 * 256) {
        const int gx = max(1, min(cdiv(ntiles, 8), 256 * blocks_per_cu));
        hipLaunchKernelGGL(pool_scan_kernel<8>, dim3(gx, cdiv(Q, 32)), dim3(512), lds, s, qhat, pool, Q, N, d, scores);
    } else {
        const int gx = max(1, min(cdiv(ntiles, 4), 256 * blocks_per_cu));
        hipLaunchKernelGGL(pool_scan_kernel<4>, dim3(gx, cdiv(Q, 32)), dim3(256), lds, s, qhat, pool, Q, N, d, scores);
    }
    R4D_CHECK_LAUNCH("pool_scan");
    return R4D_OK;
}

// ------------------------------------------------------------------------------------ top-k machinery
// Candidate order == np.argsort(-x, kind='stable'): larger score first, ties by smaller index.  NaN
// sorts last (as in numpy).
template <typename T>
struct Cand {
    T v;
    long long i;
};
template <typename T>
__device__ __forceinline__ bool better(const Cand<T>& a, const Cand<T>& b) {
    return (a.v > b.v) || (a.v == b.v && a.i < b.i);
}
template <typename T>
__device__ __forceinline__ Cand<T> shfl_xor_cand(const Cand<T>& c, int o) {
    Cand<T> r;
    r.v = __shfl_xor(c.v, o, 64);
    r.i = __shfl_xor(c.i, o, 64);
    return r;
}
template <typename T>
__device__ __forceinline__ T neg_inf() { return (T)(-INFINITY); }

constexpr int TOPK_E = 16;                 // candidates per lane
constexpr int TOPK_SEG = 64 * TOPK_E;      // candidates per WAVEFRONT segment

// Candidate inside a segment: value + POSITION in the row / candidate list.  Positions order ties exactly like
// global indices do (level 1: position == column; deeper levels: candidate lists are laid out by (segment, rank) and
// every segment is already sorted by (value desc, index asc)), so the reduction carries 32-bit positions and the
// 64-bit global index is looked up only for the k winners.
template <typename T>
struct WCand {
    T v;
    unsigned int p;
};
template <typename T>
__device__ __forceinline__ bool wbetter(const WCand<T>& a, const WCand<T>& b) {
    return (a.v > b.v) | ((a.v == b.v) & (a.p < b.p));          // branch-free: || / && compile to exec-mask branches
}

// One wavefront reduces one segment of 1024 candidates of one row to its k best: k rounds of a 6-step shuffle
// arg-best over (value, position) pairs with removal -- no LDS, no barrier.  Every lane keeps its current local best;
// only the round's winner rescans its 16 registers.
//   vals [rows, ld]; idx_in (nullable) parallel global indices, else index = column + index_offset.
//   out_v / out_i [rows, nseg*k]
template <typename T>
__global__ __launch_bounds__(256) void topk_seg_kernel(const T* __restrict__ vals, const long long* __restrict__ idx_in,
                                                       int n, long long ld, int k, long long index_offset, int nseg,
                                                       int rows, T* __restrict__ out_v, long long* __restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);        // (row, segment) of this wave
    if (unit >= (long long)rows * nseg) return;
    const int row = (int)(unit / nseg), seg = (int)(unit % nseg);
    const T* v = vals + (long long)row * ld;
    const WCand<T> none = {neg_inf<T>(), 0xffffffffu};
    WCand<T> c[TOPK_E];
#pragma unroll
    for (int e = 0; e < TOPK_E; ++e) {
        const int col = seg * TOPK_SEG + e * 64 + lane;
        const T x = v[min(col, n - 1)];                                 // clamped address, value masked below
        c[e].v = (col < n && x == x) ? x : neg_inf<T>();                // NaN sorts last (as numpy)
        c[e].p = (col < n) ? (unsigned)col : 0xffffffffu;
    }
    auto local_best = [&](int& slot) {
        WCand<T> b = none;
        slot = -1;
#pragma unroll
        for (int e = 0; e < TOPK_E; ++e) {
            const bool bt = wbetter(c[e], b);
            b.v = bt ? c[e].v : b.v; b.p = bt ? c[e].p : b.p; slot = bt ? e : slot;
        }
        return b;
    };
    int slot;
    WCand<T> mine = local_best(slot);
    T* ov = out_v + ((long long)row * nseg + seg) * k;
    long long* oi = out_i + ((long long)row * nseg + seg) * k;
    const long long* ii = idx_in ? idx_in + (long long)row * ld : nullptr;
    WCand<T> won = none;                                                // lane r keeps the r-th winner (k <= 64)
    for (int r = 0; r < k; ++r) {
        WCand<T> w = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            WCand<T> t;
            t.v = __shfl_xor(w.v, o, 64);
            t.p = __shfl_xor(w.p, o, 64);
            const bool bt = wbetter(t, w);
            w.v = bt ? t.v : w.v; w.p = bt ? t.p : w.p;
        }
        if (lane == r) won = w;
        if (__any(slot >= 0 && mine.p == w.p)) {                        // positions are unique: exactly one owner
            const bool own = slot >= 0 && mine.p == w.p;
#pragma unroll
            for (int e = 0; e < TOPK_E; ++e) {
                const bool kill = own & (e == slot);
                c[e].v = kill ? none.v : c[e].v; c[e].p = kill ? none.p : c[e].p;
            }
            int s2;
            const WCand<T> nb = local_best(s2);
            mine.v = own ? nb.v : mine.v; mine.p = own ? nb.p : mine.p; slot = own ? s2 : slot;
        }
    }
    // the k index look-ups and stores go out in parallel, off the selection loop's critical path
    if (lane < k) {
        const bool pad = won.p == 0xffffffffu;
        const long long gi = ii ? ii[pad ? 0 : won.p] : (long long)won.p + index_offset;
        ov[lane] = won.v;
        oi[lane] = pad ? 0x7fffffffffffffffLL : gi;
    }
}

template <typename T>
static size_t topk_ws_bytes(int rows, int n, int k) {
    size_t total = 0;
    long long cur = n;
    while (cur > TOPK_SEG) {
        const long long nseg = (cur + TOPK_SEG - 1) / TOPK_SEG;
        total += align_up((size_t)rows * nseg * k * sizeof(T), 256) + align_up((size_t)rows * nseg * k * 8, 256);
        cur = nseg * k;
    }
    return total + 256;
}

// rows x n values -> rows x k best (value, index).  Multi-level: wavefront segments of 1024 candidates.
template <typename T>
static int topk_rows(const T* vals, int rows, int n, long long ld, int k, long long index_offset, T* out_v,
                     long long* out_i, void* ws, size_t ws_bytes, hipStream_t s) {
    R4D_REQUIRE(k >= 1 && k <= 64 && k <= n, "topk: k=%d must be in [1, min(64, n=%d)]", k, n);
    R4D_REQUIRE(rows >= 1, "topk: no rows");
    if (ws_bytes < topk_ws_bytes<T>(rows, n, k)) {
        set_error("topk: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    const T* cv = vals;
    const long long* ci = nullptr;
    long long cur = n, cld = ld;
    char* wp = (char*)ws;
    while (true) {
        const int nseg = (int)((cur + TOPK_SEG - 1) / TOPK_SEG);
        T* ov;
        long long* oi;
        if (nseg == 1) { ov = out_v; oi = out_i; }
        else {
            ov = (T*)wp; wp += align_up((size_t)rows * nseg * k * sizeof(T), 256);
            oi = (long long*)wp; wp += align_up((size_t)rows * nseg * k * 8, 256);
        }
        const long long units = (long long)rows * nseg;
        ProfScope prof(PK_TOPK, (double)rows * cur * (sizeof(T) + (ci ? 8 : 0)), s);
        hipLaunchKernelGGL((topk_seg_kernel<T>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, cv, ci, (int)cur, cld,
                           k, index_offset, nseg, rows, ov, oi);
        R4D_CHECK_LAUNCH("topk_seg");
        if (nseg == 1) break;
        cv = ov; ci = oi; cur = (long long)nseg * k; cld = cur;
    }
    return R4D_OK;
}

// int64 -> int32 index narrowing for the f64 (Jaccard) API
__global__ void narrow_idx_kernel(const long long* __restrict__ in, int32_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// ------------------------------------------------------------------------------------ shard merge
// vals/idx [G,Q,k] -> row-major candidate lists [Q, G*k] (shard-major, so positions order ties like global indices),
// then the same wavefront top-k.
__global__ __launch_bounds__(256) void gather_candidates_kernel(const float* __restrict__ vals,
                                                                const long long* __restrict__ idx, int G, int Q, int k,
                                                                float* __restrict__ ov, long long* __restrict__ oi) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)G * Q * k) return;
    const int j = (int)(t % k), g = (int)((t / k) % G), q = (int)(t / ((long long)k * G));
    const long long src = ((long long)g * Q + q) * k + j;
    ov[t] = vals[src];
    oi[t] = idx[src];
}

// ------------------------------------------------------------------------------------ full-row ranking
// perm = stable argsort of -scores by rank counting: rank(i) = #{j : better(j, i)}.  O(n^2) compares per
// row, embarrassingly parallel and exactly np.argsort(-x, kind='stable') (file-compat mode only: the
// reference writes the full permutation of the pool for every query, train_retriever.py:357-362).
template <typename T>
__global__ __launch_bounds__(256) void rank_count_kernel(const T* __restrict__ scores, int n, int32_t* __restrict__ perm) {
    __shared__ T tile[1024];
    const int row = blockIdx.y;
    const T* v = scores + (long long)row * n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    T mine = (i < n) ? v[i] : (T)0;
    if (mine != mine) mine = neg_inf<T>();
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            T x = (j0 + t < n) ? v[j0 + t] : neg_inf<T>();
            tile[t] = (x != x) ? neg_inf<T>() : x;
        }
        __syncthreads();
        const int lim = min(1024, n - j0);
        for (int t = 0; t < lim; ++t) {
            const T x = tile[t];
            rank += (x > mine) || (x == mine && (j0 + t) < i);
        }
    }
    if (i < n) perm[(long long)row * n + rank] = i;
}

template <typename T>
static int argsort_desc(const T* scores, int rows, int n, int32_t* perm, hipStream_t s) {
    R4D_REQUIRE(scores && perm, "argsort: null pointer");
    R4D_REQUIRE(rows >= 0 && rows <= 65535 && n >= 1 && n <= 65536, "argsort: rows=%d (<= 65535), n=%d (<= 65536) out of range", rows, n);
    if (rows == 0) return R4D_OK;
    ProfScope prof(PK_RANK_COUNT, (double)rows * n * (sizeof(T) + 4), s);
    hipLaunchKernelGGL((rank_count_kernel<T>), dim3(cdiv(n, 256), rows), dim3(256), 0, s, scores, n, perm);
    R4D_CHECK_LAUNCH("rank_count");
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_normalize_rows_f32(const float* x_d, int32_t n, int32_t d, float* out_d, void* stream) {
    R4D_REQUIRE(x_d && out_d && n >= 0 && d >= 1, "normalize_rows: bad arguments");
    if (n == 0) return R4D_OK;
    ProfScope prof(PK_NORMALIZE, 8.0 * n * d, (hipStream_t)stream);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x_d, n, d, out_d);
    R4D_CHECK_LAUNCH("normalize_rows");
    return R4D_OK;
}

size_t r4d_score_topk_workspace_bytes(int32_t Q, int32_t N, int32_t k) {
    if (Q <= 0 || N <= 0 || k <= 0) return 0;
    return align_up((size_t)Q * N * sizeof(float), 256) + topk_ws_bytes<float>(Q, N, k);
}

int r4d_score_topk_f32(const float* q_hat_d, const float* pool_hat_d, int32_t Q, int32_t N, int32_t d, int32_t k,
                       int64_t index_offset, float* out_val_d, int64_t* out_idx_d, float* out_scores_d,
                       void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(q_hat_d && pool_hat_d && out_val_d && out_idx_d, "score_topk: null pointer");
    R4D_REQUIRE(Q >= 1 && Q <= 65535 && N >= 1 && d >= 4 && d % 4 == 0, "score_topk: bad shape Q=%d N=%d d=%d", Q, N, d);
    R4D_REQUIRE(k >= 1 && k <= 64 && k <= N, "score_topk: k=%d must be in [1, min(64, N=%d)]", k, N);
    if (!workspace_d || workspace_bytes < r4d_score_topk_workspace_bytes(Q, N, k)) {
        set_error("score_topk: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* scores = out_scores_d ? out_scores_d : (float*)workspace_d;
    char* ws = (char*)workspace_d + align_up((size_t)Q * N * sizeof(float), 256);
    int rc;
    if (Q <= 64 && d % 32 == 0 && d <= 1024) {
        rc = launch_pool_scan(q_hat_d, pool_hat_d, Q, N, d, scores, s);         // HBM-bound regime: stream the pool once
    } else {
        GemmArgs g;                                                             // MFMA-bound regime: tiled GEMM
        memset(&g, 0, sizeof(g));
        g.A = q_hat_d; g.B = pool_hat_d; g.C = scores;
        g.M = Q; g.N = N; g.K = d; g.lda = d; g.ldb = d; g.ldc = N;
        g.b_trans = 1; g.b_rows = N; g.nbatch = 1; g.nb1 = 1; g.epilogue = EPI_HALF_PLUS; g.scale_div = 1.f;
        rc = launch_gemm_f32(g, s);
    }
    if (rc) return rc;
    return topk_rows<float>(scores, Q, N, N, k, index_offset, out_val_d, (long long*)out_idx_d, ws,
                            workspace_bytes - align_up((size_t)Q * N * sizeof(float), 256), s);
}

int r4d_topk_f32(const float* m_d, int32_t rows, int32_t n, int32_t k, float* out_val_d, int64_t* out_idx_d,
                 void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(m_d && out_val_d && out_idx_d && workspace_d, "topk_f32: null pointer");
    R4D_REQUIRE(rows >= 1 && n >= 1, "topk_f32: empty input");
    return topk_rows<float>(m_d, rows, n, n, k, 0, out_val_d, (long long*)out_idx_d, workspace_d, workspace_bytes,
                            (hipStream_t)stream);
}

size_t r4d_merge_topk_workspace_bytes(int32_t G, int32_t Q, int32_t k) {
    if (G <= 0 || Q <= 0 || k <= 0) return 0;
    return align_up((size_t)G * Q * k * 4, 256) + align_up((size_t)G * Q * k * 8, 256) + topk_ws_bytes<float>(Q, G * k, k);
}

int r4d_merge_topk_f32(const float* vals_d, const int64_t* idx_d, int32_t G, int32_t Q, int32_t k, float* out_val_d,
                       int64_t* out_idx_d, void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(vals_d && idx_d && out_val_d && out_idx_d && workspace_d, "merge_topk: null pointer");
    R4D_REQUIRE(G >= 1 && Q >= 1 && k >= 1 && k <= 64, "merge_topk: G=%d Q=%d k=%d out of range", G, Q, k);
    if (workspace_bytes < r4d_merge_topk_workspace_bytes(G, Q, k)) {
        set_error("merge_topk: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const long long tot = (long long)G * Q * k;
    float* cv = (float*)workspace_d;
    long long* ci = (long long*)((char*)workspace_d + align_up((size_t)tot * 4, 256));
    char* ws = (char*)ci + align_up((size_t)tot * 8, 256);
    {
        ProfScope prof(PK_MERGE_TOPK, 24.0 * tot, s);
        hipLaunchKernelGGL(gather_candidates_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, vals_d,
                           (const long long*)idx_d, G, Q, k, cv, ci);
        R4D_CHECK_LAUNCH("gather_candidates");
    }
    // level 1 of topk_rows must read the gathered indices: run it by hand with idx_in = ci
    const float* v = cv;
    const long long* ii = ci;
    long long cur = (long long)G * k, cld = cur;
    while (true) {
        const int nseg = (int)((cur + TOPK_SEG - 1) / TOPK_SEG);
        float* ov;
        long long* oi;
        if (nseg == 1) { ov = out_val_d; oi = (long long*)out_idx_d; }
        else {
            ov = (float*)ws; ws += align_up((size_t)Q * nseg * k * 4, 256);
            oi = (long long*)ws; ws += align_up((size_t)Q * nseg * k * 8, 256);
        }
        const long long units = (long long)Q * nseg;
        ProfScope prof(PK_TOPK, (double)Q * cur * 12.0, s);
        hipLaunchKernelGGL((topk_seg_kernel<float>), dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, v, ii, (int)cur,
                           cld, k, 0LL, nseg, Q, ov, oi);
        R4D_CHECK_LAUNCH("topk_seg(merge)");
        if (nseg == 1) break;
        v = ov; ii = oi; cur = (long long)nseg * k; cld = cur;
    }
    return R4D_OK;
}

int r4d_argsort_desc_f32(const float* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* stream) {
    return argsort_desc<float>(scores_d, rows, n, perm_d, (hipStream_t)stream);
}
int r4d_argsort_desc_f64(const double* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* stream) {
    return argsort_desc<double>(scores_d, rows, n, perm_d, (hipStream_t)stream);
}

size_t r4d_topk_f64_workspace_bytes(int32_t rows, int32_t n, int32_t k) {
    if (rows <= 0 || n <= 0 || k <= 0) return 0;
    return topk_ws_bytes<double>(rows, n, k) + align_up((size_t)rows * k * 8, 256);
}

int r4d_topk_f64(const double* m_d, int32_t rows, int32_t n, int32_t k, double* out_val_d, int32_t* out_idx_d,
                 void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(m_d && out_val_d && out_idx_d && workspace_d, "topk_f64: null pointer");
    R4D_REQUIRE(rows >= 1 && n >= 1, "topk_f64: empty input");
    if (workspace_bytes < r4d_topk_f64_workspace_bytes(rows, n, k)) {
        set_error("topk_f64: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    long long* idx64 = (long long*)workspace_d;
    char* ws = (char*)workspace_d + align_up((size_t)rows * k * 8, 256);
    int rc = topk_rows<double>(m_d, rows, n, n, k, 0, out_val_d, idx64, ws,
                               workspace_bytes - align_up((size_t)rows * k * 8, 256), s);
    if (rc) return rc;
    const long long tot = (long long)rows * k;
    hipLaunchKernelGGL(narrow_idx_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, idx64, out_idx_d, tot);
    R4D_CHECK_LAUNCH("narrow_idx");
    return R4D_OK;
}

}  // extern "C"
