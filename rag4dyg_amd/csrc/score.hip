// Retrieval scoring on gfx950: row normalisation, query-vs-pool cosine scan (fp32 MFMA), device top-k
// with the canonical (score desc, index asc) order, multi-shard merge and full-row stable ranking.
// Reference: train/train_retriever.py:433-438 (normalise, matmul, (x+1)/2) and :357-358,461-467 (argsort).
#include <math.h>
#include <string.h>
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
//     barrier in the loop; lane (j, h) streams row j of its tile with 16-byte global loads (k permuted
//     identically on both operands: component c of load s is k = 8s + 4h + c), 8 loads = 8 KB per wave in flight;
//   * exact-f32 MFMA 32x32x2 accumulates S[32 q x 32 rows]; the epilogue applies (x+1)/2 and writes 128-byte
//     row segments of the score matrix, from which topk_seg_kernel selects.
typedef float f32x16s __attribute__((ext_vector_type(16)));
constexpr int SCAN_U = 8;

constexpr int SCAN_LDQ = 33;                          // Qs row stride: conflict-free fill AND fragment reads

__global__ __launch_bounds__(256) void pool_scan_kernel(const float* __restrict__ qhat, const float* __restrict__ pool,
                                                        int Q, int N, int d, float* __restrict__ scores) {
    extern __shared__ float Qs[];                       // [d][33]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.y * 32;
    for (int q = 0; q < 32; ++q) {
        const bool ok = q0 + q < Q;
        const float* src = qhat + (long long)min(q0 + q, Q - 1) * d;
        for (int k = tid; k < d; k += 256) {
            const float v = src[k];
            Qs[k * SCAN_LDQ + q] = ok ? v : 0.f;
        }
    }
    __syncthreads();
    const int ntiles = (N + 31) / 32;
    const int nsteps = d / 8;
    for (int t = blockIdx.x * 4 + wid; t < ntiles; t += gridDim.x * 4) {
        const int row = t * 32 + li;
        const float4* __restrict__ prow =
            reinterpret_cast<const float4*>(pool + (long long)min(row, N - 1) * d) + lh;     // clamped: always valid
        f32x16s acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // loads are unconditional from clamped indices (a `cond ? *p : 0` select would become a flat load of a
        // scratch zero); out-of-range steps are skipped by the wave-uniform guards around the MFMAs
        float4 b0[SCAN_U], b1[SCAN_U];
#pragma unroll
        for (int u = 0; u < SCAN_U; ++u) b0[u] = prow[2 * min(u, nsteps - 1)];
        for (int s0 = 0; s0 < nsteps; s0 += 2 * SCAN_U) {
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) b1[u] = prow[2 * min(s0 + SCAN_U + u, nsteps - 1)];
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) {
                if (s0 + u < nsteps) {
                    const float* qa = Qs + (8 * (s0 + u) + 4 * lh) * SCAN_LDQ + li;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[0], b0[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[SCAN_LDQ], b0[u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * SCAN_LDQ], b0[u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[3 * SCAN_LDQ], b0[u].w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) b0[u] = prow[2 * min(s0 + 2 * SCAN_U + u, nsteps - 1)];
#pragma unroll
            for (int u = 0; u < SCAN_U; ++u) {
                if (s0 + SCAN_U + u < nsteps) {
                    const float* qa = Qs + (8 * (s0 + SCAN_U + u) + 4 * lh) * SCAN_LDQ + li;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[0], b1[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[SCAN_LDQ], b1[u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * SCAN_LDQ], b1[u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[3 * SCAN_LDQ], b1[u].w, acc, 0, 0, 0);
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
    const size_t lds = (size_t)d * SCAN_LDQ * sizeof(float);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute((const void*)pool_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    140 * 1024) != hipSuccess) {
                set_error("pool_scan: cannot raise dynamic LDS limit");
                return R4D_ERR_HIP;
            }
            raised = true;
        }
    }
    const int ntiles = cdiv(N, 32);
    const int blocks_per_cu = lds > 80 * 1024 ? 1 : 2;
    const int gx = max(1, min(cdiv(ntiles, 4), 256 * blocks_per_cu));
    // algorithmic bytes (SURVEY 8d B_score): pool read once + queries + score rows out
    ProfScope prof(PK_POOL_SCAN, 4.0 * N * d * cdiv(Q, 32) + 4.0 * Q * d + 4.0 * (double)Q * N, s);
    hipLaunchKernelGGL(pool_scan_kernel, dim3(gx, cdiv(Q, 32)), dim3(256), lds, s, qhat, pool, Q, N, d, scores);
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

constexpr int TOPK_E = 16;                 // candidates per thread
constexpr int TOPK_SEG = 256 * TOPK_E;     // candidates per workgroup

// k rounds of workgroup-wide arg-best with removal over the register-resident candidates c[0..E).
// Every thread keeps its current local best; only the round's winner rescans.  Results to out[0..k).
template <typename T>
__device__ void block_topk_rounds(Cand<T> (&c)[TOPK_E], int k, T* out_v, long long* out_i) {
    __shared__ Cand<T> wbest[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const Cand<T> none = {neg_inf<T>(), 0x7fffffffffffffffLL};
    auto local_best = [&](int& slot) {
        Cand<T> b = none;
        slot = -1;
#pragma unroll
        for (int e = 0; e < TOPK_E; ++e)
            if (better(c[e], b)) { b = c[e]; slot = e; }
        return b;
    };
    int slot;
    Cand<T> mine = local_best(slot);
    for (int r = 0; r < k; ++r) {
        Cand<T> w = mine;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const Cand<T> t = shfl_xor_cand(w, o);
            if (better(t, w)) w = t;
        }
        if (lane == 0) wbest[wid] = w;
        __syncthreads();
        Cand<T> g = wbest[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (better(wbest[q], g)) g = wbest[q];
        __syncthreads();
        if (threadIdx.x == 0) { out_v[r] = g.v; out_i[r] = g.i; }
        if (slot >= 0 && mine.i == g.i && mine.v == g.v) {          // indices are unique: exactly one owner
#pragma unroll
            for (int e = 0; e < TOPK_E; ++e)
                if (e == slot) c[e] = none;
            mine = local_best(slot);
        }
    }
}

// One workgroup reduces one segment of TOPK_SEG candidates of one row to its k best.
//   vals [rows, ld]; idx_in (nullable) parallel global indices, else index = column + index_offset.
//   out_v/out_i [rows, nseg*k]
template <typename T>
__global__ __launch_bounds__(256) void topk_seg_kernel(const T* __restrict__ vals, const long long* __restrict__ idx_in,
                                                       int n, long long ld, int k, long long index_offset,
                                                       T* __restrict__ out_v, long long* __restrict__ out_i) {
    const int seg = blockIdx.x, row = blockIdx.y, nseg = gridDim.x;
    const T* v = vals + (long long)row * ld;
    const long long* ii = idx_in ? idx_in + (long long)row * ld : nullptr;
    Cand<T> c[TOPK_E];
#pragma unroll
    for (int e = 0; e < TOPK_E; ++e) {
        const int col = seg * TOPK_SEG + e * 256 + threadIdx.x;
        if (col < n) {
            T x = v[col];
            c[e].v = (x != x) ? neg_inf<T>() : x;
            c[e].i = ii ? ii[col] : (long long)col + index_offset;
        } else {
            c[e].v = neg_inf<T>();
            c[e].i = 0x7fffffffffffffffLL;
        }
    }
    block_topk_rounds<T>(c, k, out_v + ((long long)row * nseg + seg) * k, out_i + ((long long)row * nseg + seg) * k);
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

// rows x n values -> rows x k best (value, index).  Multi-level: segments of 4096 candidates.
template <typename T>
static int topk_rows(const T* vals, int rows, int n, long long ld, int k, long long index_offset, T* out_v,
                     long long* out_i, void* ws, size_t ws_bytes, hipStream_t s) {
    R4D_REQUIRE(k >= 1 && k <= 64 && k <= n, "topk: k=%d must be in [1, min(64, n=%d)]", k, n);
    R4D_REQUIRE(rows >= 1 && rows <= 65535, "topk: %d rows outside [1, 65535] per call (grid.y limit)", rows);
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
        ProfScope prof(PK_TOPK, (double)rows * cur * (sizeof(T) + (ci ? 8 : 0)), s);
        hipLaunchKernelGGL((topk_seg_kernel<T>), dim3(nseg, rows), dim3(256), 0, s, cv, ci, (int)cur, cld, k,
                           index_offset, ov, oi);
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
// vals/idx [G,Q,k] -> [Q,k]; one workgroup per query row; G*k <= 4096.
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ vals, const long long* __restrict__ idx,
                                                         int G, int Q, int k, float* __restrict__ out_v,
                                                         long long* __restrict__ out_i) {
    const int q = blockIdx.x;
    Cand<float> c[TOPK_E];
#pragma unroll
    for (int e = 0; e < TOPK_E; ++e) {
        const int t = e * 256 + threadIdx.x;
        if (t < G * k) {
            const int g = t / k, j = t % k;
            const long long off = ((long long)g * Q + q) * k + j;
            c[e].v = vals[off];
            c[e].i = idx[off];
        } else {
            c[e].v = -INFINITY;
            c[e].i = 0x7fffffffffffffffLL;
        }
    }
    block_topk_rounds<float>(c, k, out_v + (long long)q * k, out_i + (long long)q * k);
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
    if (Q <= 64 && d % 8 == 0 && d <= 1024) {
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

int r4d_merge_topk_f32(const float* vals_d, const int64_t* idx_d, int32_t G, int32_t Q, int32_t k, float* out_val_d,
                       int64_t* out_idx_d, void* stream) {
    R4D_REQUIRE(vals_d && idx_d && out_val_d && out_idx_d, "merge_topk: null pointer");
    R4D_REQUIRE(G >= 1 && Q >= 1 && k >= 1 && k <= 64 && (long long)G * k <= TOPK_SEG,
                "merge_topk: G=%d k=%d out of range (G*k <= %d)", G, k, TOPK_SEG);
    ProfScope prof(PK_MERGE_TOPK, 12.0 * G * Q * k, (hipStream_t)stream);
    hipLaunchKernelGGL(merge_topk_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, vals_d, (const long long*)idx_d,
                       G, Q, k, out_val_d, (long long*)out_idx_d);
    R4D_CHECK_LAUNCH("merge_topk");
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
