// Skinny fp32 GEMM for the decode step: y[M,N] = epilogue(x[M,K] . wT[N,K]^T + bias), M <= 32 rows.
//
// With a handful of rows a Conv1D is a WEIGHT STREAM (K*N*4 bytes read once, ~0 reuse): the tiled kernels run it as
// N/64 lone tiles with K-long serial loops (12-48 busy CUs, 33 us per projection at d = 768 -- 0.8 ms of a 1.1 ms decode
// step).  This kernel is the pool-scan design (score.hip) pointed at a weight matrix: the rows of x take the place of
// the queries (row-major in LDS, one ds_read_b128 per four MFMAs), every wavefront owns a 32-row tile of wT and streams it
// with 16-byte loads, 16 loads in flight, exact-f32 MFMA 32x32x2 -- and K is SPLIT over gridDim.y slices of 256 so that
// a 768 x 768 projection still spreads over 72 wavefronts.  Slice partials go to a caller-provided scratch and are
// added in slice order by the epilogue kernel (bias / gelu_new / residual): deterministic, no atomics.
//
// Measured on the decode step (L6 d768, 32 sequences): that first kernel took 14.8 us per projection and its epilogue
// 6.8 us -- neither launch-bound (the floor is 2.8 us per launch, 1.8 us per graph node: tools/launch_floor.hip) nor
// bandwidth-bound (7 MB in 15 us), but a LATENCY CHAIN: stage x, barrier, one round trip for the weights, then 128
// dependent 32x32x2 MFMAs (64 cycles each = 3.4 us) in a single accumulator.  gemm_skinny8_kernel shortens the chain and
// drops launches: EIGHT wavefronts share a 32-row tile of wT and split its k range (slices of 512 or 768: 16 or 24 MFMAs
// per wave), every global load of the kernel is issued before anything waits (x pieces coalesced and re-read as
// fragments from a per-wave LDS region), the eight partial tiles meet in LDS, and when one slice covers K the bias / gelu_new / residual
// epilogue and -- for the two projections that read a LayerNorm -- the LayerNorm itself (row statistics recomputed
// per workgroup from the L2-resident x, applied to the fragments) run in the same launch.  K = 4d (mlp c_proj) still
// splits over gridDim.y and keeps the epilogue kernel.
#include <stdlib.h>
#include "common.h"

namespace r4d {

typedef float f32x16g __attribute__((ext_vector_type(16)));
constexpr int SK_KC = 256;                              // k-slice per workgroup
constexpr int SK_LDX = SK_KC + 4;                       // Xs row stride (floats): conflict-free b128 reads

__device__ __forceinline__ float gelu_new_sk(float x) {   // same form as gemm_f32_kc.hip
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

// partial[ks][m][n] = sum over k in slice ks of x[m,k] * wT[n,k]        (m < 32 padded with zero rows)
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float* __restrict__ x, const float* __restrict__ wT, int M,
                                                          int N, int K, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float Xs[32 * SK_LDX];     // [32 rows][SK_KC + 4], k contiguous
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = blockIdx.y * SK_KC;
    for (int m = wid; m < 32; m += 4) {                // one x row per wave and trip, 16 bytes per lane
        const bool ok = m < M;
        float4 v = reinterpret_cast<const float4*>(x + (long long)min(m, M - 1) * K + k0)[lane];      // SK_KC / 4 == 64
        if (!ok) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
        *reinterpret_cast<float4*>(Xs + m * SK_LDX + 4 * lane) = v;
    }
    __syncthreads();
    const int tile = blockIdx.x * 4 + wid;
    const int ntiles = (N + 31) / 32;
    if (tile >= ntiles) return;
    const int row = tile * 32 + li;
    // k order as in pool_scan_kernel: lane half h of row j owns the 64-byte halves [32g + 16h, +16) of each 128-byte line
    const float4* __restrict__ wrow =
        reinterpret_cast<const float4*>(wT + (long long)min(row, N - 1) * K + k0) + 4 * lh;     // clamped: always valid
    f32x16g acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NG = SK_KC / 32;                     // 8 groups of 32 k
    float4 b[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) b[g][u] = wrow[8 * g + u];          // the whole slice of this row: 32 loads in flight
    const float* xfrag = Xs + li * SK_LDX + 16 * lh;   // + 32g + 4u: ONE ds_read_b128 per four MFMAs
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 xa = *reinterpret_cast<const float4*>(xfrag + 32 * g + 4 * u);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.x, b[g][u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.y, b[g][u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.z, b[g][u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa.w, b[g][u].w, acc, 0, 0, 0);
        }
    if (row < N) {
        float* dst = partial + (long long)blockIdx.y * 32 * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            dst[(long long)m * N + row] = acc[r];      // rows m >= M are zero: written, never read
        }
    }
}

// y[m,n] = epilogue(sum_ks partial[ks][m][n] + bias[n] (+ resid[m,n])), slices added in order
__global__ __launch_bounds__(256) void gemm_skinny_epilogue_kernel(const float* __restrict__ partial, int KS, int M, int N,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ resid, int epilogue,
                                                                   float* __restrict__ y) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float v = 0.f;
    for (int ks = 0; ks < KS; ++ks) v += partial[((long long)ks * 32 + m) * N + n];
    v += bias ? bias[n] : 0.f;
    if (epilogue == EPI_GELU) v = gelu_new_sk(v);
    else if (epilogue == EPI_RESIDUAL) v += resid[idx];
    y[idx] = v;
}

// ---------------------------------------------------------------------------------------------------- 8-wave kernel
#ifndef SK_DBG
#define SK_DBG 0              // tuning aid (tools/kc_ablate.sh gemm_skinny.hip SK_DBG n): 1 no MFMAs, 2 no weight loads, 4 no x loads, 8 no LDS reduction
#endif
constexpr int S8_NW = 8;
constexpr int S8_LDR = 33;                             // partial-tile row stride in LDS (floats)

__device__ __forceinline__ float wave_sum_sk(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (ceil(N/32), KS), 512 threads.  Slice length SL = 8 * 32 * NG.  KS == 1: y = epilogue(x' . wT^T + bias) with
// x' = LN ? LayerNorm(x; ln_w, ln_b, eps) : x.   KS > 1: partial[ks][m][n] (m < 32) for gemm_skinny_epilogue_kernel.
template <int NG, bool LN>
__global__ __launch_bounds__(64 * S8_NW) void gemm_skinny8_kernel(const float* __restrict__ x, const float* __restrict__ wT,
                                                                  int M, int N, int K, const float* __restrict__ bias,
                                                                  const float* __restrict__ resid, int epilogue,
                                                                  float* __restrict__ y, float* __restrict__ partial,
                                                                  const float* __restrict__ ln_w,
                                                                  const float* __restrict__ ln_b, float eps) {
    constexpr int KW = 32 * NG, SL = S8_NW * KW;
    // x goes through LDS: a fragment load straight from memory touches 32 cache lines for 1 KB, so each wave loads ITS
    // k-piece of the 32 rows coalesced (PPR lanes x 16 bytes per row), parks it in a private padded region and reads the
    // fragments back with conflict-free ds_read_b128 (measured: 385 -> 374 us per decode step; the ablation's 2.7 us per
    // launch for the x loads was mostly their LATENCY in the chain, which staging does not remove)
    constexpr int LDX = KW + 4, PPR = KW / 4, RPP = 64 / PPR, NPASS = 32 / RPP;
    __shared__ __attribute__((aligned(16))) float xs_all[S8_NW * 32 * LDX];
    __shared__ float red[S8_NW * 32 * S8_LDR];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int k0 = blockIdx.y * SL + wid * KW;
    const int row = blockIdx.x * 32 + li;
    // k order as in pool_scan_kernel: lane half h of a row owns the 64-byte halves [32g + 16h, +16) of each 128-byte line
    const float4* __restrict__ wrow = reinterpret_cast<const float4*>(wT + (long long)min(row, N - 1) * K + k0) + 4 * lh;
    float* xs = xs_all + wid * 32 * LDX;
    const int pr = lane / PPR, pc = lane % PPR;        // NG = 3: 24 lanes per row, lanes 48..63 idle in the staging
    // EVERY global load of the kernel is issued here, before anything waits: weights (the long HBM latency), x fragments,
    // LayerNorm gain / shift fragments, and the epilogue's bias / residual values -- one round trip instead of four
    float4 b[NG][4], a[NG][4], gw[LN ? NG : 1][4], gb[LN ? NG : 1][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) b[g][u] = (SK_DBG & 2) ? make_float4(1.f, 2.f, 3.f, (float)tid) : wrow[8 * g + u];
    float4 xv[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {                  // rows >= M repeat row M-1: computed, never stored
        const int r = min(min(j * RPP + pr, 31), M - 1);
        xv[j] = (SK_DBG & 4) ? make_float4(1.f, 2.f, 3.f, (float)tid)
                             : reinterpret_cast<const float4*>(x + (long long)r * K + k0)[pc];
    }
    if (LN) {
        const float4* __restrict__ gwp = reinterpret_cast<const float4*>(ln_w + k0) + 4 * lh;
        const float4* __restrict__ gbp = reinterpret_cast<const float4*>(ln_b + k0) + 4 * lh;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) { gw[g][u] = gwp[8 * g + u]; gb[g][u] = gbp[8 * g + u]; }
    }
    const int n = blockIdx.x * 32 + li;                // epilogue: thread -> (m, n), 32 consecutive n per row, 16 rows per pass
    const int nc = min(n, N - 1), m0 = min(tid >> 5, M - 1), m1 = min((tid >> 5) + 16, M - 1);
    const bool fused = partial == nullptr;
    const float bias_n = (fused && bias) ? bias[nc] : 0.f;
    const float res0 = (fused && epilogue == EPI_RESIDUAL) ? resid[(long long)m0 * N + nc] : 0.f;
    const float res1 = (fused && epilogue == EPI_RESIDUAL) ? resid[(long long)m1 * N + nc] : 0.f;
    if (pr < RPP)
#pragma unroll
        for (int j = 0; j < NPASS; ++j) *reinterpret_cast<float4*>(xs + (j * RPP + pr) * LDX + 4 * pc) = xv[j];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) a[g][u] = *reinterpret_cast<const float4*>(xs + li * LDX + 32 * g + 16 * lh + 4 * u);
    if (LN) {
        // K == SL: the eight waves hold the eight k-pieces of the SAME 32 rows (row li per lane) -- row statistics from
        // the fragments themselves: piece sums meet in LDS, mean first, then the centred squares (two-pass like ln4_kernel)
        float s_ = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) s_ += (a[g][u].x + a[g][u].y) + (a[g][u].z + a[g][u].w);
        s_ += __shfl_xor(s_, 32);
        if (lh == 0) red[wid * 32 + li] = s_;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < S8_NW; ++w) mean += red[w * 32 + li];
        mean /= (float)K;
        float q = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[g][u].x -= mean; a[g][u].y -= mean; a[g][u].z -= mean; a[g][u].w -= mean;
                q += (a[g][u].x * a[g][u].x + a[g][u].y * a[g][u].y) + (a[g][u].z * a[g][u].z + a[g][u].w * a[g][u].w);
            }
        q += __shfl_xor(q, 32);
        if (lh == 0) red[256 + wid * 32 + li] = q;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < S8_NW; ++w) var += red[256 + w * 32 + li];
        const float rstd = rsqrtf(var / (float)K + eps);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[g][u].x = a[g][u].x * rstd * gw[g][u].x + gb[g][u].x;
                a[g][u].y = a[g][u].y * rstd * gw[g][u].y + gb[g][u].y;
                a[g][u].z = a[g][u].z * rstd * gw[g][u].z + gb[g][u].z;
                a[g][u].w = a[g][u].w * rstd * gw[g][u].w + gb[g][u].w;
            }
        __syncthreads();                               // red is reused for the partial tiles
    }
    f32x16g acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SK_DBG & 1) { acc[(4 * g + u) & 15] += a[g][u].x * b[g][u].x + a[g][u].y * b[g][u].y + a[g][u].z * b[g][u].z + a[g][u].w * b[g][u].w; continue; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u].x, b[g][u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u].y, b[g][u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u].z, b[g][u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][u].w, b[g][u].w, acc, 0, 0, 0);
        }
    if (SK_DBG & 8) {                                  // every wave stores its own tile: no LDS, no barrier
        if (row < N && wid == 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[(long long)min((r & 3) + 8 * (r >> 2) + 4 * lh, M - 1) * N + row] = acc[r] + bias_n + res0;
        return;
    }
    float* mine = red + wid * 32 * S8_LDR;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lh) * S8_LDR + li] = acc[r];
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int m = (tid >> 5) + 16 * pass;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < S8_NW; ++w) v += red[(w * 32 + m) * S8_LDR + li];       // wave order: deterministic
        if (n >= N) continue;
        if (!fused) {
            partial[((long long)blockIdx.y * 32 + m) * N + n] = v;                  // rows m >= M: written, never read
        } else if (m < M) {
            v += bias_n;
            if (epilogue == EPI_GELU) v = gelu_new_sk(v);
            else if (epilogue == EPI_RESIDUAL) v += pass ? res1 : res0;
            y[(long long)m * N + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- 16-column kernel
// Same plan as gemm_skinny8_kernel with HALF the columns per workgroup (16 rows of wT, v_mfma_f32_16x16x4_f32 on two 16-row
// halves of x): a projection is spread over twice as many CUs (N = 768: 48 instead of 24; N = 2304: 144; N = 3072: 192), so
// both per-CU costs of the chain halve -- the exact-f32 MFMA work of a tile (8 waves x 48 MFMAs x 64 clocks = 2.6 us per CU
// with 32 columns) and the tile's weight bytes through the CU's one L1.  Two more seams are gone:
//  * the LayerNorm of c_attn / c_fc no longer sits in front of the MFMAs.  LN(x) . W = rstd * (sum_k x_k g_k W_kn - mean * c1_n)
//    + c2_n with c1_n = sum_k g_k W_kn and c2_n = sum_k b_k W_kn: the MFMAs start on x_k * g_k as soon as the operands have
//    landed, while mean / variance (two-pass, from the same fragments) and c1 / c2 (two FMAs per weight element a lane holds
//    anyway) are formed beside them and meet the tiles in the LDS reduction.
//  * K = 4d (mlp c_proj) splits over gridDim.y as before, but the slices are combined by the LAST workgroup of a column
//    tile to arrive (write-through partials + one relaxed ticket, Guideline 16; slices added in slice order whoever arrives
//    last: deterministic) -- no epilogue launch.
constexpr int S16_LDR = 17;                            // partial-tile row stride in LDS (floats)
constexpr int S16_MAX_TILES = 256;                     // ticket counters at the head of the scratch (KS > 1): N <= 4096

__device__ __forceinline__ void publish_sk(float* p, float v) {      // write-through (sc1) store, see topk.hip
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float consume_sk(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

typedef float f32x4g __attribute__((ext_vector_type(4)));

// grid (ceil(N/16), KS), 512 threads.  KS == 1: y = epilogue(x' . wT^T + bias), x' = LN ? LayerNorm(x) : x.
// KS > 1 (never with LN): partial[ks][32][N] + counters[tile]; the last arriver writes y.
// LN = 2: the LayerNorm is PRE-FOLDED into the operands (r4d_fold_layernorm_f32, once per checkpoint): wT holds g_k W_nk and
// ln_w points at [2][N] = c1 (sum_k g_k W_nk), c2 (sum_k beta_k W_nk) -- no gain / shift loads, no gain multiply in front of the
// MFMAs, no c1 / c2 arithmetic; the row statistics still come from the x fragments
template <int NG, int LN>
__global__ __launch_bounds__(64 * S8_NW) void gemm_skinny16_kernel(const float* __restrict__ x, const float* __restrict__ wT,
                                                                   int M, int N, int K, const float* __restrict__ bias,
                                                                   const float* __restrict__ resid, int epilogue,
                                                                   float* __restrict__ y, float* partial, unsigned* counters,
                                                                   const float* __restrict__ ln_w,
                                                                   const float* __restrict__ ln_b, float eps) {
    constexpr int KW = 32 * NG, SL = S8_NW * KW, NJ = KW / 16;
    constexpr int LDX = KW + 4, PPR = KW / 4, RPP = 64 / PPR, NPASS = 32 / RPP;
    __shared__ __attribute__((aligned(16))) float xs_all[S8_NW * 32 * LDX];
    __shared__ float red_t[S8_NW * 32 * S16_LDR];
    __shared__ float red_s[LN ? 2 * S8_NW * 32 : 1];   // [0]: piece sums, [1]: centred squares, per (wave, row)
    __shared__ float red_c[LN == 1 ? 2 * S8_NW * 16 : 1];   // c1 / c2 pieces per (wave, column)
    __shared__ unsigned s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = lane & 15, q = lane >> 4;
    const int KS = gridDim.y;
    const int k0 = blockIdx.y * SL + wid * KW;
    const int col = blockIdx.x * 16 + n16;
    const float* __restrict__ wrow = wT + (long long)min(col, N - 1) * K + k0 + 4 * q;
    float* xs = xs_all + wid * 32 * LDX;
    const int pr = lane / PPR, pc = lane % PPR;
    // every global load of the kernel up front: weights (HBM latency), x pieces (L2), LayerNorm gain / shift, bias, residual
    float4 b[NJ], gw[LN == 1 ? NJ : 1], gb[LN == 1 ? NJ : 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float4*>(wrow + 16 * j);
    float4 xv[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {                  // rows >= M repeat row M-1: computed, never stored
        const int r = min(min(j * RPP + pr, 31), M - 1);
        xv[j] = reinterpret_cast<const float4*>(x + (long long)r * K + k0)[pc];
    }
    if (LN == 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            gw[j] = *reinterpret_cast<const float4*>(ln_w + k0 + 16 * j + 4 * q);
            gb[j] = *reinterpret_cast<const float4*>(ln_b + k0 + 16 * j + 4 * q);
        }
    }
    const int em = tid >> 4, en = blockIdx.x * 16 + (tid & 15);      // epilogue: one output per thread, 16 columns per row
    const int enc = min(en, N - 1), emc = min(em, M - 1);
    const bool fused = KS == 1;
    const float bias_n = bias ? bias[enc] : 0.f;
    const float fc1 = LN == 2 ? ln_w[enc] : 0.f, fc2 = LN == 2 ? ln_w[N + enc] : 0.f;      // folded column constants
    const float res = (fused && epilogue == EPI_RESIDUAL) ? resid[(long long)emc * N + enc] : 0.f;
    if (pr < RPP)
#pragma unroll
        for (int j = 0; j < NPASS; ++j) *reinterpret_cast<float4*>(xs + (j * RPP + pr) * LDX + 4 * pc) = xv[j];
    __syncthreads();
    float4 a[2][NJ];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) a[t][j] = *reinterpret_cast<const float4*>(xs + (16 * t + n16) * LDX + 16 * j + 4 * q);
    f32x4g acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = f32x4g{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 g = LN == 1 ? gw[j] : make_float4(1.f, 1.f, 1.f, 1.f);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(LN == 1 ? a[t][j].x * g.x : a[t][j].x, b[j].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(LN == 1 ? a[t][j].y * g.y : a[t][j].y, b[j].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(LN == 1 ? a[t][j].z * g.z : a[t][j].z, b[j].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(LN == 1 ? a[t][j].w * g.w : a[t][j].w, b[j].w, acc[t], 0, 0, 0);
        }
    if (LN) {
        // row statistics from the fragments (lane (n16, q) holds row 16t + n16, k = 16j + 4q + c of this wave's piece) and the
        // column constants c1 / c2 from the weights it holds: piece sums meet in LDS (two-pass variance like ln4_kernel)
        float s_[2], c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s_[t] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) s_[t] += (a[t][j].x + a[t][j].y) + (a[t][j].z + a[t][j].w);
            s_[t] += __shfl_xor(s_[t], 16);
            s_[t] += __shfl_xor(s_[t], 32);
        }
        if (LN == 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                c1 += (gw[j].x * b[j].x + gw[j].y * b[j].y) + (gw[j].z * b[j].z + gw[j].w * b[j].w);
                c2 += (gb[j].x * b[j].x + gb[j].y * b[j].y) + (gb[j].z * b[j].z + gb[j].w * b[j].w);
            }
            c1 += __shfl_xor(c1, 16); c1 += __shfl_xor(c1, 32);
            c2 += __shfl_xor(c2, 16); c2 += __shfl_xor(c2, 32);
        }
        if (q == 0) {
            red_s[wid * 32 + n16] = s_[0]; red_s[wid * 32 + 16 + n16] = s_[1];
            if (LN == 1) { red_c[wid * 16 + n16] = c1; red_c[S8_NW * 16 + wid * 16 + n16] = c2; }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < S8_NW; ++w) mean += red_s[w * 32 + 16 * t + n16];
            mean /= (float)K;
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float dx = a[t][j].x - mean, dy = a[t][j].y - mean, dz = a[t][j].z - mean, dw = a[t][j].w - mean;
                qq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
            if (q == 0) red_s[S8_NW * 32 + wid * 32 + 16 * t + n16] = qq;
        }
    }
    float* mine = red_t + wid * 32 * S16_LDR;          // C layout of 16x16x4: lane (n16, q) holds rows 4q + r, column n16
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(16 * t + 4 * q + r) * S16_LDR + n16] = acc[t][r];
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < S8_NW; ++w) v += red_t[(w * 32 + em) * S16_LDR + (tid & 15)];      // wave order: deterministic
    if (LN) {
        float mean = 0.f, var = 0.f, c1 = fc1, c2 = fc2;
#pragma unroll
        for (int w = 0; w < S8_NW; ++w) {
            mean += red_s[w * 32 + em]; var += red_s[S8_NW * 32 + w * 32 + em];
            if (LN == 1) { c1 += red_c[w * 16 + (tid & 15)]; c2 += red_c[S8_NW * 16 + w * 16 + (tid & 15)]; }
        }
        mean /= (float)K;
        v = rsqrtf(var / (float)K + eps) * (v - mean * c1) + c2;
    }
    if (fused) {
        if (en < N && em < M) {
            v += bias_n;
            if (epilogue == EPI_GELU) v = gelu_new_sk(v);
            else if (epilogue == EPI_RESIDUAL) v += res;
            y[(long long)em * N + en] = v;
        }
        return;
    }
    // split-K: publish this slice's tile, take a ticket; the last workgroup of the column tile adds the slices in order
    if (en < N) publish_sk(partial + ((long long)blockIdx.y * 32 + em) * N + en, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        s_last = __hip_atomic_fetch_add(&counters[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)KS - 1u;
    __syncthreads();
    if (!s_last) return;
    // same hand-off as topk.hip (write-through sc1 partials, drained before the ticket behind a workgroup barrier, sc1 loads in
    // the last arriver after a barrier it joined): compiler barrier so that no partial load is hoisted above the ticket;
    // build.py checks the sc1 lowering
    asm volatile("" ::: "memory");
    if (tid == 0) __hip_atomic_store(&counters[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (en < N && em < M) {
        float o = 0.f;
        for (int ks = 0; ks < KS; ++ks) o += consume_sk(partial + ((long long)ks * 32 + em) * N + en);
        o += bias_n;
        if (epilogue == EPI_GELU) o = gelu_new_sk(o);
        else if (epilogue == EPI_RESIDUAL) o += resid[(long long)em * N + en];
        y[(long long)em * N + en] = o;
    }
}


// W'[n][k] = g[k] W[n][k];  lnc[n] = sum_k g[k] W[n][k];  lnc[N + n] = sum_k beta[k] W[n][k]   (one wavefront per weight row)
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ wT, const float* __restrict__ g,
                                                             const float* __restrict__ beta, int N, int K, float* __restrict__ wTg,
                                                             float* __restrict__ lnc) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float c1 = 0.f, c2 = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = wT[(long long)n * K + k], gw = g[k] * w;
        wTg[(long long)n * K + k] = gw;
        c1 += gw; c2 += beta[k] * w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
    if (lane == 0) { lnc[n] = c1; lnc[N + n] = c2; }
}
int launch_fold_layernorm(const float* wT, const float* g, const float* beta, int N, int K, float* wTg, float* lnc, hipStream_t s) {
    R4D_REQUIRE(wT && g && beta && wTg && lnc && N >= 1 && K >= 1, "fold_layernorm: bad arguments");
    hipLaunchKernelGGL(fold_layernorm_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, wT, g, beta, N, K, wTg, lnc);
    R4D_CHECK_LAUNCH("fold_layernorm");
    return R4D_OK;
}

static int skinny8_ng(int K) { return K % 768 == 0 ? 3 : (K % 512 == 0 ? 2 : 0); }       // 0: the 4-wave kernel

bool gemm_skinny_fuses_ln(int M, int K, int N) { return gemm_skinny_supported(M, K, N) && (K == 768 || K == 512); }

size_t gemm_skinny_scratch_floats(int K, int N) { return (size_t)cdiv(K, SK_KC) * 32 * N + S16_MAX_TILES; }   // + ticket counters

bool gemm_skinny_supported(int M, int K, int N) { return M >= 1 && M <= 32 && K % SK_KC == 0 && K >= SK_KC && N >= 1; }

void* gemm_skinny_counters(float* scratch, size_t* bytes) { *bytes = S16_MAX_TILES * sizeof(unsigned); return scratch; }

// `ln_fold` (nullable): wT is the PRE-FOLDED weight g_k W_nk and ln_fold its [2][N] column constants (launch_fold_layernorm);
// ln_w / ln_b are then unused
int launch_gemm_skinny(const float* x, const float* wT, const float* bias, const float* resid, int M, int K, int N,
                       int epilogue, float* y, float* scratch, hipStream_t s, const float* ln_w, const float* ln_b,
                       float ln_eps, bool counters_zeroed, const float* ln_fold) {
    R4D_REQUIRE(!ln_fold || gemm_skinny_fuses_ln(M, K, N), "skinny gemm: no folded LayerNorm for K=%d", K);
    R4D_REQUIRE(gemm_skinny_supported(M, K, N) && scratch, "skinny gemm: unsupported shape M=%d K=%d N=%d", M, K, N);
    R4D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wT % 16) == 0, "skinny gemm: operands must be 16-byte aligned");
    R4D_REQUIRE(!ln_w || (ln_b && gemm_skinny_fuses_ln(M, K, N)), "skinny gemm: no fused LayerNorm for K=%d", K);
    const int ng = skinny8_ng(K);
    int KS;
    static int use16 = -1;
    if (use16 < 0) { const char* e = getenv("R4D_SKINNY16"); use16 = e ? atoi(e) : 1; }      // tuning aid: 0 -> the 32-column kernels
    if (ng && use16 && (K / (S8_NW * 32 * ng) == 1 || cdiv(N, 16) <= S16_MAX_TILES)) {      // 8 waves per 16-row tile of wT
        const int SL = S8_NW * 32 * ng;
        KS = K / SL;
        R4D_REQUIRE(KS == 1 || (!ln_w && !ln_fold), "skinny gemm: no fused LayerNorm across k slices");
        unsigned* counters = reinterpret_cast<unsigned*>(scratch);
        float* partial = KS > 1 ? scratch + S16_MAX_TILES : nullptr;
        if (KS > 1 && !counters_zeroed) R4D_HIP(hipMemsetAsync(counters, 0, S16_MAX_TILES * sizeof(unsigned), s));
        ProfScope prof(PK_GEMM_SKINNY, 4.0 * K * (double)N + 4.0 * M * K + (KS > 1 ? 8.0 * KS * 32.0 * N : 4.0 * M * N), s);
        const dim3 grid(cdiv(N, 16), KS), block(64 * S8_NW);
#define SK16_(NG_, LN_)                                                                                                    \
    hipLaunchKernelGGL((gemm_skinny16_kernel<NG_, LN_>), grid, block, 0, s, x, wT, M, N, K, bias, resid, epilogue, y, partial, \
                       counters, (LN_) == 2 ? ln_fold : ln_w, ln_b, ln_eps)
        if (ln_fold) R4D_BRANCH(SK16_LN_FOLDED);
        else if (ng == 3) { if (ln_w) R4D_BRANCH(SK16_NG3_LN); else R4D_BRANCH(SK16_NG3); }
        else              { if (ln_w) R4D_BRANCH(SK16_NG2_LN); else R4D_BRANCH(SK16_NG2); }
        if (KS > 1) R4D_BRANCH(SK16_SPLITK);
        if (ng == 3) { if (ln_fold) SK16_(3, 2); else if (ln_w) SK16_(3, 1); else SK16_(3, 0); }
        else         { if (ln_fold) SK16_(2, 2); else if (ln_w) SK16_(2, 1); else SK16_(2, 0); }
#undef SK16_
        R4D_CHECK_LAUNCH("gemm_skinny16");
        return R4D_OK;
    }
    R4D_REQUIRE(!ln_fold, "skinny gemm: the folded LayerNorm needs the 16-column kernel");
    if (ng) {                                          // 8 waves per 32-row tile of wT
        const int SL = S8_NW * 32 * ng;
        KS = K / SL;
        float* partial = KS > 1 ? scratch + S16_MAX_TILES : nullptr;    // never the ticket counters at the head of the scratch
        ProfScope prof(PK_GEMM_SKINNY, 4.0 * K * (double)N + 4.0 * M * K + (KS > 1 ? 4.0 * KS * 32.0 * N : 4.0 * M * N), s);
        const dim3 grid(cdiv(N, 32), KS), block(64 * S8_NW);
#define SK8_(NG_, LN_)                                                                                                    \
    hipLaunchKernelGGL((gemm_skinny8_kernel<NG_, LN_>), grid, block, 0, s, x, wT, M, N, K, bias, resid, epilogue, y, partial, \
                       ln_w, ln_b, ln_eps)
        if (ng == 3) R4D_BRANCH(SK8_NG3); else R4D_BRANCH(SK8_NG2);
        if (ln_w) R4D_BRANCH(SK8_LN);
        if (KS > 1) R4D_BRANCH(SK8_SPLITK);
        if (ng == 3) { if (ln_w) SK8_(3, true); else SK8_(3, false); }
        else         { if (ln_w) SK8_(2, true); else SK8_(2, false); }
#undef SK8_
        R4D_CHECK_LAUNCH("gemm_skinny8");
        if (KS == 1) return R4D_OK;
    } else {
        KS = K / SK_KC;
        // algorithmic bytes: the weight matrix once (+ x per slice, + partials)
        ProfScope prof(PK_GEMM_SKINNY, 4.0 * K * (double)N + 4.0 * M * K + 4.0 * KS * 32.0 * N, s);
        R4D_BRANCH(SK_PLAIN);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(cdiv(cdiv(N, 32), 4), KS), dim3(256), 0, s, x, wT, M, N, K, scratch + S16_MAX_TILES);
        R4D_CHECK_LAUNCH("gemm_skinny");
    }
    {
        ProfScope prof(PK_GEMM_SKINNY_EPI, 4.0 * KS * M * (double)N + 8.0 * M * N, s);
        hipLaunchKernelGGL(gemm_skinny_epilogue_kernel, dim3((unsigned)cdiv((long long)M * N, 256)), dim3(256), 0, s, scratch + S16_MAX_TILES,
                           KS, M, N, bias, resid, epilogue, y);
        R4D_CHECK_LAUNCH("gemm_skinny_epilogue");
    }
    return R4D_OK;
}

}  // namespace r4d

namespace r4d { int dbgflag_sk() { return SK_DBG != 0; } }
