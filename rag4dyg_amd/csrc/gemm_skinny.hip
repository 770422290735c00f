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

static int skinny8_ng(int K) { return K % 768 == 0 ? 3 : (K % 512 == 0 ? 2 : 0); }       // 0: the 4-wave kernel

bool gemm_skinny_fuses_ln(int M, int K, int N) { return gemm_skinny_supported(M, K, N) && (K == 768 || K == 512); }

size_t gemm_skinny_scratch_floats(int K, int N) { return (size_t)cdiv(K, SK_KC) * 32 * N; }

bool gemm_skinny_supported(int M, int K, int N) { return M >= 1 && M <= 32 && K % SK_KC == 0 && K >= SK_KC && N >= 1; }

int launch_gemm_skinny(const float* x, const float* wT, const float* bias, const float* resid, int M, int K, int N,
                       int epilogue, float* y, float* scratch, hipStream_t s, const float* ln_w, const float* ln_b,
                       float ln_eps) {
    R4D_REQUIRE(gemm_skinny_supported(M, K, N) && scratch, "skinny gemm: unsupported shape M=%d K=%d N=%d", M, K, N);
    R4D_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wT % 16) == 0, "skinny gemm: operands must be 16-byte aligned");
    R4D_REQUIRE(!ln_w || (ln_b && gemm_skinny_fuses_ln(M, K, N)), "skinny gemm: no fused LayerNorm for K=%d", K);
    const int ng = skinny8_ng(K);
    int KS;
    if (ng) {                                          // 8 waves per 32-row tile of wT
        const int SL = S8_NW * 32 * ng;
        KS = K / SL;
        float* partial = KS > 1 ? scratch : nullptr;
        ProfScope prof(PK_GEMM_SKINNY, 4.0 * K * (double)N + 4.0 * M * K + (KS > 1 ? 4.0 * KS * 32.0 * N : 4.0 * M * N), s);
        const dim3 grid(cdiv(N, 32), KS), block(64 * S8_NW);
#define SK8_(NG_, LN_)                                                                                                    \
    hipLaunchKernelGGL((gemm_skinny8_kernel<NG_, LN_>), grid, block, 0, s, x, wT, M, N, K, bias, resid, epilogue, y, partial, \
                       ln_w, ln_b, ln_eps)
        if (ng == 3) { if (ln_w) SK8_(3, true); else SK8_(3, false); }
        else         { if (ln_w) SK8_(2, true); else SK8_(2, false); }
#undef SK8_
        R4D_CHECK_LAUNCH("gemm_skinny8");
        if (KS == 1) return R4D_OK;
    } else {
        KS = K / SK_KC;
        // algorithmic bytes: the weight matrix once (+ x per slice, + partials)
        ProfScope prof(PK_GEMM_SKINNY, 4.0 * K * (double)N + 4.0 * M * K + 4.0 * KS * 32.0 * N, s);
        hipLaunchKernelGGL(gemm_skinny_kernel, dim3(cdiv(cdiv(N, 32), 4), KS), dim3(256), 0, s, x, wT, M, N, K, scratch);
        R4D_CHECK_LAUNCH("gemm_skinny");
    }
    {
        ProfScope prof(PK_GEMM_SKINNY_EPI, 4.0 * KS * M * (double)N + 8.0 * M * N, s);
        hipLaunchKernelGGL(gemm_skinny_epilogue_kernel, dim3((unsigned)cdiv((long long)M * N, 256)), dim3(256), 0, s, scratch,
                           KS, M, N, bias, resid, epilogue, y);
        R4D_CHECK_LAUNCH("gemm_skinny_epilogue");
    }
    return R4D_OK;
}

}  // namespace r4d

namespace r4d { int dbgflag_sk() { return SK_DBG != 0; } }
