// fp32 MFMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] . B + bias), exact-f32 v_mfma_f32_32x32x2_f32.
//
// Serves every dense contraction of the encoder: Conv1D (modeling_utils.py:1267-1271) c_attn / c_proj /
// c_fc / mlp.c_proj with fused bias (+ gelu_new | + residual), the batched per-head Q.K^T / P.V of
// Attention._attn (modeling_gpt2.py:141,159) with causal tile skipping, the tied lm_head and the
// query-vs-pool cosine scan (B given as [N,K]).  1e-4-relative-fp32 parity rules out bf16 inputs, so this
// is the f32-in/f32-acc MFMA (157 TFLOP/s peak == the f32 vector peak, but one VGPR per operand and the
// VALU left free for the epilogue).
//
// Tiling: WGM x WGN wavefronts (2x2, or 4x2 for the 128x128 tile); block tile BM x BN x BK, wave tile
// (BM/WGM) x (BN/WGN) made of TM x TN 32x32 MFMA tiles.  LDS images are k-major ([k][m], [k][n]).  The rows (columns) of a wave tile
// are INTERLEAVED over its MFMA tiles -- MFMA tile t owns rows TM*i + t -- so one lane's TM (TN) operands
// for a k-step are adjacent in LDS and come from ONE conflict-free ds_read_b32/b64/b128, and the
// epilogue stores TN adjacent columns per lane (256-512 B contiguous per row segment).  A (and B when
// given as [N,K]) is transposed on the LDS write (2-way conflicts on ds_write_b32 are free).  LDS is
// double-buffered and the next k-tile's global loads are issued before the MFMA phase (one barrier per
// k-tile).  Tile shape is picked per launch by a wave-quantisation cost model (launch_gemm_f32).
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace r4d {

typedef float f32x16 __attribute__((ext_vector_type(16)));


__device__ __forceinline__ float gelu_new_f(float x) {
    // gelu_new(x) = 0.5x(1+tanh(u)), u = sqrt(2/pi)(x+0.044715x^3)  -- modeling_gpt2.py:25,206.
    // Algebraically 0.5(1+tanh(u)) = 1/(1+exp(-2u)) = 1/(1+exp2(x*(k0 + k1*x^2))) with k0 = -2 sqrt(2/pi) log2(e),
    // k1 = 0.044715 k0: mul, fma, mul, v_exp_f32, add, v_rcp_f32, mul -- every epilogue VALU instruction is taken from
    // the MFMA issue slots of the co-resident workgroup, the ocml tanhf form (~40) cost 15 % of a c_fc tile.
    // |error| < 3e-7 |x| (checked against the oracle at 1e-5 relative in tests/test_gpu_ops.py).
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

template <int N>
struct FragLoad;
template <>
struct FragLoad<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = p[0]; }
};
template <>
struct FragLoad<2> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct FragLoad<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// Shape / stride block passed by value; the POINTERS are separate kernel arguments so that the compiler
// knows they are global (pointers inside a by-value struct are treated as generic -> flat_load, whose
// out-of-order return couples every LDS wait to the global loads in flight).
struct GemmShape {
    int M, N, K, lda, ldb, ldc, ldr, b_rows, a_cols, nb1, epilogue, causal, k_total;
    long long sA0, sA1, sB0, sB1, sC0, sC1;
    float scale_div;
};

// AT: A is given as [K,M] (m-contiguous rows of k; weight gradients X^T . dY contract over the token rows of both
// operands): its LDS image is a straight copy like a row-major B.  With g.k_total > 0 the launch is a split-K one: batch z
// contracts rows [z*K, min((z+1)*K, k_total)) and writes its own partial C (reduced by splitk_reduce_kernel).
template <int BM, int BN, int BK, int WGM, int WGN, bool BT, bool AT = false>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 8) ? 4 : 1) void gemm_f32_kernel(const float* __restrict__ Ag, const float* __restrict__ Bg,
                                                            float* __restrict__ Cg, const float* __restrict__ biasg,
                                                            const float* __restrict__ residg, const GemmShape g) {
    constexpr int NTHREADS = 64 * WGM * WGN;                          // WGM x WGN wavefronts
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    // A^T-on-write image: BM + 2 (even: keeps ds_read_b64 aligned).  Straight-copy image (AT): unpadded like a row-major B
    // (the two k-rows of a fragment read belong to different 32-lane halves, which never conflict)
    constexpr int LDA = AT ? BM : BM + 2;
    constexpr int LDB = BT ? BN + 2 : BN;
    constexpr int KV = BK / 4;                                       // float4 per tile row
    constexpr int NLA = BM * KV / NTHREADS;                          // float4 loads per thread, A tile
    constexpr int NLB = BN * KV / NTHREADS;
    static_assert(NLA >= 1 && NLB >= 1 && (TM == 1 || TM == 2 || TM == 4) && (TN == 1 || TN == 2 || TN == 4), "tile");
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (bid % 8 = XCD group), each with
    // its own 4 MB L2.  Remap so that an XCD walks a CONTIGUOUS range of tiles (n fastest): the workgroups
    // that share one A row-panel then hit the same L2 instead of fetching it 8 times.  Bijective for any grid.
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
    // Inside an XCD's range the tiles are walked in groups of GROUP_M row-panels (m fastest inside a group): the
    // ~64 tiles resident on an XCD then span ~8 row-panels x ~8 column-panels, i.e. ~2 MB of A + ~2 MB of B in its
    // 4 MB L2, instead of 4 row-panels x every column-panel of B.
    constexpr int GROUP_M = 8;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int per_group = GROUP_M * tiles_n;
    const int grp = bid / per_group, first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (bid % per_group) % gsz, tile_n = (bid % per_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    if (g.causal == CAUSAL_QK && n0 > m0 + BM - 1) return;           // tile strictly above the diagonal

    const int z0 = blockIdx.z / g.nb1, z1 = blockIdx.z % g.nb1;
    const float* __restrict__ A = Ag + z0 * g.sA0 + z1 * g.sA1;
    const float* __restrict__ B = Bg + z0 * g.sB0 + z1 * g.sB1;
    float* __restrict__ C = Cg + z0 * g.sC0 + z1 * g.sC1;

    int kend = g.K;
    if (g.causal == CAUSAL_PV) kend = min(g.K, m0 + BM);             // keys beyond the tile's last row are masked
    const int krows = g.k_total > 0 ? min(g.K, g.k_total - (int)blockIdx.z * g.K) : g.b_rows;   // valid rows of a [K,*] operand
    if (g.k_total > 0) kend = krows;
    const int nkt = (kend + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, lh = lane >> 5;

    // per-thread staging coordinates (loop-invariant).  Out-of-range rows/columns are CLAMPED to a valid
    // address and the loaded value is zeroed afterwards: a `cond ? *p : 0` select makes hipcc pick between
    // the global pointer and a zero constant in scratch, i.e. flat_load + vmcnt(0) right after the load.
    const float* a_src[NLA]; bool a_ok[NLA]; int a_dst[NLA], a_k[NLA];
#pragma unroll
    for (int r = 0; r < NLA; ++r) {
        const int idx = tid + r * NTHREADS;
        if (AT) {
            const int krow = idx / (BM / 4), c = idx % (BM / 4);
            a_ok[r] = (m0 + 4 * c) < g.M;
            a_k[r] = krow;
            a_src[r] = A + (a_ok[r] ? m0 + 4 * c : 0);
            a_dst[r] = krow * LDA + 4 * c;
        } else {
            const int row = idx / KV, c4 = idx % KV;
            a_ok[r] = (m0 + row) < g.M;
            a_k[r] = 4 * c4;
            a_src[r] = A + (long long)min(m0 + row, g.M - 1) * g.lda + 4 * c4;
            a_dst[r] = (4 * c4) * LDA + row;
        }
    }
    const float* b_src[NLB]; bool b_ok[NLB]; int b_dst[NLB], b_k[NLB];
#pragma unroll
    for (int r = 0; r < NLB; ++r) {
        const int idx = tid + r * NTHREADS;
        if (BT) {
            const int row = idx / KV, c4 = idx % KV;
            b_ok[r] = (n0 + row) < g.b_rows;
            b_k[r] = 4 * c4;
            b_src[r] = B + (long long)min(n0 + row, g.b_rows - 1) * g.ldb + 4 * c4;
            b_dst[r] = (4 * c4) * LDB + row;
        } else {
            const int krow = idx / (BN / 4), c = idx % (BN / 4);
            b_ok[r] = (n0 + 4 * c) < g.N;
            b_k[r] = krow;
            b_src[r] = B + (b_ok[r] ? n0 + 4 * c : 0);
            b_dst[r] = krow * LDB + 4 * c;
        }
    }
    float4 ra[NLA], rb[NLB];

    // The validity mask of a staged float4 is computed when the load is ISSUED but applied when the value is
    // written to LDS (after the MFMA phase): masking right after the load would put the s_waitcnt in front of the
    // MFMAs and expose the whole global-load latency every k-tile.
    bool ra_ok[NLA], rb_ok[NLB];
#define R4D_ZERO_UNLESS(V, OK) { if (!(OK)) { V.x = 0.f; V.y = 0.f; V.z = 0.f; V.w = 0.f; } }
#define R4D_LOAD_TILES(KT)                                                                                         \
    {                                                                                                              \
        const int k0_ = (KT) * BK;                                                                                 \
        _Pragma("unroll") for (int r = 0; r < NLA; ++r) {                                                          \
            if (AT) {                                                                                              \
                const int gk_ = k0_ + a_k[r];                                                                      \
                ra_ok[r] = a_ok[r] && (gk_ < krows);                                                               \
                ra[r] = *reinterpret_cast<const float4*>(a_src[r] + (long long)min(gk_, krows - 1) * g.lda);       \
            } else {                                                                                               \
                const bool kin_ = k0_ + a_k[r] < g.a_cols;                                                         \
                ra_ok[r] = a_ok[r] && kin_;                                                                        \
                ra[r] = *reinterpret_cast<const float4*>(a_src[r] + (kin_ ? k0_ : 0));                             \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int r = 0; r < NLB; ++r) {                                                          \
            if (BT) {                                                                                              \
                const bool kin_ = k0_ + b_k[r] < g.K;                                                              \
                rb_ok[r] = b_ok[r] && kin_;                                                                        \
                rb[r] = *reinterpret_cast<const float4*>(b_src[r] + (kin_ ? k0_ : 0));                             \
            } else {                                                                                               \
                const int gk_ = k0_ + b_k[r];                                                                      \
                rb_ok[r] = b_ok[r] && (gk_ < krows);                                                               \
                rb[r] = *reinterpret_cast<const float4*>(b_src[r] + (long long)min(gk_, krows - 1) * g.ldb);       \
            }                                                                                                      \
        }                                                                                                          \
    }
#define R4D_STORE_TILES(BUF)                                                                                       \
    {                                                                                                              \
        _Pragma("unroll") for (int r = 0; r < NLA; ++r) {                                                          \
            R4D_ZERO_UNLESS(ra[r], ra_ok[r])                                                                       \
            if (AT) {                                                                                              \
                *reinterpret_cast<float4*>(&As[BUF][a_dst[r]]) = ra[r];                                            \
            } else {                                                                                               \
                float* p = &As[BUF][a_dst[r]];                                                                     \
                p[0] = ra[r].x; p[LDA] = ra[r].y; p[2 * LDA] = ra[r].z; p[3 * LDA] = ra[r].w;                      \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int r = 0; r < NLB; ++r) {                                                          \
            R4D_ZERO_UNLESS(rb[r], rb_ok[r])                                                                       \
            if (BT) {                                                                                              \
                float* p = &Bs[BUF][b_dst[r]];                                                                     \
                p[0] = rb[r].x; p[LDB] = rb[r].y; p[2 * LDB] = rb[r].z; p[3 * LDB] = rb[r].w;                      \
            } else {                                                                                               \
                *reinterpret_cast<float4*>(&Bs[BUF][b_dst[r]]) = rb[r];                                            \
            }                                                                                                      \
        }                                                                                                          \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    R4D_LOAD_TILES(0)
    R4D_STORE_TILES(0)
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) R4D_LOAD_TILES(kt + 1)                             // in flight under the MFMA phase
        const float* as = &As[cur][wm * WM + TM * li];
        const float* bs = &Bs[cur][wn * WN + TN * li];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
            FragLoad<TM>::ld(as + (2 * kk + lh) * LDA, a);
            FragLoad<TN>::ld(bs + (2 * kk + lh) * LDB, b);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) R4D_STORE_TILES(cur ^ 1)
        __syncthreads();
    }
#undef R4D_LOAD_TILES
#undef R4D_ZERO_UNLESS
#undef R4D_STORE_TILES

    // epilogue.  MFMA C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); with the interleave
    // the wave-tile row is TM*row + i and the wave-tile column TN*col + j.
    const int col0 = n0 + wn * WN + TN * li;
    float bias[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias[j] = (biasg && col0 + j < g.N) ? biasg[col0 + j] : 0.f;
    const bool full_cols = col0 + TN - 1 < g.N;
    const bool vec_ok = (TN > 1) && full_cols && (g.ldc % TN == 0) && (g.epilogue != EPI_RESIDUAL || g.ldr % TN == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float res[16][TN];
        if (g.epilogue == EPI_RESIDUAL) {
            // all 16 residual loads issued back to back from clamped (always valid) addresses
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(m0 + wm * WM + TM * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i, g.M - 1);
                const float* rp = residg + (long long)row * g.ldr;
                if (TN == 2 && vec_ok) {
                    const float2 t = *reinterpret_cast<const float2*>(rp + col0);
                    res[r][0] = t.x; res[r][1 % TN] = t.y;
                } else if (TN == 4 && vec_ok) {
                    const float4 t = *reinterpret_cast<const float4*>(rp + col0);
                    res[r][0] = t.x; res[r][1 % TN] = t.y; res[r][2 % TN] = t.z; res[r][3 % TN] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) res[r][j] = rp[min(col0 + j, g.N - 1)];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WM + TM * ((r & 3) + 8 * (r >> 2) + 4 * lh) + i;
            float v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r] + bias[j];
            if (g.epilogue == EPI_GELU) {
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = gelu_new_f(v[j]);
            } else if (g.epilogue == EPI_RESIDUAL) {
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] += res[r][j];
            } else if (g.epilogue == EPI_SCALE_DIV) {
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = v[j] / g.scale_div;
            } else if (g.epilogue == EPI_HALF_PLUS) {
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = (v[j] + 1.0f) / 2.0f;
            }
            if (row < g.M) {
                float* cp = C + (long long)row * g.ldc + col0;
                if (TN == 2 && vec_ok) *reinterpret_cast<float2*>(cp) = make_float2(v[0], v[1 % TN]);
                else if (TN == 4 && vec_ok) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1 % TN], v[2 % TN], v[3 % TN]);
                else {
#pragma unroll
                    for (int j = 0; j < TN; ++j) if (col0 + j < g.N) cp[j] = v[j];
                }
            }
        }
    }
}

// C[M,N] = sum over the S split-K partials part[s][M][N] (fixed order: deterministic), float4 per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, long long mn4, int S,
                                                            float* __restrict__ C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    const float4* p = reinterpret_cast<const float4*>(part) + i;
    float4 a = p[0];
    for (int s = 1; s < S; ++s) {
        const float4 b = p[(long long)s * mn4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(C)[i] = a;
}

// ---------------------------------------------------------------------------------------------- dispatch
struct TileCfg { int bm, bn, cls, blocks_per_cu, waves_per_block; double eff; };
// eff = steady-state fraction of the f32 MFMA peak measured for the tile on a saturating grid
// (tools/gemm_bench.py, MI355X); used only to rank tiles in the cost model below.  All tiles use BK = 32:
// half the barriers of BK = 16 and full 128-byte lines per A row (+3-5 % measured).  The 128x128 tile runs
// 8 wavefronts (4x2, wave tile 32x64): twice the L2->LDS reuse of 128x64 at the same 4 waves per SIMD.
static const TileCfg kTiles[] = {
    {128, 128, PK_GEMM_128x128_NN, 2, 8, 0.78},
    {128, 64, PK_GEMM_128x64_NN, 4, 4, 0.74},
    {64, 64, PK_GEMM_64x64_NN, 6, 4, 0.70},
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);
constexpr int BKT = 32;

template <int BM, int BN, int BK, int WGM = 2, int WGN = 2>
static int launch_variant(const GemmArgs& g, int cls, hipStream_t stream) {
    constexpr int NTHREADS = 64 * WGM * WGN;
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    dim3 grid(tiles, 1, g.nbatch);
    // algorithmic flop: 2MNK dense; the causal launches count only the lower-triangular half
    const double flop = (g.causal ? 1.0 : 2.0) * (double)g.M * g.N * g.K * g.nbatch;
    ProfScope prof(cls + (g.b_trans ? 1 : 0), flop, stream);
    GemmShape sh;
    sh.M = g.M; sh.N = g.N; sh.K = g.K; sh.lda = g.lda; sh.ldb = g.ldb; sh.ldc = g.ldc; sh.ldr = g.ldr;
    sh.b_rows = g.b_rows; sh.a_cols = g.a_cols; sh.nb1 = g.nb1; sh.epilogue = g.epilogue; sh.causal = g.causal;
    sh.sA0 = g.sA0; sh.sA1 = g.sA1; sh.sB0 = g.sB0; sh.sB1 = g.sB1; sh.sC0 = g.sC0; sh.sC1 = g.sC1;
    sh.scale_div = g.scale_div; sh.k_total = 0;
    if (BM == 128 && BN == 128) R4D_BRANCH(F32_128x128); else if (BM == 128) R4D_BRANCH(F32_128x64); else R4D_BRANCH(F32_64x64);
    if (g.b_trans) R4D_BRANCH(F32_NT); else R4D_BRANCH(F32_NN);
    if (g.b_trans)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WGM, WGN, true>), grid, dim3(NTHREADS), 0, stream, g.A, g.B, g.C, g.bias,
                           g.resid, sh);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WGM, WGN, false>), grid, dim3(NTHREADS), 0, stream, g.A, g.B, g.C, g.bias,
                           g.resid, sh);
    R4D_CHECK_LAUNCH("gemm_f32");
    return R4D_OK;
}

// Cost model.  Every CU executes ceil(blocks/256) tiles, `conc` of them concurrently; a SIMD needs about three
// resident wavefronts to keep its MFMA pipe fed (measured: one 4-wave workgroup alone on a CU reaches about a
// third of the steady-state rate), hence the occupancy factor.
static int pick_tile(const GemmArgs& g) {
    static int forced = -2;
    if (forced == -2) {
        const char* e = getenv("R4D_GEMM_TILE");      // tuning aid: 0..2 forces a tile shape
        forced = e ? atoi(e) : -1;
    }
    if (forced >= 0 && forced < kNumTiles) return forced;
    int best = kNumTiles - 1;
    double best_cost = 1e300;
    for (int t = 0; t < kNumTiles; ++t) {
        const TileCfg& c = kTiles[t];
        long long blocks = (long long)cdiv(g.M, c.bm) * cdiv(g.N, c.bn) * g.nbatch;
        if (g.causal == CAUSAL_QK) blocks = blocks / 2 + (long long)cdiv(g.M, c.bm) * g.nbatch / 2;   // lower triangle
        const long long per_cu = (blocks + 255) / 256;
        const double conc = (double)(per_cu < c.blocks_per_cu ? per_cu : c.blocks_per_cu);
        const double wps = conc * c.waves_per_block / 4.0;                 // resident wavefronts per SIMD
        const double eff = c.eff * (wps >= 3.0 ? 1.0 : wps / 3.0);
        const double cost = (double)per_cu * c.bm * c.bn / eff;
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return best;
}

// C[M,N] = A[Kt,M]^T . B[Kt,N]  (both operands row-major over the contraction index: weight gradients), split over the
// contraction so that a small [M,N] still fills the chip: S partial products of ceil(Kt/S) rows each in `scratch`
// (gemm_tn_scratch_floats), summed in a fixed order.  M, N multiples of 4.
static int tn_splits(int M, int N, int Kt) {
    const int tiles = cdiv(M, 128) * cdiv(N, 128);
    int S = cdiv(1024, tiles);                                       // ~4 workgroups per CU
    const int smax = cdiv(Kt, 256);                                  // at least 8 k-tiles per split
    if (S > smax) S = smax;
    if (S > 64) S = 64;
    return S < 1 ? 1 : S;
}
size_t gemm_tn_scratch_floats(int M, int N, int Kt) {
    const int S = tn_splits(M, N, Kt);
    return S > 1 ? (size_t)S * M * N : 0;
}
int launch_gemm_f32_tn(const float* A, const float* B, float* C, int M, int N, int Kt, int lda, int ldb, float* scratch,
                       hipStream_t stream, float* colsum_out, float* colsum_scratch, size_t colsum_scratch_floats, bool* colsum_done) {
    if (colsum_done) *colsum_done = false;
    R4D_REQUIRE(M > 0 && N > 0 && Kt > 0 && M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0,
                "gemm_tn: M=%d N=%d Kt=%d lda=%d ldb=%d (multiples of 4 wanted)", M, N, Kt, lda, ldb);
    R4D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "gemm_tn: 16-byte alignment");
    const int S = tn_splits(M, N, Kt);
    if (g_gemm_split3 && S > 1 && gemm_s3tn_supported(M, N, Kt, lda, ldb)) {    // bf16x3 form (gemm_s3tn.hip); S == 1: tiny problems stay here
        int Sx3 = 1;
        const int S3 = gemm_s3tn_slices(M, N, Kt, S);
        const bool cs = colsum_out && colsum_scratch && colsum_done && (size_t)S3 * N <= colsum_scratch_floats &&
                        ((uintptr_t)colsum_scratch % 16) == 0 && ((uintptr_t)colsum_out % 16) == 0;
        const int rc3 = launch_gemm_s3tn(A, B, scratch, M, N, Kt, lda, ldb, S3, &Sx3, stream, cs ? colsum_scratch : nullptr);
        if (rc3) return rc3;
        const long long mn4_ = (long long)M * N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn4_ + 255) / 256)), dim3(256), 0, stream, scratch, mn4_, Sx3, C);
        R4D_CHECK_LAUNCH("splitk_reduce");
        if (cs) {                                                    // the slices' column sums, added in slice order
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, stream, colsum_scratch, (long long)(N / 4), Sx3, colsum_out);
            R4D_CHECK_LAUNCH("colsum_reduce");
            *colsum_done = true;
        }
        return R4D_OK;
    }
    const int kper = cdiv(cdiv(Kt, S), BKT) * BKT;
    const int Sx = cdiv(Kt, kper);                                   // splits that have rows
    R4D_REQUIRE(S == 1 || scratch, "gemm_tn: split-K scratch missing");
    GemmShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.M = M; sh.N = N; sh.K = Sx > 1 ? kper : Kt; sh.lda = lda; sh.ldb = ldb; sh.ldc = N; sh.ldr = N;
    sh.b_rows = Kt; sh.a_cols = Kt; sh.nb1 = 1; sh.epilogue = EPI_NONE; sh.causal = CAUSAL_NONE; sh.k_total = Kt;
    sh.sA0 = (long long)kper * lda; sh.sB0 = (long long)kper * ldb; sh.sC0 = (long long)M * N; sh.scale_div = 1.f;
    if (Sx > 1) R4D_BRANCH(TN_SPLITK); else R4D_BRANCH(TN_SINGLE);
    {
        ProfScope prof(PK_GEMM_128x128_NN, 2.0 * M * N * Kt, stream);
        // (tried: 2 x 2 waves with 64 x 64 wave tiles -- twice the MFMAs per LDS read, half the waves: no change, 48.4 vs 48.1 ms backward)
        hipLaunchKernelGGL((gemm_f32_kernel<128, 128, BKT, 4, 2, false, true>), dim3(cdiv(M, 128) * cdiv(N, 128), 1, Sx), dim3(512), 0,
                           stream, A, B, Sx > 1 ? scratch : C, nullptr, nullptr, sh);
        R4D_CHECK_LAUNCH("gemm_f32_tn");
    }
    if (Sx > 1) {
        const long long mn4 = (long long)M * N / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, stream, scratch, mn4, Sx, C);
        R4D_CHECK_LAUNCH("splitk_reduce");
    }
    return R4D_OK;
}

int launch_gemm_f32(const GemmArgs& g0, hipStream_t stream) {
    GemmArgs g = g0;
    R4D_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    R4D_REQUIRE(g.nbatch >= 1 && g.nbatch <= 65535, "gemm: batch count %d outside [1, 65535] (grid.z limit)", g.nbatch);
    R4D_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: lda/ldb must be multiples of 4 (got %d,%d)", g.lda, g.ldb);
    R4D_REQUIRE(g.causal == CAUSAL_PV || g.K % 4 == 0, "gemm: K=%d must be a multiple of 4", g.K);
    R4D_REQUIRE(g.b_trans || g.N % 4 == 0, "gemm: N=%d must be a multiple of 4 for row-major B", g.N);
    R4D_REQUIRE(((uintptr_t)g.A % 16) == 0 && ((uintptr_t)g.B % 16) == 0, "gemm: A/B must be 16-byte aligned");
    if (g.a_cols <= 0) g.a_cols = g.K;
    {   // both operands k-contiguous: the b128-LDS kernel (gemm_f32_kc.hip); R4D_GEMM_KC=0 keeps this file's kernel
        static int use_kc = -1;
        if (use_kc < 0) { const char* e = getenv("R4D_GEMM_KC"); use_kc = e ? atoi(e) : 1; }
        const bool vec4 = ((g.lda | g.ldb | g.sA0 | g.sA1 | g.sB0 | g.sB1) & 3) == 0 && g.K % 32 == 0;   // float4 staging, whole k-tiles
        const bool fits = (long long)g.M * g.lda < (1ll << 29) && (long long)g.N * g.ldb < (1ll << 29) &&   // 32-bit byte offsets
                          128ll * g.ldc < (1ll << 29) && (!g.resid || 128ll * g.ldr < (1ll << 29));   // per-tile C / residual descriptors
        if (use_kc && g.b_trans && g.causal != CAUSAL_PV && g.a_cols == g.K && g.b_rows == g.N && vec4 && fits)
            return launch_gemm_f32_kc(g, stream);
    }
    const int t = pick_tile(g);
    switch (t) {
        case 0: return launch_variant<128, 128, BKT, 4, 2>(g, kTiles[0].cls, stream);
        case 1: return launch_variant<128, 64, BKT>(g, kTiles[1].cls, stream);
        default: return launch_variant<64, 64, BKT>(g, kTiles[2].cls, stream);
    }
}

}  // namespace r4d
