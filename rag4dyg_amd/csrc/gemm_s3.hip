// "bf16x3" GEMM: C[M,N] = epilogue(A[M,K] . W^T + bias) computed on the bf16 matrix cores at fp32 accuracy.
//
// Every fp32 operand is written as the EXACT sum of three bf16 numbers, x = hi + mid + lo (round-to-nearest at each step;
// 3 x 8 significant bits cover the 24 of an fp32), and a product a.b is evaluated as the six partial products whose
// weight is >= 2^-16 of it: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi -- each a v_mfma_f32_32x32x16_bf16 with an fp32
// accumulator (bf16 x bf16 is exact in fp32; the dropped terms mid.lo, lo.mid, lo.lo are <= 2^-23 |a.b|).  One fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 64 cycles for k = 2) becomes six bf16 MFMAs of 32 cycles for k = 16: 2.67x fewer matrix-pipe
// cycles per flop, and -- measured against float64 on the encoder's dominant shape (profiles/r03_s3_accuracy)
// -- a SMALLER error than the exact-f32 instruction, because the fp32 accumulation chains are 8x shorter (one rounding
// per 16 products and plane pair instead of one per product).
//
// Operands: A is the fp32 activation [M,K] as every other kernel leaves it -- it is split on the fly while its tile is
// staged (11 VALU instructions per two elements, done once per element and workgroup: 8 % of the MFMA cycles at
// BN = 256); W is a STATIC weight, split once per checkpoint into three k-contiguous bf16 planes [3][N][K]
// (r4d_split3_planes_bf16), so its staging is a plain copy.
//
// Tile structure (BK = 32, 8 wavefronts): LDS holds per stage three bf16 planes of the A tile and three of the W tile,
// rows of 64 bytes with the 16-byte chunk index XOR-ed with (row >> 2) & 3, which makes every ds_read_b128 of a fragment
// (lane = row, half-wave = k chunk) and every ds_write_b128 of the staging conflict-free without padding; a lane's six
// fragments of a k-step are 16 bytes each = the 8 consecutive k the bf16 MFMA wants.  Global loads are buffer loads with
// a loop-invariant lane offset and the k-tile (and the plane) in the scalar offset; out-of-range rows are clamped, K % 32
// == 0 is a precondition; register-staged pipeline with NBUF LDS stages and one barrier per k-tile.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "common.h"

#ifndef S3_NRS
#define S3_NRS 1   // register stages of the 128 x 256 tile; 2 measured no faster (K = 512: 157-166 vs 163-168 TFLOP/s, 240 vs 206 VGPRs): the k-tile loads are not latency-bound
#endif
#ifndef S3_DBG
#define S3_DBG 0   // tuning aid (tools/kc_ablate.sh gemm_s3.hip S3_DBG n): bit 0 drops the fragment reads, bit 1 the LDS staging stores (and the split), bit 2 the barrier, bit 3 the global loads, bit 4 only the split arithmetic
#endif

namespace r4d {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: low half = bf16(a), RNE
    const f32x2s v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2s));
}
// two fp32 -> packed (hi, hi), (mid, mid), (lo, lo)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

__device__ __forceinline__ f32x2s gelu_new_s3(f32x2s x) {            // the epilogue of gemm_f32_kc.hip, same instructions
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    const f32x2s a = x * x * k1 + k0;
    const f32x2s w = x * a;
    f32x2s e;
    e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
    e = e + 1.0f;
    f32x2s r;
    r.x = __builtin_amdgcn_rcpf(e.x); r.y = __builtin_amdgcn_rcpf(e.y);
    return x * r;
}
__device__ __forceinline__ float gelu_new_s3_1(float x) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

// d gelu_new / dx with the exponential of gelu_new_s3: tanh(u) = 1 - 2 / (1 + e^(2u))
__device__ __forceinline__ float gelu_new_grad_s3(float x) {
    const float c = 0.7978845608028654f;
    const float x2 = x * x;
    const float u2 = 2.0f * c * 1.4426950408889634f * x * __builtin_fmaf(x2, 0.044715f, 1.0f);     // 2u log2(e)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u2));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * __builtin_fmaf(x2, 3.0f * 0.044715f, 1.0f);
}

struct S3Shape {
    int M, N, K, lda, ldc, ldr;
    int plane_bytes;          // N * K * 2: distance between the bf16 planes of W
};

// ---------------------------------------------------------------------------------------------- weight planes
// w element (n, k) at w[k * ld_k + n * ld_n]  (reference Conv1D layout [K,N]: ld_k = N, ld_n = 1; a [N,K] copy: 1, K)
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ w, int N, int K, long long ld_k,
                                                            long long ld_n, unsigned short* __restrict__ planes) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 8 rows of 32 per pass
    const bool n_fast = ld_n == 1;                                    // read along the unit stride
    for (int r = ty; r < 32; r += 8) {
        const int n = n_fast ? n0 + tx : n0 + r, k = n_fast ? k0 + r : k0 + tx;
        const float v = (n < N && k < K) ? w[(long long)k * ld_k + (long long)n * ld_n] : 0.f;
        if (n_fast) tile[tx][r] = v; else tile[r][tx] = v;            // tile[n][k]
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                                // write along k
        const int n = n0 + r, k = k0 + tx;
        if (n >= N || k >= K) continue;
        const float x = tile[r][tx];
        unsigned h, m, l;
        split3_pair(x, 0.f, h, m, l);
        const size_t o = (size_t)n * K + k, P = (size_t)N * K;
        planes[o] = (unsigned short)(h & 0xffffu);
        planes[P + o] = (unsigned short)(m & 0xffffu);
        planes[2 * P + o] = (unsigned short)(l & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------------- the GEMM
// BF32: the second operand is NOT pre-split -- Bp points at fp32 rows [N][K] (k-contiguous, row stride K) that are split on the
// fly like A (the pool of the retrieval scoring GEMM: 4 bytes per element read instead of 6, no plane copy to keep)
//
// SCAN = NG > 0 (with BF32; the retrieval scoring GEMM): the ARITHMETIC OF THE POOL-SCAN KERNELS of score.hip, value for value, so that a
// score does not depend on whether its query travelled in a 32-query scan block or in a tile of this GEMM (one scoring
// arithmetic: tests/test_gpu_ops.py::test_scores_and_topk_do_not_depend_on_query_batching stays torch.equal):
//   * the scan splits the contraction over KW = K / (32 NG) wavefronts, wave w owning the 128-byte lines w, w + KW, ... of every
//     row, each wave accumulating its NG lines from zero and the KW partial sums then added in the order w = 0 .. KW-1: here the
//     k-tiles (= lines) are visited in exactly that order -- (w, g) -> line g KW + w -- the accumulator is folded into a second
//     one (`fin`) and cleared after every NG k-tiles;
//   * within a line, MFMA step s of lane half h holds k = 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3} (the scan's two
//     16-byte loads per step): the staging loads fetch those two pieces, 32 bytes apart, instead of 8 consecutive k;
//   * same split (split3_pair == scan_split_pair), same six products in the same order, queries as the MFMA's first operand.
template <int BM, int BN, int WGM, int WGN, int NBUF, int NRS, int EPI, bool BF32 = false, int SCAN = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4) void gemm_s3_kernel(
    const float* __restrict__ Ag, const unsigned short* __restrict__ Bp, float* __restrict__ Cg,
    const float* __restrict__ biasg, const float* __restrict__ residg, const S3Shape g) {
    constexpr int BK = 32;
    constexpr int NTHREADS = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NIA = BM * 4 / NTHREADS, NIB = BN * 4 / NTHREADS;   // (row, 8-k chunk) items per thread
    constexpr int A_PLANE = BM * 4, B_PLANE = BN * 4;                 // uint4 units (a row = 4 chunks of 16 bytes)
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    constexpr int D = NBUF - 1;                                       // k-tiles between the LDS store and its use
    static_assert(NIA >= 1 && NIB >= 1 && NIA <= 2 && NIB <= 2 && TM >= 1 && TN >= 1 && (NBUF == 2 || NBUF == 3), "tile");
    static_assert(NRS == 1 || (NRS == 2 && NBUF == 2), "two register stages: with the two-stage LDS ring only");
    __shared__ u32x4s lds[NBUF * STAGE];

    // XCD-aware grouped tile order (gemm_f32_kc.hip)
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
    constexpr int GROUP_M = 8;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int per_group = GROUP_M * tiles_n;
    const int grp = bid / per_group, first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (bid % per_group) % gsz, tile_n = (bid % per_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = g.K / BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, lh = lane >> 5;

    // staging coordinates: item idx = tid + i * NTHREADS -> row = idx >> 2, chunk = idx & 3
    int a_off[NIA], b_off[NIB], a_dst[NIA], b_dst[NIB];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        const int idx = tid + i * NTHREADS, row = idx >> 2, c = idx & 3;
        a_off[i] = (min(m0 + row, g.M - 1) * g.lda + (SCAN ? (c >> 1) * 16 + (c & 1) * 4 : c * 8)) * 4;
        a_dst[i] = row * 4 + (c ^ ((row >> 2) & 3));
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int idx = tid + i * NTHREADS, row = idx >> 2, c = idx & 3;
        b_off[i] = (min(n0 + row, g.N - 1) * g.K + (SCAN ? (c >> 1) * 16 + (c & 1) * 4 : c * 8)) * (BF32 ? 4 : 2);
        b_dst[i] = row * 4 + (c ^ ((row >> 2) & 3));
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ag), 0, (int)(((long long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Bp), 0, BF32 ? (int)((long long)g.N * g.K * 4) : 3 * g.plane_bytes, 0x00020000);

    // staging registers (fully unrolled indices only).  NRS = 2: two register stages -- a k-tile's loads are requested TWO
    // iterations before they are split and stored (at K = 512 the A rows of a tile come from beyond the XCD's L2, one
    // iteration of flight time does not cover that latency)
    u32x4s ra[NRS][NIA][2], rb[NRS][NIB][3];            // BF32: rb[..][0..1] = the 8 fp32 of the item, split at the LDS store
    static_assert(SCAN == 0 || BF32, "scan order: with the fp32 second operand only");
    constexpr int P2 = SCAN ? 32 : 16;                                // byte distance of an item's two 16-byte pieces
    const int scan_kw = SCAN ? nkt / (SCAN ? SCAN : 1) : 1;           // wavefronts the scan kernels split this K over
#define S3_LOAD(RS, KT)                                                                            \
    {                                                                                              \
        const int kq_ = min((KT), nkt - 1);                                                        \
        const int kt_ = SCAN ? (kq_ % (SCAN ? SCAN : 1)) * scan_kw + kq_ / (SCAN ? SCAN : 1) : kq_;   /* iteration -> line g KW + w */ \
        _Pragma("unroll") for (int i = 0; i < NIA; ++i) {                                          \
            ra[RS][i][0] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[i], kt_ * (BK * 4), 0); \
            ra[RS][i][1] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[i] + P2, kt_ * (BK * 4), 0); \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NIB; ++i) {                                          \
            if (BF32) {                                                                            \
                rb[RS][i][0] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i], kt_ * (BK * 4), 0); \
                rb[RS][i][1] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i] + P2, kt_ * (BK * 4), 0); \
            } else {                                                                               \
                _Pragma("unroll") for (int p = 0; p < 3; ++p)                                      \
                    rb[RS][i][p] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i], kt_ * (BK * 2) + p * g.plane_bytes, 0); \
            }                                                                                      \
        }                                                                                          \
    }
#define S3_STORE(RS, STG)                                                                          \
    {                                                                                              \
        u32x4s* sa_ = lds + (STG) * STAGE;                                                         \
        u32x4s* sb_ = sa_ + 3 * A_PLANE;                                                           \
        _Pragma("unroll") for (int i = 0; i < NIA; ++i) {                                          \
            u32x4s h_, m_, l_;                                                                     \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                        \
                /* (cast the WHOLE vector: __builtin_bit_cast on an ext-vector element reads element 0) */ \
                const f32x4s src_ = __builtin_bit_cast(f32x4s, ra[RS][i][q >> 1]);                 \
                const float x0_ = src_[(q & 1) * 2], x1_ = src_[(q & 1) * 2 + 1];                  \
                unsigned hh_, mm_, ll_;                                                            \
                if (S3_DBG & 16) { hh_ = __builtin_bit_cast(unsigned, x0_); mm_ = __builtin_bit_cast(unsigned, x1_); ll_ = hh_ ^ mm_; } \
                else split3_pair(x0_, x1_, hh_, mm_, ll_);                                         \
                h_[q] = hh_; m_[q] = mm_; l_[q] = ll_;                                             \
            }                                                                                      \
            sa_[a_dst[i]] = h_; sa_[A_PLANE + a_dst[i]] = m_; sa_[2 * A_PLANE + a_dst[i]] = l_;    \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NIB; ++i) {                                          \
            if (BF32) {                                                                            \
                u32x4s h_, m_, l_;                                                                 \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
                    const f32x4s src_ = __builtin_bit_cast(f32x4s, rb[RS][i][q >> 1]);            \
                    unsigned hh_, mm_, ll_;                                                        \
                    split3_pair(src_[(q & 1) * 2], src_[(q & 1) * 2 + 1], hh_, mm_, ll_);          \
                    h_[q] = hh_; m_[q] = mm_; l_[q] = ll_;                                         \
                }                                                                                  \
                sb_[b_dst[i]] = h_; sb_[B_PLANE + b_dst[i]] = m_; sb_[2 * B_PLANE + b_dst[i]] = l_; \
            } else {                                                                               \
                _Pragma("unroll") for (int p = 0; p < 3; ++p) sb_[p * B_PLANE + b_dst[i]] = rb[RS][i][p]; \
            }                                                                                      \
        }                                                                                          \
    }

    // fragment addresses: lane (li, lh), k-step s -> chunk 2s + lh of row li (+ 32 per tile)
    const int fq = (li >> 2) & 3;
    const int f_off0 = li * 4 + ((0 + lh) ^ fq), f_off1 = li * 4 + ((2 + lh) ^ fq);
    const int fa_base = wm * WM * 4, fb_base = 3 * A_PLANE + wn * WN * 4;

    f32x16s acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16s fin[SCAN ? TM : 1][SCAN ? TN : 1];                       // SCAN: the sum of the finished slices' partial sums
    if constexpr (SCAN != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[i][j][r] = 0.f;
    }

#define S3_MFMA(A_, B_, I_, J_) \
    acc[I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, A_), __builtin_bit_cast(bf16x8s, B_), acc[I_][J_], 0, 0, 0)
    // fragment registers, two sets: the reads of k-step s+1 travel under the MFMAs of k-step s
    // (SCAN: ONE set -- the second accumulator takes the registers -- a fragment is re-read as soon as the last MFMA that uses it has
    //  issued: program order and the register dependences give the progressive refill, the other wavefront of the SIMD covers the rest)
    u32x4s fa[SCAN ? 1 : 2][TM][3], fb[SCAN ? 1 : 2][TN][3];
#define S3_FRAGS(SET_, STG, S)                                                                     \
    {                                                                                              \
        constexpr int SET = SCAN ? 0 : (SET_);                                                     \
        const u32x4s* st_ = lds + (STG) * STAGE;                                                   \
        const int fo_ = (S) ? f_off1 : f_off0;                                                     \
        /* in the order the MFMAs want them: lo(A) . hi(B) first */                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][2] = (S3_DBG & 1) ? dbg_frag : st_[fa_base + 2 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][0] = (S3_DBG & 1) ? dbg_frag : st_[fb_base + 0 * B_PLANE + j * 128 + fo_]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][0] = (S3_DBG & 1) ? dbg_frag : st_[fa_base + 0 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][2] = (S3_DBG & 1) ? dbg_frag : st_[fb_base + 2 * B_PLANE + j * 128 + fo_]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][1] = (S3_DBG & 1) ? dbg_frag : st_[fa_base + 1 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][1] = (S3_DBG & 1) ? dbg_frag : st_[fb_base + 1 * B_PLANE + j * 128 + fo_]; \
    }
    /* smallest partial products first; consecutive MFMAs go to different accumulators */
#define S3_MFMAS(SET_)                                                                             \
    {                                                                                              \
        constexpr int SET = SCAN ? 0 : (SET_);                                                     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][2], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][0], fb[SET][j][2], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][1], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][1], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][0], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) S3_MFMA(fa[SET][i][0], fb[SET][j][0], i, j); \
    }

    u32x4s dbg_frag = {(unsigned)tid, 0x3f803f80u, 0x3f803f80u, (unsigned)lane};   // (ablation builds only)
    if (S3_DBG & 1) asm volatile("" : "+v"(dbg_frag));
    // prologue: k-tiles 0 .. D-1 into their stages, k-tile D (and D+1 with two register stages) into the staging registers
    if constexpr (NRS == 2) {
        S3_LOAD(0, 0)
        S3_LOAD(1, 1)                                                 // k-tiles 0 and 1 travel together: one exposed latency
        S3_STORE(0, 0)
        S3_LOAD(0, 2)
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t) {
            S3_LOAD(0, t)
            S3_STORE(0, t)
        }
        S3_LOAD(0, D)
    }
    __syncthreads();

    // iteration kt (the staging registers hold k-tile kt+D, requested during iteration kt-1):
    //   [after the barrier] the 3(TM+TN) fragment reads of k-step 0 of stage kt % NBUF;
    //   k-step 0: its 6 TM TN MFMAs with, between them, the reads of k-step 1, the split of the staged A elements and the
    //             LDS stores of k-tile kt+D into stage (kt+D) % NBUF (last read in iteration kt-1: every wave is past that barrier);
    //   k-step 1: its MFMAs with the global loads of k-tile kt+D+1 between them (in flight until the middle of iteration kt+1);
    //   barrier.
    // sched_group_barrier pins that interleaving: left alone, the scheduler reads each fragment group right in front of
    // its MFMAs (exposed LDS latency) and sinks the loads to the end of the iteration (half an iteration of flight time).
    constexpr int NMF = 6 * TM * TN, NFR = 3 * (TM + TN), NDW = 3 * (NIA + NIB), NVM = 2 * NIA + (BF32 ? 2 : 3) * NIB;
    // (two register stages: the set that holds k-tile kt+1 has the parity of its LDS stage WR; it is refilled with k-tile kt+3)
#define S3_ITER(CUR, WR)                                                                           \
    {                                                                                              \
        const int RS_ = NRS == 2 ? ((WR) & 1) : 0;                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        S3_FRAGS(0, CUR, 0)                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(S3_DBG & 2)) S3_STORE(RS_, WR)                                                       \
        if (!SCAN) S3_FRAGS(1, CUR, 1)                                                             \
        S3_MFMAS(0)                                                                                \
        if (SCAN) S3_FRAGS(0, CUR, 1)       /* one fragment set: k-step 1 is read into the registers k-step 0 leaves, in program order */ \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (!SCAN && m_ < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              \
            if (SCAN && m_ >= TM * TN && m_ < TM * TN + NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                     \
            if (m_ >= NMF - 2 * NDW && ((NMF - 1 - m_) & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(S3_DBG & 8)) S3_LOAD(RS_, kt + D + NRS)                                              \
        S3_MFMAS(1)                                                                                \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (SCAN && (kt % (SCAN ? SCAN : 1)) == (SCAN ? SCAN : 1) - 1) {   /* a slice ends: fold its partial sum, start the next from zero */ \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)              \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) { fin[SCAN ? i : 0][SCAN ? j : 0][r] += acc[i][j][r]; acc[i][j][r] = 0.f; } \
        }                                                                                          \
        if (!(S3_DBG & 4)) __syncthreads();                                                        \
    }
    int kt = 0;
    if constexpr (NBUF == 2) {
        for (; kt + 1 < nkt; kt += 2) {
            S3_ITER(0, 1)
            { ++kt; S3_ITER(1, 0) }
            --kt;
        }
        if (kt < nkt) S3_ITER(0, 1)
    } else {
        for (; kt + 2 < nkt; kt += 3) {                               // compile-time stages
            S3_ITER(0, 2)
            { ++kt; S3_ITER(1, 0) }
            { ++kt; S3_ITER(2, 1) }
            kt -= 2;
        }
        int cur = 0, wr = 2;                                          // kt is a multiple of 3 here: 0..2 k-tiles left
        for (; kt < nkt; ++kt) {
            S3_ITER(cur, wr)
            cur = cur == 2 ? 0 : cur + 1;
            wr = wr == 2 ? 0 : wr + 1;
        }
    }
#undef S3_ITER
#undef S3_FRAGS
#undef S3_MFMAS
#undef S3_MFMA
#undef S3_STORE
#undef S3_LOAD

    if constexpr (SCAN != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = fin[i][j];
    }
#if S3_DBG & 32
    {   // ablation: no epilogue at all (one conditional store keeps the accumulators alive)
        float ssum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += acc[i][j][r];
        if (ssum == 12345.678f) Cg[0] = ssum;
        return;
    }
#endif
    // epilogue: the one of gemm_f32_kc.hip (C/D layout is dtype-independent: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    float* __restrict__ C = Cg;
    const bool interior = (m0 + BM <= g.M) & (n0 + BN <= g.N);       // wave-uniform
    if (interior) {
        const int lane_c = ((wm * WM + 4 * lh) * g.ldc + wn * WN + li) * 4;
        const int lane_r = ((wm * WM + 4 * lh) * g.ldr + wn * WN + li) * 4;
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            C + (long long)m0 * g.ldc + n0, 0, ((BM - 1) * g.ldc + BN) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>((EPI == EPI_RESIDUAL || EPI >= EPI_GELU_KEEP) ? residg + (long long)m0 * g.ldr + n0 : Ag), 0,
            (EPI == EPI_RESIDUAL || EPI >= EPI_GELU_KEEP) ? ((BM - 1) * g.ldr + BN) * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bias = biasg ? biasg[n0 + wn * WN + j * 32 + li] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float res[16];
                if (EPI == EPI_RESIDUAL || EPI == EPI_GELU_GRAD) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            r_rsrc, lane_r, ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0));
                }
#pragma unroll
                for (int r2 = 0; r2 < 16; r2 += 2) {
                    f32x2s v2 = {acc[i][j][r2] + bias, acc[i][j][r2 + 1] + bias};
                    if (EPI == EPI_GELU_KEEP) {                       // the pre-activation, for the backward pass
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int r = r2 + h2;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, h2 ? v2.y : v2.x), r_rsrc, lane_r,
                                                                  ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0);
                        }
                    }
                    if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) v2 = gelu_new_s3(v2);
                    else if (EPI == EPI_RESIDUAL) { v2.x += res[r2]; v2.y += res[r2 + 1]; }
                    else if (EPI == EPI_GELU_GRAD) { v2.x *= gelu_new_grad_s3(res[r2]); v2.y *= gelu_new_grad_s3(res[r2 + 1]); }
                    else if (EPI == EPI_HALF_PLUS) { v2.x = (v2.x + 1.0f) / 2.0f; v2.y = (v2.y + 1.0f) / 2.0f; }     // train_retriever.py:438
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = r2 + h2;
                        const float v = h2 ? v2.y : v2.x;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), c_rsrc, lane_c,
                                                              ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32) * 4, 0);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {                                   // edge tiles: clamped reads, guarded stores
        const int col = n0 + wn * WN + j * 32 + li;
        const bool col_ok = col < g.N;
        const int colc = min(col, g.N - 1);
        const float bias = biasg ? biasg[colc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float res[16];
            if (EPI == EPI_RESIDUAL || EPI == EPI_GELU_GRAD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, g.M - 1);
                    res[r] = residg[(long long)row * g.ldr + colc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bias;
                if (EPI == EPI_GELU_KEEP && row < g.M && col_ok) const_cast<float*>(residg)[(long long)row * g.ldr + col] = v;
                if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) v = gelu_new_s3_1(v);
                else if (EPI == EPI_RESIDUAL) v += res[r];
                else if (EPI == EPI_GELU_GRAD) v *= gelu_new_grad_s3(res[r]);
                else if (EPI == EPI_HALF_PLUS) v = (v + 1.0f) / 2.0f;
                if (row < g.M && col_ok) C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- persistent form of the 128 x 256 tile
// One workgroup per CU walks tiles v, v + G, v + 2G ... (G = grid size, a multiple of 8: the XCD of a tile sequence is fixed)
// and the k-loop pipeline runs ACROSS tile boundaries: the last two iterations of a tile request and stage the first two
// k-tiles of the NEXT tile, so a tile starts with its operands already in LDS / in flight instead of behind a workgroup
// launch and two exposed memory latencies, and the epilogue's stores (younger than those loads in the in-order vmcnt queue)
// drain under the next tile's first iterations.  At K = 512 a 128 x 256 tile is only 16 iterations long: launch, prologue
// and epilogue were ~14 us of a 51 us tile (tools/s3_bench.py, S3_DBG ablations).  Requires an even number of k-tiles.
// Tiles beyond the first one per workgroup are handed out DYNAMICALLY (a static stride loses the balancing a launch's own
// dispatcher gives: measured 8-17 % slower at 4 tiles per CU): eight ticket counters, one per XCD class (blockIdx & 7), so a
// workgroup keeps drawing tiles whose operands its XCD's L2 already holds; the ticket of the tile after next is requested a
// whole tile ahead (one atomic per tile and workgroup, latency hidden).  The counters of a launch live in one of 64 slots
// (host round-robin: launches on different streams do not share one); the last workgroup to leave resets its slot, so every
// slot is zero whenever a launch starts -- no memset in the stream.
__device__ unsigned g_s3p_slots[64][16];

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_s3p_kernel(
    const float* __restrict__ Ag, const unsigned short* __restrict__ Bp, float* __restrict__ Cg,
    const float* __restrict__ biasg, const float* __restrict__ residg, const S3Shape g, const int ntiles, const int slot) {
    constexpr int BM = 128, BN = 256, BK = 32, WGN = 4, WM = 64, WN = 64, TM = 2, TN = 2;
    constexpr int NIB = 2;                                            // (row, chunk) items of the W tile per thread (A: one)
    constexpr int A_PLANE = BM * 4, B_PLANE = BN * 4, STAGE = 3 * (A_PLANE + B_PLANE);
    __shared__ u32x4s lds[2 * STAGE];
    __shared__ int s_next;                                            // virtual index of the tile AFTER the current one
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, lh = lane >> 5;
    const int nkt = g.K / BK;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    unsigned* const tickets = g_s3p_slots[slot];
    const int cls = blockIdx.x & 7, G = (int)gridDim.x;
    // XCD-aware grouped tile order over the VIRTUAL block index v (v & 7 == blockIdx.x & 7 for every tile of this workgroup)
    auto origin = [&](int v, int& m0, int& n0) {
        const int xq = ntiles >> 3, xr = ntiles & 7, xcd = v & 7;
        const int bid = xcd * xq + min(xcd, xr) + (v >> 3);
        constexpr int GROUP_M = 8;
        const int per_group = GROUP_M * tiles_n;
        const int grp = bid / per_group, first_m = grp * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        m0 = (first_m + (bid % per_group) % gsz) * BM;
        n0 = ((bid % per_group) / gsz) * BN;
    };
    const int st_row = tid >> 2, st_c = tid & 3;                      // staging item: row, 8-k chunk (A: rows 0..127; W: + 128)
    const int st_dst = st_row * 4 + (st_c ^ ((st_row >> 2) & 3));
    auto offsets = [&](int m0, int n0, int& ao, int (&bo)[NIB]) {
        ao = (min(m0 + st_row, g.M - 1) * g.lda + st_c * 8) * 4;
#pragma unroll
        for (int i = 0; i < NIB; ++i) bo[i] = (min(n0 + st_row + i * 128, g.N - 1) * g.K + st_c * 8) * 2;
    };
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ag), 0, (int)(((long long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Bp), 0, 3 * g.plane_bytes, 0x00020000);
    u32x4s ra[2], rb[NIB][3];
#define P_LOAD(AO, BO, KT)                                                                         \
    {                                                                                              \
        ra[0] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, AO, (KT) * (BK * 4), 0);             \
        ra[1] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, AO + 16, (KT) * (BK * 4), 0);        \
        _Pragma("unroll") for (int i = 0; i < NIB; ++i)                                            \
            _Pragma("unroll") for (int p = 0; p < 3; ++p)                                          \
                rb[i][p] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, BO[i], (KT) * (BK * 2) + p * g.plane_bytes, 0); \
    }
#define P_STORE(STG)                                                                               \
    {                                                                                              \
        u32x4s* sa_ = lds + (STG) * STAGE + st_dst;                                                \
        u32x4s h_, m_, l_;                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
            const f32x4s src_ = __builtin_bit_cast(f32x4s, ra[q >> 1]);                            \
            unsigned hh_, mm_, ll_;                                                                \
            split3_pair(src_[(q & 1) * 2], src_[(q & 1) * 2 + 1], hh_, mm_, ll_);                  \
            h_[q] = hh_; m_[q] = mm_; l_[q] = ll_;                                                 \
        }                                                                                          \
        sa_[0] = h_; sa_[A_PLANE] = m_; sa_[2 * A_PLANE] = l_;                                     \
        _Pragma("unroll") for (int i = 0; i < NIB; ++i)                                            \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) sa_[3 * A_PLANE + p * B_PLANE + i * 512] = rb[i][p]; \
    }
    const int fq = (li >> 2) & 3;
    const int f_off0 = li * 4 + ((0 + lh) ^ fq), f_off1 = li * 4 + ((2 + lh) ^ fq);
    const int fa_base = wm * WM * 4, fb_base = 3 * A_PLANE + wn * WN * 4;
    u32x4s fa[2][TM][3], fb[2][TN][3];
    f32x16s acc[TM][TN];
#define P_FRAGS(SET, STG, S)                                                                       \
    {                                                                                              \
        const u32x4s* st_ = lds + (STG) * STAGE;                                                   \
        const int fo_ = (S) ? f_off1 : f_off0;                                                     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][2] = st_[fa_base + 2 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][0] = st_[fb_base + 0 * B_PLANE + j * 128 + fo_]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][0] = st_[fa_base + 0 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][2] = st_[fb_base + 2 * B_PLANE + j * 128 + fo_]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][1] = st_[fa_base + 1 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][1] = st_[fb_base + 1 * B_PLANE + j * 128 + fo_]; \
    }
#define P_MFMA(A_, B_, I_, J_) \
    acc[I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, A_), __builtin_bit_cast(bf16x8s, B_), acc[I_][J_], 0, 0, 0)
#define P_MFMAS(SET)                                                                               \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][2], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][0], fb[SET][j][2], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][1], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][1], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][0], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) P_MFMA(fa[SET][i][0], fb[SET][j][0], i, j); \
    }
    // same pinned interleaving as gemm_s3_kernel: k-step 0's MFMAs over the reads of k-step 1, the split and the LDS stores;
    // k-step 1's MFMAs over the global loads
#define P_ITER(CUR, WR, AO, BO, KTL)                                                               \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        P_FRAGS(0, CUR, 0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        P_STORE(WR)                                                                                \
        P_FRAGS(1, CUR, 1)                                                                         \
        P_MFMAS(0)                                                                                 \
        _Pragma("unroll") for (int m_ = 0; m_ < 24; ++m_) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                     \
            if (m_ >= 6 && ((23 - m_) & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        P_LOAD(AO, BO, KTL)                                                                        \
        P_MFMAS(1)                                                                                 \
        _Pragma("unroll") for (int m_ = 0; m_ < 24; ++m_) {                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                         \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        __syncthreads();                                                                           \
    }

    int m0, n0;
    origin((int)blockIdx.x, m0, n0);
    int ao, bo[NIB], aon, bon[NIB];
    offsets(m0, n0, ao, bo);
    P_LOAD(ao, bo, 0)
    if (tid == 0) s_next = G + 8 * (int)__hip_atomic_fetch_add(&tickets[cls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + cls;
    P_STORE(0)
    P_LOAD(ao, bo, 1)
    __syncthreads();
    for (;;) {
        const int vn = s_next;                                        // drawn one tile ago; every wave reads it before the tile's first barrier
        const bool has_next = vn < ntiles;                            // workgroup-uniform
        unsigned tk = 0;
        if (tid == 0 && has_next) tk = __hip_atomic_fetch_add(&tickets[cls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int m0n, n0n;
        origin(has_next ? vn : (int)blockIdx.x, m0n, n0n);
        offsets(m0n, n0n, aon, bon);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int kt = 0;
        for (; kt + 2 < nkt; kt += 2) {
            P_ITER(0, 1, ao, bo, kt + 2)
            P_ITER(1, 0, ao, bo, kt + 3)
        }
        // the tile's last two iterations stage and request the NEXT tile's k-tiles 0 and 1 (its own again when there is none)
        P_ITER(0, 1, aon, bon, 0)
        if (tid == 0) s_next = has_next ? G + 8 * (int)tk + cls : ntiles;      // behind >= 1 barrier of this tile, published by the next
        P_ITER(1, 0, aon, bon, 1)

        // epilogue (accumulator layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) per 32 x 32 tile)
        const bool interior = (m0 + BM <= g.M) & (n0 + BN <= g.N);
        if (interior) {
            const int lane_c = ((wm * WM + 4 * lh) * g.ldc + wn * WN + li) * 4;
            const int lane_r = ((wm * WM + 4 * lh) * g.ldr + wn * WN + li) * 4;
            const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                Cg + (long long)m0 * g.ldc + n0, 0, ((BM - 1) * g.ldc + BN) * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>((EPI == EPI_RESIDUAL || EPI >= EPI_GELU_KEEP) ? residg + (long long)m0 * g.ldr + n0 : Ag), 0,
                (EPI == EPI_RESIDUAL || EPI >= EPI_GELU_KEEP) ? ((BM - 1) * g.ldr + BN) * 4 : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float bias = biasg ? biasg[n0 + wn * WN + j * 32 + li] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float res[16];
                    if (EPI == EPI_RESIDUAL || EPI == EPI_GELU_GRAD) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                r_rsrc, lane_r, ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0));
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 16; r2 += 2) {
                        f32x2s v2 = {acc[i][j][r2] + bias, acc[i][j][r2 + 1] + bias};
                        if (EPI == EPI_GELU_KEEP) {
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const int r = r2 + h2;
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, h2 ? v2.y : v2.x), r_rsrc, lane_r,
                                                                      ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0);
                            }
                        }
                        if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) v2 = gelu_new_s3(v2);
                        else if (EPI == EPI_RESIDUAL) { v2.x += res[r2]; v2.y += res[r2 + 1]; }
                        else if (EPI == EPI_GELU_GRAD) { v2.x *= gelu_new_grad_s3(res[r2]); v2.y *= gelu_new_grad_s3(res[r2 + 1]); }
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int r = r2 + h2;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, h2 ? v2.y : v2.x), c_rsrc, lane_c,
                                                                  ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32) * 4, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {                           // edge tiles: clamped reads, guarded stores
                const int col = n0 + wn * WN + j * 32 + li;
                const bool col_ok = col < g.N;
                const int colc = min(col, g.N - 1);
                const float bias = biasg ? biasg[colc] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float vv = acc[i][j][r] + bias;
                        if (EPI == EPI_GELU_KEEP && row < g.M && col_ok) const_cast<float*>(residg)[(long long)row * g.ldr + col] = vv;
                        if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) vv = gelu_new_s3_1(vv);
                        else if (EPI == EPI_RESIDUAL) vv += residg[(long long)min(row, g.M - 1) * g.ldr + colc];
                        else if (EPI == EPI_GELU_GRAD) vv *= gelu_new_grad_s3(residg[(long long)min(row, g.M - 1) * g.ldr + colc]);
                        if (row < g.M && col_ok) Cg[(long long)row * g.ldc + col] = vv;
                    }
                }
            }
        }
        if (!has_next) break;
        m0 = m0n; n0 = n0n; ao = aon;
#pragma unroll
        for (int i = 0; i < NIB; ++i) bo[i] = bon[i];
    }
    if (tid == 0) {                                                   // last workgroup out resets the slot for its next launch
        if (__hip_atomic_fetch_add(&tickets[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(&tickets[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef P_ITER
#undef P_MFMAS
#undef P_MFMA
#undef P_FRAGS
#undef P_STORE
#undef P_LOAD
}

// ---------------------------------------------------------------------------------------------- host side
struct S3Tile { int bm, bn, cls; double eff; };
static const S3Tile kS3[] = {
    {128, 256, PK_GEMM_S3_128x256, 1.0},
    {128, 128, PK_GEMM_S3_128x128, 0.9},
};
constexpr int kNumS3 = 2;

template <int BM, int BN, int WGM, int WGN, int NBUF, int NRS>
static int launch_s3(const S3Args& a, int cls, hipStream_t stream) {
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    ProfScope prof(cls, 2.0 * (double)a.M * a.N * a.K, stream);
    S3Shape sh;
    sh.M = a.M; sh.N = a.N; sh.K = a.K; sh.lda = a.lda; sh.ldc = a.ldc; sh.ldr = a.ldr; sh.plane_bytes = a.N * a.K * 2;
#define S3_LAUNCH_(E)                                                                              \
    hipLaunchKernelGGL((gemm_s3_kernel<BM, BN, WGM, WGN, NBUF, NRS, E>), dim3(tiles), dim3(64 * WGM * WGN), 0, stream, a.A, \
                       a.planes, a.C, a.bias, a.resid, sh)
    switch (a.epilogue) {
        case EPI_NONE: S3_LAUNCH_(EPI_NONE); break;
        case EPI_GELU: S3_LAUNCH_(EPI_GELU); break;
        case EPI_RESIDUAL: S3_LAUNCH_(EPI_RESIDUAL); break;
        case EPI_GELU_KEEP: S3_LAUNCH_(EPI_GELU_KEEP); break;
        case EPI_GELU_GRAD: S3_LAUNCH_(EPI_GELU_GRAD); break;
        default: set_error("gemm_s3: unknown epilogue %d", a.epilogue); return R4D_ERR_INVALID;
    }
#undef S3_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_s3");
    return R4D_OK;
}

static int launch_s3p(const S3Args& a, hipStream_t stream) {
    const int ntiles = cdiv(a.M, 128) * cdiv(a.N, 256);
    ProfScope prof(PK_GEMM_S3_128x256, 2.0 * (double)a.M * a.N * a.K, stream);
    S3Shape sh;
    sh.M = a.M; sh.N = a.N; sh.K = a.K; sh.lda = a.lda; sh.ldc = a.ldc; sh.ldr = a.ldr; sh.plane_bytes = a.N * a.K * 2;
    const int grid = 256;                                            // one workgroup per CU (144 KB of LDS each); ntiles > 256 here
    static std::atomic<unsigned> seq{0};
    const int slot = (int)(seq.fetch_add(1) % 64u);
#define SP_LAUNCH_(E) hipLaunchKernelGGL((gemm_s3p_kernel<E>), dim3(grid), dim3(512), 0, stream, a.A, a.planes, a.C, a.bias, a.resid, sh, ntiles, slot)
    switch (a.epilogue) {
        case EPI_NONE: SP_LAUNCH_(EPI_NONE); break;
        case EPI_GELU: SP_LAUNCH_(EPI_GELU); break;
        case EPI_RESIDUAL: SP_LAUNCH_(EPI_RESIDUAL); break;
        case EPI_GELU_KEEP: SP_LAUNCH_(EPI_GELU_KEEP); break;
        case EPI_GELU_GRAD: SP_LAUNCH_(EPI_GELU_GRAD); break;
        default: set_error("gemm_s3: unknown epilogue %d", a.epilogue); return R4D_ERR_INVALID;
    }
#undef SP_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_s3p");
    return R4D_OK;
}

static int s3_launch_tile(const S3Args& a, int t, hipStream_t stream) {
    static int pers = -1;
    if (pers < 0) { const char* e = getenv("R4D_GEMM_S3_PERSISTENT"); pers = e ? atoi(e) : 1; }   // tuning aid: 0 = one launch slot per tile
    // measured (M = 63232; tools/s3_bench.py): K 512 N 1536 593 -> 576 us, K 512 N 2048 + gelu 689 -> 665 us, M 8864 81 -> 75 us;
    // with the RESIDUAL epilogue the persistent form is SLOWER (K 512 N 512: 189 -> 227 us, K 2048 N 512: 625 -> 700 us, static and
    // dynamic tile hand-out alike), so those launches keep one launch slot per tile (pers == 2 forces it for them too)
    const bool loads_in_epilogue = a.epilogue == EPI_RESIDUAL || a.epilogue == EPI_GELU_GRAD;     // (GELU_GRAD: 57.6 vs 57.9 ms per training step)
    // The persistent form draws its tiles from ticket counters in one of 64 process-wide slots, chosen round-robin at LAUNCH time
    // (g_s3p_slots): two launches must never share a slot while both run.  Eager launches on up to 64 streams cannot (a slot comes
    // round again only after 63 later launches have been issued behind it); a launch CAPTURED into a graph would bake its slot in
    // and could be replayed beside an eager launch that drew the same one -- both would skip tiles, silently -- so a
    // capturing stream takes the one-workgroup-per-tile form, which keeps no state.
    bool capturing = false;
    if (t == 0 && pers) {
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cst) != hipSuccess) { (void)hipGetLastError(); cst = hipStreamCaptureStatusNone; }
        capturing = cst != hipStreamCaptureStatusNone;
    }
    if (t == 0 && pers && !capturing && (pers == 2 || !loads_in_epilogue) && (a.K / 32) % 2 == 0 && cdiv(a.M, 128) * cdiv(a.N, 256) > 256) {
        R4D_BRANCH(S3_PERSISTENT);
        return launch_s3p(a, stream);
    }
    if (t == 0) R4D_BRANCH(S3_128x256); else R4D_BRANCH(S3_128x128);
    switch (t) {
        case 0: return launch_s3<128, 256, 2, 4, 2, S3_NRS>(a, kS3[0].cls, stream);
        default: return launch_s3<128, 128, 2, 4, 3, 1>(a, kS3[1].cls, stream);
    }
}

bool gemm_s3_supported(int M, int K, int N) {
    return M >= 1 && K >= 32 && K % 32 == 0 && N >= 1 && (long long)N * K * 6 < (1ll << 31) && (long long)M * K < (1ll << 29) &&
           128ll * N < (1ll << 29);
}

int launch_gemm_s3(const S3Args& a, hipStream_t stream) {
    R4D_REQUIRE(a.A && a.planes && a.C, "gemm_s3: null pointer");
    R4D_REQUIRE(gemm_s3_supported(a.M, a.K, a.N), "gemm_s3: unsupported shape M=%d K=%d N=%d (K %% 32 == 0 wanted)", a.M, a.K, a.N);
    R4D_REQUIRE(a.lda % 4 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.planes % 16) == 0, "gemm_s3: alignment");
    R4D_REQUIRE(a.epilogue < EPI_RESIDUAL || a.resid, "gemm_s3: this epilogue needs the second buffer");
    R4D_REQUIRE(a.epilogue != EPI_SCALE_DIV && a.epilogue != EPI_HALF_PLUS, "gemm_s3: epilogue %d has no instantiation", a.epilogue);
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("R4D_GEMM_S3_TILE"); forced = e ? atoi(e) : -1; }
    if (forced >= 0 && forced < kNumS3) return s3_launch_tile(a, forced, stream);
    int best = 0;
    double best_cost = 1e300;
    for (int t = 0; t < kNumS3; ++t) {
        if (kS3[t].eff <= 0.0) continue;
        const long long blocks = (long long)cdiv(a.M, kS3[t].bm) * cdiv(a.N, kS3[t].bn);
        const double cost = (double)((blocks + 255) / 256) * kS3[t].bm * kS3[t].bn / kS3[t].eff;
        if (cost < best_cost) { best_cost = cost; best = t; }
    }
    return s3_launch_tile(a, best, stream);
}

// C[M,N] = epilogue(A[M,K] . B[N,K]^T) with BOTH operands fp32 and split on the fly (the retrieval scoring GEMM at Q > 64:
// A = normalised queries, B = the normalised pool shard); epilogue EPI_NONE or EPI_HALF_PLUS
bool gemm_s3_f32b_supported(int M, int K, int N) {
    return gemm_s3_supported(M, K, N) && (long long)N * K * 4 < (1ll << 31);
}
int launch_gemm_s3_f32b(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldc, int epilogue, hipStream_t stream) {
    R4D_REQUIRE(A && B && C && gemm_s3_f32b_supported(M, K, N), "gemm_s3_f32b: unsupported shape M=%d K=%d N=%d", M, K, N);
    R4D_REQUIRE(lda % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_s3_f32b: alignment");
    R4D_BRANCH(S3_F32B);
    const int tiles = cdiv(M, 128) * cdiv(N, 256);
    ProfScope prof(PK_GEMM_S3_128x256, 2.0 * (double)M * N * K, stream);
    S3Shape sh;
    sh.M = M; sh.N = N; sh.K = K; sh.lda = lda; sh.ldc = ldc; sh.ldr = ldc; sh.plane_bytes = 0;
    const unsigned short* Bp = reinterpret_cast<const unsigned short*>(B);
    if (epilogue == EPI_HALF_PLUS)
        hipLaunchKernelGGL((gemm_s3_kernel<128, 256, 2, 4, 2, 1, EPI_HALF_PLUS, true>), dim3(tiles), dim3(512), 0, stream, A, Bp, C, nullptr, nullptr, sh);
    else if (epilogue == EPI_NONE)
        hipLaunchKernelGGL((gemm_s3_kernel<128, 256, 2, 4, 2, 1, EPI_NONE, true>), dim3(tiles), dim3(512), 0, stream, A, Bp, C, nullptr, nullptr, sh);
    else { set_error("gemm_s3_f32b: epilogue %d has no instantiation", epilogue); return R4D_ERR_INVALID; }
    R4D_CHECK_LAUNCH("gemm_s3_f32b");
    return R4D_OK;
}

// The scoring GEMM in the pool-scan kernels' arithmetic (SCAN template parameter): S = (q_hat . pool_hat^T + 1) / 2 for Q >= 64, bit for
// bit the scores pool_scan_*_kernel<KW, NG> writes (score.hip), with the pool read Q / 128 times instead of Q / 32 times and every
// element split once per 128 x 256 tile instead of once per wavefront that touches it.  ng = lines per wavefront of the scan variant
// of this d (d == 32 KW NG).  +1 = no instantiation.
int launch_gemm_s3_scan_order(const float* q_hat, const float* pool_hat, float* scores, int Q, int N, int d, int ng, hipStream_t stream) {
    if (!gemm_s3_f32b_supported(Q, d, N) || ng < 2 || ng > 4 || (d / 32) % ng != 0) return 1;      // (ng == 1, d <= 128: the fold after every k-tile spills; the 32-query blocks keep those)
    if ((((uintptr_t)q_hat | (uintptr_t)pool_hat) & 15u) != 0 || d % 4 != 0) return 1;
    R4D_BRANCH(SCAN_TILED);
    const int tiles = cdiv(Q, 128) * cdiv(N, 256);
    S3Shape sh;
    sh.M = Q; sh.N = N; sh.K = d; sh.lda = d; sh.ldc = N; sh.ldr = N; sh.plane_bytes = 0;
    const unsigned short* Bp = reinterpret_cast<const unsigned short*>(pool_hat);
#define S3_SCAN_LAUNCH_(NG_) hipLaunchKernelGGL((gemm_s3_kernel<128, 256, 2, 4, 2, 1, EPI_HALF_PLUS, true, NG_>), dim3(tiles), dim3(512), 0, stream, q_hat, Bp, scores, nullptr, nullptr, sh)
    switch (ng) {
        case 2: S3_SCAN_LAUNCH_(2); break;
        case 3: S3_SCAN_LAUNCH_(3); break;
        default: S3_SCAN_LAUNCH_(4); break;
    }
#undef S3_SCAN_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_s3_scan_order");
    return R4D_OK;
}

int launch_split3_planes(const float* w, int N, int K, long long ld_k, long long ld_n, unsigned short* planes, hipStream_t s) {
    R4D_REQUIRE(w && planes && N >= 1 && K >= 1, "split3_planes: bad arguments");
    R4D_REQUIRE(ld_n == 1 || ld_k == 1, "split3_planes: one of the two strides must be 1");
    hipLaunchKernelGGL(split3_planes_kernel, dim3(cdiv(N, 32), cdiv(K, 32)), dim3(256), 0, s, w, N, K, ld_k, ld_n, planes);
    R4D_CHECK_LAUNCH("split3_planes");
    return R4D_OK;
}

int dbgflag_s3() { return S3_DBG != 0; }

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_split3_planes_bf16(const float* w_d, int32_t K, int32_t N, int32_t transposed, uint16_t* planes_d, void* stream) {
    // transposed == 0: w_d is the reference Conv1D layout [K,N] (in, out); != 0: w_d is [N,K]
    return launch_split3_planes(w_d, N, K, transposed ? 1 : N, transposed ? K : 1, planes_d, (hipStream_t)stream);
}

int r4d_conv1d_s3_f32(const float* x_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d, int32_t M,
                      int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream) {
    R4D_REQUIRE(epilogue >= 0 && epilogue <= 2, "conv1d_s3: epilogue %d not in {0,1,2}", epilogue);
    S3Args a;
    memset(&a, 0, sizeof(a));
    a.A = x_d; a.planes = planes_d; a.C = y_d; a.bias = bias_d; a.resid = residual_d;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
    return launch_gemm_s3(a, (hipStream_t)stream);
}

}  // extern "C"
