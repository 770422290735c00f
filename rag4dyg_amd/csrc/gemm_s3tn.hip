// Weight gradients on the bf16 matrix cores at fp32 accuracy: C[I,J] = sum over the token rows m of X[m,I]^T . dY[m,J]
// (Conv1D under autograd, modeling_utils.py:1267-1271: dW = x^T . dy) -- the "TN" form of gemm_s3.hip: BOTH operands are
// activations, row-major over the CONTRACTED index, so
//   * both tiles (32 tokens x 128 features of X, 32 tokens x 256 features of dY) are split on the fly into three bf16 terms
//     (gemm_s3.hip: x = hi + mid + lo exactly; six partial products per fp32 product) while they are staged;
//   * their LDS images stay token-major -- exactly what the coalesced global rows deliver -- and the k-contiguous MFMA
//     fragments come out of `ds_read_b64_tr_b16`, gfx950's transposing LDS read (per 16 lanes: a 4-token x 16-feature block
//     delivered feature-major: lane i gets feature i of the four tokens), two reads per 8-token fragment.  The 16-byte chunk
//     index of a row is XOR-ed with (token & 3) << 2, which spreads the four token rows of a read over all 64 banks
//     (unswizzled, 256-byte and 512-byte rows put them on the same 16: 4-way conflicts);
//   * the contraction is split over the workgroups (the token rows are 40 - 170 thousand, the outputs 512 x 512 ... 512 x 2048):
//     slice z writes its own fp32 partial, summed in slice order by the caller's reduce launch -- deterministic.
// Tile 128 x 256 (I x J), BK = 32 tokens, 8 wavefronts as 2 x 4 (wave tile 64 x 64), two LDS stages of 72 KB, register-staged
// global loads one iteration ahead; I % 128 == 0 and J % 256 == 0 (every Conv1D of the GPT-2 shapes), rows past M read as zero
// through the buffer range check.
#include <string.h>
#include "common.h"

namespace r4d {

typedef float f32x16t __attribute__((ext_vector_type(16)));
typedef float f32x2t __attribute__((ext_vector_type(2)));
typedef float f32x4t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4t __attribute__((ext_vector_type(4)));
typedef short s16x4t __attribute__((ext_vector_type(4)));
typedef short s16x8t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16_t(float a, float b) {
    const f32x2t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2t));
}
__device__ __forceinline__ void split3_pair_t(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16_t(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pk_bf16_t(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pk_bf16_t(s0, s1);
}

struct TnShape {
    int M;                 // token rows in all
    int I, J, lda, ldb;    // features of X / dY and their row strides (floats)
    int kper;              // token rows per slice (multiple of 32)
};

// `dbp` (nullable): partial column sums of dY per slice, [slices][J] -- the bias gradient db = sum over the token rows of dY falls
// out of the staging of the dY tiles (the workgroups of the first I tile add the values they stage anyway: 16 adds per k-tile
// and thread), summed per thread over its tokens in order, then over the 16 threads of a column chunk in order: a fixed order
__global__ __launch_bounds__(512, 2) void gemm_s3tn_kernel(const float* __restrict__ Xg, const float* __restrict__ Yg,
                                                           float* __restrict__ Cg, const TnShape g, float* __restrict__ dbp) {
    constexpr int BI = 128, BJ = 256, BK = 32, WI = 64, WJ = 64, TI = 2, TJ = 2;
    constexpr int A_ROW = BI * 2, B_ROW = BJ * 2;                     // bytes per token row of one plane image
    constexpr int A_PLANE = BK * A_ROW, B_PLANE = BK * B_ROW;         // 8 KB, 16 KB
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);                    // 72 KB
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tiles_j = g.J / BJ;
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
    const int i0 = ti * BI, j0 = tj * BJ;
    const int m_lo = blockIdx.z * g.kper, m_hi = min(g.M, m_lo + g.kper);
    const int nkt = (m_hi - m_lo + BK - 1) / BK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid >> 2, wj = wid & 3;

    // staging: a thread owns 8 consecutive features of one token: A item (token = tid >> 4, chunk = tid & 15), B items
    // (token = idx >> 5, chunk = idx & 31) for idx = tid, tid + 512.  Rows past M: the buffer range check returns zeros.
    const int a_tok = tid >> 4, a_ch = tid & 15;
    const int a_voff = (a_tok * g.lda + i0 + a_ch * 8) * 4;
    const int a_dst = a_tok * A_ROW + ((a_ch ^ ((a_tok & 3) << 2)) << 4);
    int b_voff[2], b_dst[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int idx = tid + r * 512, tok = idx >> 5, ch = idx & 31;
        b_voff[r] = (tok * g.ldb + j0 + ch * 8) * 4;
        b_dst[r] = tok * B_ROW + ((ch ^ ((tok & 3) << 2)) << 4);
    }
    // descriptors end at row M: whole rows beyond it are out of range (zeros)
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xg), 0, (int)((long long)g.M * g.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yg), 0, (int)((long long)g.M * g.ldb * 4), 0x00020000);
    u32x4t ra[2], rb[2][2];
    const bool do_cs = dbp != nullptr && ti == 0;                     // workgroup-uniform
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    // column sums over the staged dY items of a k-tile that belongs to this slice (the registers hold k-tile KT)
#define TN_COLSUM(KT)                                                                              \
    if (do_cs && (KT) < nkt) {                                                                     \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                            \
            const f32x4t c0_ = __builtin_bit_cast(f32x4t, rb[r][0]), c1_ = __builtin_bit_cast(f32x4t, rb[r][1]); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { cs[e] += c0_[e]; cs[4 + e] += c1_[e]; } \
        }                                                                                          \
    }
#define TN_LOAD(KT)                                                                                \
    {                                                                                              \
        const int so_a_ = (m_lo + (KT) * BK) * g.lda * 4, so_b_ = (m_lo + (KT) * BK) * g.ldb * 4; \
        ra[0] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff, so_a_, 0);                   \
        ra[1] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff + 16, so_a_, 0);              \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                            \
            rb[r][0] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff[r], so_b_, 0);         \
            rb[r][1] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff[r] + 16, so_b_, 0);    \
        }                                                                                          \
    }
#define TN_SPLIT8(R0, R1, H, Mi, L)                                                                \
    {                                                                                              \
        const f32x4t s0_ = __builtin_bit_cast(f32x4t, R0), s1_ = __builtin_bit_cast(f32x4t, R1);  \
        unsigned hh_[4], mm_[4], ll_[4];                                                           \
        split3_pair_t(s0_[0], s0_[1], hh_[0], mm_[0], ll_[0]); split3_pair_t(s0_[2], s0_[3], hh_[1], mm_[1], ll_[1]); \
        split3_pair_t(s1_[0], s1_[1], hh_[2], mm_[2], ll_[2]); split3_pair_t(s1_[2], s1_[3], hh_[3], mm_[3], ll_[3]); \
        H = u32x4t{hh_[0], hh_[1], hh_[2], hh_[3]}; Mi = u32x4t{mm_[0], mm_[1], mm_[2], mm_[3]}; L = u32x4t{ll_[0], ll_[1], ll_[2], ll_[3]}; \
    }
#define TN_STORE(STG)                                                                              \
    {                                                                                              \
        unsigned char* st_ = lds + (STG) * STAGE;                                                  \
        u32x4t h_, m_, l_;                                                                         \
        TN_SPLIT8(ra[0], ra[1], h_, m_, l_)                                                        \
        *reinterpret_cast<u32x4t*>(st_ + a_dst) = h_;                                              \
        *reinterpret_cast<u32x4t*>(st_ + A_PLANE + a_dst) = m_;                                    \
        *reinterpret_cast<u32x4t*>(st_ + 2 * A_PLANE + a_dst) = l_;                                \
        _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                            \
            TN_SPLIT8(rb[r][0], rb[r][1], h_, m_, l_)                                              \
            *reinterpret_cast<u32x4t*>(st_ + 3 * A_PLANE + b_dst[r]) = h_;                         \
            *reinterpret_cast<u32x4t*>(st_ + 3 * A_PLANE + B_PLANE + b_dst[r]) = m_;               \
            *reinterpret_cast<u32x4t*>(st_ + 3 * A_PLANE + 2 * B_PLANE + b_dst[r]) = l_;           \
        }                                                                                          \
    }
    // transposed fragment reads.  Lane l: h = l >> 5 (tokens 8h .. 8h+7 of the k-step), 16-lane group g16 = (l >> 4) & 1
    // (features 16 g16 .. +15 of the 32-wide tile), q = (l & 15) >> 2 (token row of the 4 x 16 block), p = l & 3 (features
    // 4p .. 4p+3 of the group): address = row (16 s + 8 h + 4 u + q), chunk ((F / 8) + 2 g16 + (p >> 1)) ^ (q << 2), byte 8 (p & 1).
    const int fh = lane >> 5, fg = (lane >> 4) & 1, fq = (lane & 15) >> 2, fp = lane & 3;
    int a_foff[TI], b_foff[TJ];
#pragma unroll
    for (int t = 0; t < TI; ++t) {
        const int ft = (wi * WI + t * 32) >> 5;                       // 32-feature tile index inside the 128-wide image
        a_foff[t] = (8 * fh + fq) * A_ROW + ((((ft ^ fq) << 2) | (2 * fg + (fp >> 1))) << 4) + 8 * (fp & 1);
    }
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int ft = (wj * WJ + t * 32) >> 5;
        b_foff[t] = (8 * fh + fq) * B_ROW + ((((ft ^ fq) << 2) | (2 * fg + (fp >> 1))) << 4) + 8 * (fp & 1);
    }
    typedef __attribute__((address_space(3))) s16x4t* lds_tr_t;
    u32x4t fa[2][TI][3], fb[2][TJ][3];
#define TN_TR(PTR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(PTR))
#define TN_FRAG(DST, BASE, ROWB, S)                                                                \
    {                                                                                              \
        const s16x4t lo_ = TN_TR((BASE) + (16 * (S)) * (ROWB)), hi_ = TN_TR((BASE) + (16 * (S) + 4) * (ROWB)); \
        const s16x8t v_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);               \
        DST = __builtin_bit_cast(u32x4t, v_);                                                      \
    }
#define TN_FRAGS(SET, STG, S)                                                                      \
    {                                                                                              \
        const unsigned char* st_ = lds + (STG) * STAGE;                                            \
        _Pragma("unroll") for (int t = 0; t < TI; ++t) TN_FRAG(fa[SET][t][2], st_ + 2 * A_PLANE + a_foff[t], A_ROW, S) \
        _Pragma("unroll") for (int t = 0; t < TJ; ++t) TN_FRAG(fb[SET][t][0], st_ + 3 * A_PLANE + b_foff[t], B_ROW, S) \
        _Pragma("unroll") for (int t = 0; t < TI; ++t) TN_FRAG(fa[SET][t][0], st_ + a_foff[t], A_ROW, S) \
        _Pragma("unroll") for (int t = 0; t < TJ; ++t) TN_FRAG(fb[SET][t][2], st_ + 3 * A_PLANE + 2 * B_PLANE + b_foff[t], B_ROW, S) \
        _Pragma("unroll") for (int t = 0; t < TI; ++t) TN_FRAG(fa[SET][t][1], st_ + A_PLANE + a_foff[t], A_ROW, S) \
        _Pragma("unroll") for (int t = 0; t < TJ; ++t) TN_FRAG(fb[SET][t][1], st_ + 3 * A_PLANE + B_PLANE + b_foff[t], B_ROW, S) \
    }
    f32x16t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define TN_MFMA(A_, B_, I_, J_) \
    acc[I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8t, A_), __builtin_bit_cast(bf16x8t, B_), acc[I_][J_], 0, 0, 0)
#define TN_MFMAS(SET)                                                                              \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][2], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][0], fb[SET][j][2], i, j); \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][1], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][1], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][0], fb[SET][j][1], i, j); \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) TN_MFMA(fa[SET][i][0], fb[SET][j][0], i, j); \
    }
    // iteration kt: registers (k-tile kt+1) -> the other stage, loads of k-tile kt+2 (rows past m_hi of THIS slice are
    // real rows of the next slice -- they must not be added here: the loop bound keeps kt + 1 < nkt for every store that is
    // used, and the slice length is a multiple of BK, so only the global tail relies on the zero fill)
#define TN_ITER(CUR)                                                                               \
    {                                                                                              \
        TN_FRAGS(0, CUR, 0)                                                                        \
        TN_COLSUM(kt + 1)                                                                          \
        TN_STORE((CUR) ^ 1)                                                                        \
        TN_LOAD(kt + 2)                                                                            \
        TN_FRAGS(1, CUR, 1)                                                                        \
        TN_MFMAS(0)                                                                                \
        TN_MFMAS(1)                                                                                \
        __syncthreads();                                                                           \
    }
    TN_LOAD(0)
    TN_COLSUM(0)
    TN_STORE(0)
    TN_LOAD(1)
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nkt; kt += 2) {
        TN_ITER(0)
        { ++kt; TN_ITER(1) }
        --kt;
    }
    if (kt < nkt) TN_ITER(0)
#undef TN_ITER
#undef TN_COLSUM
#undef TN_MFMAS
#undef TN_MFMA
#undef TN_FRAGS
#undef TN_FRAG
#undef TN_TR
#undef TN_STORE
#undef TN_SPLIT8
#undef TN_LOAD
    if (do_cs) {                                                      // thread (token row t16 = tid >> 5, chunk tid & 31): 16 rows per chunk
        float* red = reinterpret_cast<float*>(lds);                  // the last iteration's barrier has passed: the stages are free
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * 256 + (tid & 31) * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
#pragma unroll
            for (int t16 = 0; t16 < 16; ++t16) sum += red[t16 * 256 + tid];
            dbp[(size_t)blockIdx.z * g.J + j0 + tid] = sum;
        }
    }
    // partial tile of slice z (whole tiles only: I % 128 == 0, J % 256 == 0)
    float* C = Cg + (size_t)blockIdx.z * g.I * g.J;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wi * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = j0 + wj * WJ + j * 32 + li;
                C[(size_t)row * g.J + col] = acc[i][j][r];
            }
}

bool gemm_s3tn_supported(int I, int J, int M, int lda, int ldb) {
    return I % 128 == 0 && J % 256 == 0 && M >= 32 && lda % 4 == 0 && ldb % 4 == 0 && (long long)M * lda < (1ll << 29) &&
           (long long)M * ldb < (1ll << 29);
}

// slices: enough workgroups for two rounds of the chip, at least 8 k-tiles each, never more than `max_slices` (the caller's
// scratch was sized for the exact-f32 kernel's split)
int gemm_s3tn_slices(int I, int J, int M, int max_slices) {
    const int tiles = (I / 128) * (J / 256);
    int S = cdiv(512, tiles);
    const int smax = cdiv(M, 256);
    if (S > smax) S = smax;
    if (S > max_slices) S = max_slices;
    return S < 1 ? 1 : S;
}

// C (S == 1) or partial [S][I][J] (S > 1) <- X^T . dY; returns the number of slices that hold rows through *slices_out
int launch_gemm_s3tn(const float* X, const float* dY, float* out, int I, int J, int M, int lda, int ldb, int S, int* slices_out,
                     hipStream_t stream, float* db_partials) {
    R4D_REQUIRE(gemm_s3tn_supported(I, J, M, lda, ldb), "gemm_s3tn: unsupported shape I=%d J=%d M=%d", I, J, M);
    R4D_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_s3tn: 16-byte alignment");
    TnShape sh;
    sh.M = M; sh.I = I; sh.J = J; sh.lda = lda; sh.ldb = ldb;
    sh.kper = cdiv(cdiv(M, S), 32) * 32;
    const int Sx = cdiv(M, sh.kper);
    *slices_out = Sx;
    ProfScope prof(PK_GEMM_S3TN, 2.0 * (double)I * J * M, stream);
    R4D_BRANCH(S3_TN);
    hipLaunchKernelGGL(gemm_s3tn_kernel, dim3((I / 128) * (J / 256), 1, Sx), dim3(512), 0, stream, X, dY, out, sh, db_partials);
    R4D_CHECK_LAUNCH("gemm_s3tn");
    return R4D_OK;
}

}  // namespace r4d
