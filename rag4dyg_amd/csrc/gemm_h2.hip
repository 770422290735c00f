// "f16x2" GEMM: C[M,N] = epilogue(A[M,K] . W^T + bias) on the fp16 matrix cores at fp32 accuracy, THREE matrix instructions per
// fp32 product (gemm_s3.hip's bf16x3 form needs six).
//
// Every fp32 operand x is written as  x = hi + 2^-11 lo' + e  with hi = RN16(x), lo' = RN16((x - hi) 2^11)  (fp16, round to
// nearest even; x - hi is exact in fp32 and at most 2^-11 |x|, so (x - hi) 2^11 <= |x| never overflows and is a NORMAL fp16
// number down to |x - hi| = 3e-8: the second term keeps its 11 significant bits over the whole range instead of sinking into
// fp16's subnormals) and |e| <= 2^-22 |x|, 2^-24 |x| in the mean -- the unit roundoff of fp32 itself.  A product a.b is
//     hi_a hi_b  +  2^-11 (lo'_a hi_b + hi_a lo'_b)          (dropped: 2^-22 lo'_a lo'_b <= 2^-22 |a b|)
// evaluated as three v_mfma_f32_32x32x16_f16 with fp32 accumulators: the first term into one accumulator set, the two cross
// terms (which carry the factor 2^11) into a SECOND set, joined once at the end as acc0 + 2^-11 acc1 (one rounding; the power
// of two is exact).  fp16 x fp16 is exact in fp32, so the only roundings are the fp32 accumulations -- three per 16 products
// here, six with bf16x3, eight with the exact-f32 instruction (v_mfma_f32_32x32x2_f32: one per 2 products): measured against
// float64 (tools/s3_acceptance.py, profiles/r04_h2_acceptance.md) the error is no larger than either of them.
// Range: fp16 tops out at 65504 and its normal numbers end at 6.1e-5.  Activations are pre-scaled by 2^-2 (exact; undone in the
// epilogue): the A operand may reach 2^18 = 2.6e5 before hi overflows to inf (and the result to NaN: loud, not silent), and
// every element of magnitude >= 2.4e-4 keeps the full two-term precision; below that hi is an fp16 subnormal and the element's
// ABSOLUTE error stops shrinking at 6e-11 (2^-36 x 4) -- fp32's own roundoff for anything in the same dot product that is
// larger than 1e-3.  A tensor whose rows are ALL smaller than that (tools/h2_check.py, "tiny 1e-4": error 1e-6 instead of
// 3e-7) is outside the range this form is meant for; bf16x3 (gemm_s3.hip) has fp32's full exponent range.  The static weights
// are checked when their planes are made (|w| < 6e4, ops.split2_planes) and keep the bf16x3 planes otherwise.
//
// Operands: A is the fp32 activation [M,K], split on the fly while its tile is staged; W is a STATIC weight, split once per
// checkpoint into fp16 lines [N][K/32][2][32] -- 32 hi and 32 lo' values per 128 bytes -- (r4d_split2_planes_f16).  Tile structure, LDS image (64-byte rows,
// 16-byte chunk index XOR (row >> 2) & 3), buffer loads, register-staged pipeline and the pinned MFMA / DS / VMEM interleaving
// are those of gemm_s3.hip with two planes per operand instead of three: a stage of the 128 x 256 tile is 48 KB, so THREE
// stages fit (144 KB) where bf16x3 had room for two.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "h2.h"

#ifndef H2_DBG
#define H2_DBG 0   // tuning aid (tools/kc_ablate.sh gemm_h2.hip H2_DBG n): bit 0 drops the fragment reads, bit 1 the LDS staging stores (and the split), bit 2 the barrier, bit 3 the global loads, bit 5 the epilogue, bit 6 the MFMAs
#endif

namespace r4d {

typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef float f32x2h __attribute__((ext_vector_type(2)));
typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2h __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2h __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x2h gelu_new_h2(f32x2h x) {            // the epilogue of gemm_s3.hip / gemm_f32_kc.hip, same instructions
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    const f32x2h a = x * x * k1 + k0;
    const f32x2h w = x * a;
    f32x2h e;
    e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
    e = e + 1.0f;
    f32x2h r;
    r.x = __builtin_amdgcn_rcpf(e.x); r.y = __builtin_amdgcn_rcpf(e.y);
    return x * r;
}
__device__ __forceinline__ float gelu_new_h2_1(float x) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

struct H2Shape {
    int M, N, K, lda, ldc, ldr;
    int plane_bytes;          // N * K * 2: half the size of W's fp16 lines
};

// ---------------------------------------------------------------------------------------------- weight planes
// w element (n, k) at w[k * ld_k + n * ld_n]  (reference Conv1D layout [K,N]: ld_k = N, ld_n = 1; a [N,K] copy: 1, K)
// Plane layout [N][K/32][2][32] fp16: the 32 hi values of row n, k-tile kt and its 32 lo' values are ONE 128-byte line -- the
// unit the vector L1 fetches from L2 -- so a k-tile of the GEMM reads whole lines (two planes [2][N][K], the round-4 first
// form, made every k-tile fetch the 64-byte halves of twice as many lines, the other halves evicted before the next k-tile).
__global__ __launch_bounds__(256) void split2_planes_kernel(const float* __restrict__ w, int N, int K, long long ld_k,
                                                            long long ld_n, unsigned short* __restrict__ planes) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool n_fast = ld_n == 1;
    for (int r = ty; r < 32; r += 8) {
        const int n = n_fast ? n0 + tx : n0 + r, k = n_fast ? k0 + r : k0 + tx;
        const float v = (n < N && k < K) ? w[(long long)k * ld_k + (long long)n * ld_n] : 0.f;
        if (n_fast) tile[tx][r] = v; else tile[r][tx] = v;            // tile[n][k]
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                                // write along k
        const int n = n0 + r;
        if (n >= N) continue;
        unsigned h, l;
        split2_pair<false>(tile[r][tx], 0.f, h, l);
        const size_t o = ((size_t)n * (K >> 5) + blockIdx.y) * 64 + tx;
        planes[o] = (unsigned short)(h & 0xffffu);
        planes[o + 32] = (unsigned short)(l & 0xffffu);
    }
}

// ---------------------------------------------------------------------------------------------- the GEMM
template <int BM, int BN, int WGM, int WGN, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4) void gemm_h2_kernel(
    const float* __restrict__ Ag, const unsigned short* __restrict__ Bp, float* __restrict__ Cg,
    const float* __restrict__ biasg, const float* __restrict__ residg, const H2Shape g) {
    constexpr int BK = 32;
    constexpr int NTHREADS = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    // Staging moves WHOLE 128-byte lines per wave-instruction: a k-tile of an A row is one line of fp32 (32 k), a k-tile of a W
    // row one line of fp16 (32 hi + 32 lo'); eight consecutive lanes take the eight 16-byte pieces of one line, a wave eight lines.
    constexpr int NLA = BM * 8 / NTHREADS, NLB = BN * 8 / NTHREADS;   // (row, 16-byte piece) items per thread
    constexpr int A_PLANE = BM * 4, B_PLANE = BN * 4;                 // uint4 units (an image row = 32 k of one plane = 4 chunks of 16 bytes)
    constexpr int B_SKEW = 4;                                         // W's lo' plane starts 64 bytes past a multiple of 128: the eight lanes of a line store its hi and lo' halves in one ds_write_b128 group, and those must not share banks
    constexpr int B_PLANE1 = B_PLANE + B_SKEW;
    constexpr int STAGE = 2 * (A_PLANE + B_PLANE) + B_SKEW;
    constexpr int NBUF = 3;                                           // LDS stages (48 KB each at 128 x 256)
    static_assert(NLA >= 1 && NLB >= 1 && NLA <= 4 && NLB <= 8 && TM >= 1 && TN >= 1, "tile");
    __shared__ u32x4h lds[NBUF * STAGE];

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

    int a_off[NLA], b_off[NLB], a_dst[NLA], b_dst[NLB];
#pragma unroll
    for (int i = 0; i < NLA; ++i) {                                  // piece c of the line = k 4c .. 4c+3 -> half (c & 1) of image chunk c >> 1
        const int idx = tid + i * NTHREADS, row = idx >> 3, c = idx & 7;
        a_off[i] = (min(m0 + row, g.M - 1) * g.lda + c * 4) * 4;
        a_dst[i] = (row * 4 + ((c >> 1) ^ ((row >> 2) & 3))) * 2 + (c & 1);      // 8-byte units
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {                                  // piece c of the line = chunk c & 3 of plane c >> 2
        const int idx = tid + i * NTHREADS, row = idx >> 3, c = idx & 7;
        b_off[i] = (min(n0 + row, g.N - 1) * g.K * 2 + c * 8) * 2;
        b_dst[i] = (c >> 2) * B_PLANE1 + row * 4 + ((c & 3) ^ ((row >> 2) & 3));
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ag), 0, (int)(((long long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Bp), 0, 2 * g.plane_bytes, 0x00020000);

    u32x4h ra[NLA], rb[NLB];
#define H2_LOAD(KT)                                                                                \
    {                                                                                              \
        const int kt_ = min((KT), nkt - 1);                                                        \
        _Pragma("unroll") for (int i = 0; i < NLB; ++i)                                            \
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_off[i], kt_ * 128, 0);         \
        _Pragma("unroll") for (int i = 0; i < NLA; ++i)                                            \
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_off[i], kt_ * 128, 0);         \
    }
#define H2_STORE(STG)                                                                              \
    {                                                                                              \
        u32x4h* sa_ = lds + (STG) * STAGE;                                                         \
        u32x2h* sa2_ = reinterpret_cast<u32x2h*>(sa_);                                             \
        u32x4h* sb_ = sa_ + 2 * A_PLANE;                                                           \
        _Pragma("unroll") for (int i = 0; i < NLB; ++i) sb_[b_dst[i]] = rb[i];                     \
        _Pragma("unroll") for (int i = 0; i < NLA; ++i) {                                          \
            /* (cast the WHOLE vector: __builtin_bit_cast on an ext-vector element reads element 0) */ \
            const f32x4h src_ = __builtin_bit_cast(f32x4h, ra[i]);                                 \
            u32x2h h_, l_;                                                                         \
            unsigned hh_, ll_;                                                                     \
            split2_pair<true>(src_[0], src_[1], hh_, ll_);                                         \
            h_[0] = hh_; l_[0] = ll_;                                                              \
            split2_pair<true>(src_[2], src_[3], hh_, ll_);                                         \
            h_[1] = hh_; l_[1] = ll_;                                                              \
            sa2_[a_dst[i]] = h_; sa2_[2 * A_PLANE + a_dst[i]] = l_;                                \
        }                                                                                          \
    }

    // fragment addresses: lane (li, lh), k-step s -> chunk 2s + lh of row li (+ 32 per tile)
    const int fq = (li >> 2) & 3;
    const int f_off0 = li * 4 + ((0 + lh) ^ fq), f_off1 = li * 4 + ((2 + lh) ^ fq);
    const int fa_base = wm * WM * 4, fb_base = 2 * A_PLANE + wn * WN * 4;

    f32x16h acc0[TM][TN], acc1[TM][TN];                  // hi.hi  /  the two cross terms (factor 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

#define H2_MFMA(ACC, A_, B_, I_, J_) \
    if (!(H2_DBG & 64)) ACC[I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8h, A_), __builtin_bit_cast(f16x8h, B_), ACC[I_][J_], 0, 0, 0); \
    else { asm volatile("" :: "v"(A_), "v"(B_)); }
    u32x4h fa[2][TM][2], fb[2][TN][2];                   // two fragment sets: the reads of k-step s+1 travel under the MFMAs of k-step s
#define H2_FRAGS(SET, STG, S)                                                                      \
    {                                                                                              \
        const u32x4h* st_ = lds + (STG) * STAGE;                                                   \
        const int fo_ = (S) ? f_off1 : f_off0;                                                     \
        /* in the order the MFMAs want them: lo(A) . hi(B) first */                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][1] = (H2_DBG & 1) ? dbg_frag : st_[fa_base + 1 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][0] = (H2_DBG & 1) ? dbg_frag : st_[fb_base + 0 * B_PLANE + j * 128 + fo_]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][0] = (H2_DBG & 1) ? dbg_frag : st_[fa_base + 0 * A_PLANE + i * 128 + fo_]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][1] = (H2_DBG & 1) ? dbg_frag : st_[fb_base + B_PLANE1 + j * 128 + fo_]; \
    }
    /* consecutive MFMAs go to different accumulators; the two writes of one acc1 tile are TM TN instructions apart */
#define H2_MFMAS(SET)                                                                              \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2_MFMA(acc1, fa[SET][i][1], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2_MFMA(acc0, fa[SET][i][0], fb[SET][j][0], i, j); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2_MFMA(acc1, fa[SET][i][0], fb[SET][j][1], i, j); \
    }

    u32x4h dbg_frag = {(unsigned)tid, 0x3c003c00u, 0x3c003c00u, (unsigned)lane};   // (ablation builds only)
    if (H2_DBG & 1) asm volatile("" : "+v"(dbg_frag));
    // prologue: k-tiles 0 and 1 into stages 0 and 1, k-tile 2 into the staging registers, k-step 0 of k-tile 0 into fragment set 0
    H2_LOAD(0)
    H2_STORE(0)
    H2_LOAD(1)
    H2_STORE(1)
    H2_LOAD(2)
    __syncthreads();
    H2_FRAGS(0, 0, 0)

    // Three LDS stages, ONE barrier per k-tile, every fragment read under MFMAs.  Iteration kt enters with fragment set 0 = k-step 0
    // of stage CUR = kt % 3 and the staging registers = k-tile kt + 2 (requested during iteration kt - 1):
    //   k-step 0's MFMAs with, between them, the reads of k-step 1 (set 1) and, item by item, the LDS store of a staged piece of
    //   k-tile kt + 2 into stage WR = (kt + 2) % 3 (read last in iteration kt - 1, every wave is past that barrier; the A pieces
    //   are split on the way) FOLLOWED AT ONCE BY THE GLOBAL LOAD of the same piece of k-tile kt + 3 into the registers it just
    //   left: a load is consumed a whole iteration (~ 22 MFMAs, 700 cycles) after its issue.  (First form of round 4: the loads
    //   in the second half, consumed 7 MFMAs later, an s_waitcnt vmcnt at the head of every k-tile.  Measured: the same time --
    //   the launch is held by the clock under fp16 MFMA load, tools/h2_power_probe.py -- but this is the schedule that does
    //   not depend on the L2 being fast);
    //   k-step 1's MFMAs with the reads of k-step 0 of stage NXT = (kt + 1) % 3 (set 0 again) -- stored during iteration
    //   kt - 1, i.e. in front of the same barrier;
    //   barrier.
    constexpr int NMF = 3 * TM * TN, NFR = 2 * (TM + TN);
    static_assert(NMF >= NFR && NMF >= NLA + NLB, "interleave");
#define H2_ITER(CUR, NXT, WR)                                                                      \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(H2_DBG & 2)) H2_STORE(WR)                                                            \
        if (!(H2_DBG & 8)) H2_LOAD(kt + 3)                                                         \
        H2_FRAGS(1, CUR, 1)                                                                        \
        H2_MFMAS(0)                                                                                \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
            if (m_ < NLB) {                                  /* a W piece: store, reload */        \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
            } else if (m_ < NLB + NLA) {                     /* an A piece: split, store, reload */ \
                __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);                                \
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);                                 \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                 \
            }                                                                                      \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        H2_FRAGS(0, NXT, 0)                                                                        \
        H2_MFMAS(1)                                                                                \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (!(H2_DBG & 4)) __syncthreads();                                                        \
    }
    // (past the last k-tile the loads are clamped to it and the stores / reads touch stages nobody consumes: harmless, and the
    //  loop body stays free of conditions)
    int kt = 0;
    for (; kt + 2 < nkt; kt += 3) {                                   // compile-time stages
        H2_ITER(0, 1, 2)
        { ++kt; H2_ITER(1, 2, 0) }
        { ++kt; H2_ITER(2, 0, 1) }
        kt -= 2;
    }
    if (kt < nkt) {                                                   // kt is a multiple of 3 here: one or two k-tiles left
        H2_ITER(0, 1, 2)
        if (kt + 1 < nkt) { ++kt; H2_ITER(1, 2, 0) }
    }
#undef H2_ITER
#undef H2_FRAGS
#undef H2_MFMAS
#undef H2_MFMA
#undef H2_STORE
#undef H2_LOAD

#if H2_DBG & 32
    {   // ablation: no epilogue at all (one conditional store keeps the accumulators alive)
        float ssum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += acc0[i][j][r] + acc1[i][j][r];
        if (ssum == 12345.678f) Cg[0] = ssum;
        return;
    }
#endif
    // epilogue: gemm_s3.hip's (C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)), instruction for instruction, on
    // value = (acc0 + 2^-11 acc1) * unscale.  A form with the operands of the MFMA swapped -- the accumulator then holds C^T and a
    // lane stores 16 bytes per instruction, a quarter of the store instructions -- was built in round 4 and REMOVED: no faster
    // (c_attn 442 vs 430 us) and its GELU variant gave run-to-run different values in lanes 12-15 / 28-31 of each half-wave at some
    // shapes (scalar or packed arithmetic, with or without wait states; tools/h2_check.py now repeats every launch three times).
    constexpr float UNS = H2_A_UNSCALE;
    constexpr bool USES_R = EPI == EPI_RESIDUAL || EPI == EPI_GELU_KEEP;      // the second buffer is read (residual) or written (training forward: the pre-activation)
    float* __restrict__ C = Cg;
    const bool interior = (m0 + BM <= g.M) & (n0 + BN <= g.N);       // wave-uniform
    if (interior) {
        const int lane_c = ((wm * WM + 4 * lh) * g.ldc + wn * WN + li) * 4;
        const int lane_r = ((wm * WM + 4 * lh) * g.ldr + wn * WN + li) * 4;
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            C + (long long)m0 * g.ldc + n0, 0, ((BM - 1) * g.ldc + BN) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(USES_R ? residg + (long long)m0 * g.ldr + n0 : Ag), 0,
            USES_R ? ((BM - 1) * g.ldr + BN) * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bias = biasg ? biasg[n0 + wn * WN + j * 32 + li] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float res[16];
                if (EPI == EPI_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            r_rsrc, lane_r, ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0));
                }
#pragma unroll
                for (int r2 = 0; r2 < 16; r2 += 2) {
                    f32x2h v2 = {__builtin_fmaf(acc1[i][j][r2], H2_LO_UNSCALE, acc0[i][j][r2]) * UNS + bias,
                                 __builtin_fmaf(acc1[i][j][r2 + 1], H2_LO_UNSCALE, acc0[i][j][r2 + 1]) * UNS + bias};
                    if (EPI == EPI_GELU_KEEP) {                       // training forward: the pre-activation goes to the second buffer
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int r = r2 + h2;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, h2 ? v2.y : v2.x), r_rsrc, lane_r,
                                                                  ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0);
                        }
                    }
                    if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) v2 = gelu_new_h2(v2);
                    else if (EPI == EPI_RESIDUAL) { v2.x += res[r2]; v2.y += res[r2 + 1]; }
                    else if (EPI == EPI_HALF_PLUS) { v2.x = (v2.x + 1.0f) / 2.0f; v2.y = (v2.y + 1.0f) / 2.0f; }     // train_retriever.py:438
                    const float vx = v2.x, vy = v2.y;     // (copies first: __builtin_bit_cast on an ext-vector ELEMENT reads element 0)
                    unsigned o2[2] = {__builtin_bit_cast(unsigned int, vx), __builtin_bit_cast(unsigned int, vy)};
                    if (EPI == EPI_H2WORDS) h2_words<true>(v2.x, v2.y, o2[0], o2[1]);     // C is the uint32 word image of the result (attention_h2.hip)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = r2 + h2;
                        __builtin_amdgcn_raw_buffer_store_b32(o2[h2], c_rsrc, lane_c,
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = __builtin_fmaf(acc1[i][j][r], H2_LO_UNSCALE, acc0[i][j][r]) * UNS + bias;
                if (EPI == EPI_GELU_KEEP && row < g.M && col_ok) const_cast<float*>(residg)[(long long)row * g.ldr + col] = v;
                if (EPI == EPI_GELU || EPI == EPI_GELU_KEEP) v = gelu_new_h2_1(v);
                else if (EPI == EPI_RESIDUAL) v += residg[(long long)min(row, g.M - 1) * g.ldr + colc];
                else if (EPI == EPI_HALF_PLUS) v = (v + 1.0f) / 2.0f;
                else if (EPI == EPI_H2WORDS) { unsigned w0, w1; h2_words<true>(v, 0.f, w0, w1); v = __builtin_bit_cast(float, w0); }
                if (row < g.M && col_ok) C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
template <int BM, int BN, int WGM, int WGN>
static int launch_h2(const S3Args& a, int cls, hipStream_t stream) {
    const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
    ProfScope prof(cls, 2.0 * (double)a.M * a.N * a.K, stream);
    H2Shape sh;
    sh.M = a.M; sh.N = a.N; sh.K = a.K; sh.lda = a.lda; sh.ldc = a.ldc; sh.ldr = a.ldr; sh.plane_bytes = a.N * a.K * 2;
#define H2_LAUNCH_(E)                                                                              \
    hipLaunchKernelGGL((gemm_h2_kernel<BM, BN, WGM, WGN, E>), dim3(tiles), dim3(64 * WGM * WGN), 0, stream, a.A, \
                       a.planes, a.C, a.bias, a.resid, sh)
    switch (a.epilogue) {
        case EPI_NONE: H2_LAUNCH_(EPI_NONE); break;
        case EPI_GELU: H2_LAUNCH_(EPI_GELU); break;
        case EPI_RESIDUAL: H2_LAUNCH_(EPI_RESIDUAL); break;
        case EPI_H2WORDS: H2_LAUNCH_(EPI_H2WORDS); break;
        case EPI_GELU_KEEP: H2_LAUNCH_(EPI_GELU_KEEP); break;
        default: set_error("gemm_h2: epilogue %d has no instantiation", a.epilogue); return R4D_ERR_INVALID;
    }
#undef H2_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_h2");
    return R4D_OK;
}

bool gemm_h2_supported(int M, int K, int N) {
    return M >= 1 && K >= 32 && K % 32 == 0 && N >= 1 && (long long)N * K * 4 < (1ll << 31) && (long long)M * K < (1ll << 29) &&
           128ll * N < (1ll << 29);
}

int launch_gemm_h2(const S3Args& a, hipStream_t stream) {
    R4D_REQUIRE(a.A && a.planes && a.C, "gemm_h2: null pointer");
    R4D_REQUIRE(gemm_h2_supported(a.M, a.K, a.N), "gemm_h2: unsupported shape M=%d K=%d N=%d (K %% 32 == 0 wanted)", a.M, a.K, a.N);
    R4D_REQUIRE(a.lda % 4 == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.planes % 16) == 0, "gemm_h2: alignment");
    R4D_REQUIRE((a.epilogue != EPI_RESIDUAL && a.epilogue != EPI_GELU_KEEP) || a.resid,
                "gemm_h2: this epilogue needs the second buffer");
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("R4D_GEMM_H2_TILE"); forced = e ? atoi(e) : -1; }
    int t = forced;
    if (t < 0 || t > 1) {                                             // fewest tile waves; the wide tile wins ties
        const long long b0 = (long long)cdiv(a.M, 128) * cdiv(a.N, 256), b1 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128);
        const double c0 = (double)((b0 + 255) / 256) * 128 * 256, c1 = (double)((b1 + 255) / 256) * 128 * 128 / 0.9;
        t = c1 < c0 ? 1 : 0;
    }
    if (t == 0) { R4D_BRANCH(H2_128x256); return launch_h2<128, 256, 2, 4>(a, PK_GEMM_H2_128x256, stream); }
    R4D_BRANCH(H2_128x128);
    return launch_h2<128, 128, 2, 4>(a, PK_GEMM_H2_128x128, stream);
}

int launch_split2_planes(const float* w, int N, int K, long long ld_k, long long ld_n, unsigned short* planes, hipStream_t s) {
    R4D_REQUIRE(w && planes && N >= 1 && K >= 1, "split2_planes: bad arguments");
    R4D_REQUIRE(ld_n == 1 || ld_k == 1, "split2_planes: one of the two strides must be 1");
    R4D_REQUIRE(K % 32 == 0, "split2_planes: K = %d is not a multiple of 32 (the plane layout is per 32-k line)", K);
    hipLaunchKernelGGL(split2_planes_kernel, dim3(cdiv(N, 32), cdiv(K, 32)), dim3(256), 0, s, w, N, K, ld_k, ld_n, planes);
    R4D_CHECK_LAUNCH("split2_planes");
    return R4D_OK;
}

int dbgflag_h2() { return H2_DBG != 0; }

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_split2_planes_f16(const float* w_d, int32_t K, int32_t N, int32_t transposed, uint16_t* planes_d, void* stream) {
    // transposed == 0: w_d is the reference Conv1D layout [K,N] (in, out); != 0: w_d is [N,K]
    return launch_split2_planes(w_d, N, K, transposed ? 1 : N, transposed ? K : 1, planes_d, (hipStream_t)stream);
}

int r4d_conv1d_h2_f32(const float* x_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d, int32_t M,
                      int32_t K, int32_t N, int32_t epilogue, float* y_d, void* stream) {
    R4D_REQUIRE(epilogue >= 0 && epilogue <= 2, "conv1d_h2: epilogue %d not in {0,1,2}", epilogue);
    S3Args a;
    memset(&a, 0, sizeof(a));
    a.A = x_d; a.planes = planes_d; a.C = y_d; a.bias = bias_d; a.resid = residual_d;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
    return launch_gemm_h2(a, (hipStream_t)stream);
}

}  // extern "C"
