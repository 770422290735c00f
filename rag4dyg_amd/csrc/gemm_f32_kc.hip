// fp32 MFMA GEMM, "k-contiguous" form: C[M,N] = epilogue(A[M,K] . B[N,K]^T + bias) with BOTH operands stored with
// K contiguous (B = W^T for the Conv1D layers -- the host keeps a transposed copy of every static weight; Q.K^T, the
// tied lm_head and the Q > 64 pool scan are k-contiguous by nature).  Preconditions (checked by the dispatcher in
// gemm_f32.hip, which otherwise uses that file's kernel): K % 32 == 0, 16-byte aligned rows, operands < 2 GiB.
//
// The design rule, measured with tools/mfma_peak.hip and tools/kc_ablate.sh on gfx950: plain VALU and LDS
// instructions do NOT hide under v_mfma -- each one takes ~3-5 cycles of the SIMD's issue port away from the matrix
// pipe (1 v_add per MFMA: 156 -> 148 TF; 4: 132 TF) -- so an MFMA-bound loop is priced by its non-MFMA instruction
// count.  Hence:
//   * both tiles go global -> LDS as 16-byte rows with NO transposition (ds_write_b128, row stride BK+4 floats:
//     the 16-lane groups of a b128 access land on 16 distinct 16-byte slots, conflict-free);
//   * a lane's MFMA operands are read along k: ONE ds_read_b128 per operand tile feeds FOUR v_mfma_f32_32x32x2_f32
//     (k permuted identically on both operands: component c of the float4 of lane half h is k = 8g + 4h + c);
//   * NO masks and NO address arithmetic in the k-loop: out-of-range rows are clamped (their products only reach
//     accumulator rows the epilogue never stores), loads are buffer_load_dwordx4 with a loop-invariant lane offset
//     and the k-tile in the SCALAR offset;
//   * three LDS stages, one barrier per k-tile, the next tile's first fragments read before the barrier, global
//     loads in flight for a whole iteration (see the pipeline comment in the kernel).
// Per wave and BK = 16 k-tile that leaves 16 MFMAs + 6 ds_read + 2 ds_write + 2 buffer_load and NO VALU.
// Measured (MI355X, 157.3 TF peak): 8192^3 148 TF (vendor hipBLASLt 154.6); M=35456 K=512 N=1536 137 TF (the
// previous kernel 102; vendor 141).  XCD-aware grouped tile order and fused bias / gelu_new / residual / scale
// epilogues as in gemm_f32.hip.
#include <stdlib.h>
#include "common.h"

#ifndef KC_DBG
#define KC_DBG 0   // tuning aid (tools/kc_ablate.sh): bit 0 drops the fragment reads, bit 1 the staging, bit 2 the barrier, bit 3 only the global loads, bit 4 the C stores, bit 5 the whole epilogue
#endif
#if KC_DBG & 2
#define KC_DBG_STAGING(ST, LD)
#elif KC_DBG & 8
#define KC_DBG_STAGING(ST, LD) ST
#else
#define KC_DBG_STAGING(ST, LD) ST LD
#endif
#if KC_DBG & 1
#define KC_DBG_FRAGS(X)
#else
#define KC_DBG_FRAGS(X) X
#endif
#if KC_DBG & 4
#define KC_DBG_BARRIER(X)
#else
#define KC_DBG_BARRIER(X) X
#endif

namespace r4d {

typedef float f32x16k __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_new_kc(float x) {
    // gelu_new(x) = 0.5x(1+tanh(u)), u = sqrt(2/pi)(x+0.044715x^3)  -- modeling_gpt2.py:25,206.
    // Algebraically 0.5(1+tanh(u)) = 1/(1+exp(-2u)) = 1/(1+exp2(x*(k0 + k1*x^2))) with k0 = -2 sqrt(2/pi) log2(e),
    // k1 = 0.044715 k0: mul, fma, mul, v_exp_f32, add, v_rcp_f32, mul -- every epilogue VALU instruction is taken from
    // the MFMA issue slots of the co-resident workgroup, the ocml tanhf form (~40) cost 15 % of a c_fc tile.
    // |error| < 3e-7 |x| (checked against the oracle at 1e-5 relative in tests/test_gpu_ops.py).
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

typedef float f32x2k __attribute__((ext_vector_type(2)));
// two outputs at a time: the polynomial part as packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 on gfx950),
// only exp2 and rcp stay scalar -- 4.5 instead of 7 instructions per element
__device__ __forceinline__ f32x2k gelu_new_kc2(f32x2k x) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    const f32x2k a = x * x * k1 + k0;
    const f32x2k w = x * a;
    f32x2k e;
    e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
    e = e + 1.0f;
    f32x2k r;
    r.x = __builtin_amdgcn_rcpf(e.x); r.y = __builtin_amdgcn_rcpf(e.y);
    return x * r;
}

struct KcShape {
    int M, N, K, lda, ldb, ldc, ldr, nb1, epilogue, causal;
    long long sA0, sA1, sB0, sB1, sC0, sC1;
    float scale_div;
};

template <int BM, int BN, int BK, int WGM, int WGN, int MINW, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void gemm_f32_kc_kernel(
    const float* __restrict__ Ag, const float* __restrict__ Bg, float* __restrict__ Cg,
    const float* __restrict__ biasg, const float* __restrict__ residg, const KcShape g) {
    constexpr int NTHREADS = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int LDS_ROW = BK + 4;                                  // floats; 144 B at BK = 32, 80 B at BK = 16
    constexpr int KV = BK / 4;                                       // float4 per tile row
    constexpr int NLA = BM * KV / NTHREADS, NLB = BN * KV / NTHREADS;
    constexpr int NS = BK / 8;                                       // MFMA steps (8 k each) per k-tile
    constexpr int NBUF = 3;
    constexpr int A_TILE = BM * LDS_ROW, B_TILE = BN * LDS_ROW, STAGE = A_TILE + B_TILE;
    static_assert(NLA >= 1 && NLB >= 1 && TM >= 1 && TN >= 1 && (NS == 2 || NS == 4), "tile");
    __shared__ __attribute__((aligned(16))) float lds[NBUF * STAGE];   // stage b: A image at b*STAGE, B image after it

    // XCD-aware grouped tile order (see gemm_f32.hip)
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
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
    const int nkt = (g.K + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, lh = lane >> 5;

    // Staging coordinates.  NO validity masks anywhere in the k-loop: rows past M (or N) are CLAMPED to the last valid
    // row, so the duplicate data only reaches accumulator rows (columns) the epilogue never stores, and K % BK == 0 is
    // a precondition of this kernel (the dispatcher sends other K to gemm_f32.hip).  Every plain VALU instruction in
    // the loop costs the matrix pipe ~4.5 cycles (tools/mfma_peak.hip: VALU and MFMA share the issue port), so the
    // per-iteration address work is scalar: 32-bit byte offsets per lane + a uniform base that advances BK floats.
    static_assert(NTHREADS % KV == 0, "every staged float4 of a thread sits at the same k offset");
    const int st_k = 4 * (tid % KV), st_row = tid / KV;
    constexpr int ROWS_PER_PASS = NTHREADS / KV;
    int a_off[NLA], b_off[NLB];                                      // byte offsets (voffset of the buffer loads)
#pragma unroll
    for (int r = 0; r < NLA; ++r)
        a_off[r] = (min(m0 + st_row + r * ROWS_PER_PASS, g.M - 1) * g.lda + st_k) * 4;
#pragma unroll
    for (int r = 0; r < NLB; ++r)
        b_off[r] = (min(n0 + st_row + r * ROWS_PER_PASS, g.N - 1) * g.ldb + st_k) * 4;
    const int st_dst = st_row * LDS_ROW + st_k;
    // Pipeline (three LDS stages, one register stage, ONE barrier per k-tile and nothing latency-bound next to it):
    //   iteration kt:  W(kt+2): registers (global loads issued one iteration ago, landed) -> LDS stage (kt+2)%3
    //                  G(kt+3): global loads into the same registers, in flight for a whole iteration
    //                  C(kt)  : MFMA steps on stage kt%3; fragment reads run one step ahead, and the LAST step's
    //                           look-ahead already reads step 0 of k-tile kt+1 from stage (kt+1)%3 -- legal before the
    //                           barrier because that stage was written in iteration kt-1 and published by ITS barrier
    //                  barrier: publishes W(kt+2); stage kt%3 is free for W(kt+3) next iteration
    // With two stages the LDS store, the barrier and the first fragment read of the next tile sit back-to-back on
    // the critical path (~1000 cycles per 4096 MFMA cycles: measured 128 TF asymptote = 81 % with either LDS layout).
    // The loads are UNCONDITIONAL (k-tile index clamped; the tail re-reads the last tile from L2 into stages nobody
    // reads) so the loop body is branch-free and the compiler's vmcnt bookkeeping stays exact.
    // (named registers, not arrays: a loop-carried float4 array that is only copied gets promoted to LDS by hipcc)
    // Buffer loads: address = descriptor base + per-lane 32-bit voffset (loop-invariant) + SCALAR soffset (the k-tile),
    // so the loop carries no address VALU at all (global_load needs a 64-bit v_lshl_add_u64 per load: the zext of the
    // lane offset is hoisted out of the loop and the saddr form is no longer matched).
    static_assert(NLA <= 4 && NLB <= 4, "staging registers");
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A), 0, (int)(((long long)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(B), 0, (int)(((long long)(g.N - 1) * g.ldb + g.K) * 4), 0x00020000);
    u32x4k ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define KC_LD_(RSRC, OFF, SOFF) __builtin_amdgcn_raw_buffer_load_b128(RSRC, OFF, SOFF, 0)
#define KC_LOAD_INTO(A0, A1, A2, A3, B0, B1, B2, B3, SOFF)                                         \
    {                                                                                              \
        A0 = KC_LD_(a_rsrc, a_off[0], SOFF);                                                       \
        if (NLA > 1) A1 = KC_LD_(a_rsrc, a_off[NLA > 1 ? 1 : 0], SOFF);                            \
        if (NLA > 2) A2 = KC_LD_(a_rsrc, a_off[NLA > 2 ? 2 : 0], SOFF);                            \
        if (NLA > 3) A3 = KC_LD_(a_rsrc, a_off[NLA > 3 ? 3 : 0], SOFF);                            \
        B0 = KC_LD_(b_rsrc, b_off[0], SOFF);                                                       \
        if (NLB > 1) B1 = KC_LD_(b_rsrc, b_off[NLB > 1 ? 1 : 0], SOFF);                            \
        if (NLB > 2) B2 = KC_LD_(b_rsrc, b_off[NLB > 2 ? 2 : 0], SOFF);                            \
        if (NLB > 3) B3 = KC_LD_(b_rsrc, b_off[NLB > 3 ? 3 : 0], SOFF);                            \
    }
#define KC_ST_(PTR, V) *reinterpret_cast<u32x4k*>(PTR) = V
#define KC_STORE_FROM(A0, A1, A2, A3, B0, B1, B2, B3, STG)                                         \
    {                                                                                              \
        float* sa_ = lds + (STG) * STAGE + st_dst;                                                 \
        KC_ST_(sa_, A0);                                                                           \
        if (NLA > 1) KC_ST_(sa_ + 1 * ROWS_PER_PASS * LDS_ROW, A1);                                \
        if (NLA > 2) KC_ST_(sa_ + 2 * ROWS_PER_PASS * LDS_ROW, A2);                                \
        if (NLA > 3) KC_ST_(sa_ + 3 * ROWS_PER_PASS * LDS_ROW, A3);                                \
        KC_ST_(sa_ + A_TILE, B0);                                                                  \
        if (NLB > 1) KC_ST_(sa_ + A_TILE + 1 * ROWS_PER_PASS * LDS_ROW, B1);                       \
        if (NLB > 2) KC_ST_(sa_ + A_TILE + 2 * ROWS_PER_PASS * LDS_ROW, B2);                       \
        if (NLB > 3) KC_ST_(sa_ + A_TILE + 3 * ROWS_PER_PASS * LDS_ROW, B3);                       \
    }
#define KC_LOAD(KT) KC_LOAD_INTO(ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, min((KT), nkt - 1) * (BK * 4))
#define KC_STORE(KT, STG) KC_STORE_FROM(ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, STG)
#define KC_FRAGS(SET, STG, STEP)                                                                   \
    {                                                                                              \
        const float* as_ = lds + (STG) * STAGE + frag_a + 8 * (STEP);                              \
        const float* bs_ = lds + (STG) * STAGE + A_TILE + frag_b + 8 * (STEP);                     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
            fa[SET][i] = *reinterpret_cast<const float4*>(as_ + i * 32 * LDS_ROW);                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                             \
            fb[SET][j] = *reinterpret_cast<const float4*>(bs_ + j * 32 * LDS_ROW);                 \
    }
    // component-major MFMA order: consecutive MFMAs go to DIFFERENT accumulators
#define KC_MFMAS(SET)                                                                              \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                const float av_ = c == 0 ? fa[SET][i].x : c == 1 ? fa[SET][i].y : c == 2 ? fa[SET][i].z : fa[SET][i].w; \
                const float bv_ = c == 0 ? fb[SET][j].x : c == 1 ? fb[SET][j].y : c == 2 ? fb[SET][j].z : fb[SET][j].w; \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bv_, acc[i][j], 0, 0, 0);    \
            }

    const int frag_a = (wm * WM + li) * LDS_ROW + 4 * lh;
    const int frag_b = (wn * WN + li) * LDS_ROW + 4 * lh;
    float4 fa[2][TM], fb[2][TN];
    f32x16k acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: the loads of k-tiles 0, 1, 2 go out back-to-back (ONE exposed memory latency, not three): tiles 0 and 1
    // land in temporaries that die here, tile 2 in the loop's staging registers
    {
        u32x4k p0, p1, p2, p3, p4, p5, p6, p7, q0, q1, q2, q3, q4, q5, q6, q7;
        KC_LOAD_INTO(p0, p1, p2, p3, p4, p5, p6, p7, 0)
        KC_LOAD_INTO(q0, q1, q2, q3, q4, q5, q6, q7, min(1, nkt - 1) * (BK * 4))
        KC_LOAD(2)
        KC_STORE_FROM(p0, p1, p2, p3, p4, p5, p6, p7, 0)
        KC_STORE_FROM(q0, q1, q2, q3, q4, q5, q6, q7, 1)
    }
    __syncthreads();
    KC_FRAGS(0, 0, 0)

    // one k-tile; CUR/NXT/WR are the stages of k-tiles kt, kt+1, kt+2
#define KC_ITER(CUR, NXT, WR)                                                                      \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        KC_DBG_STAGING(KC_STORE(kt + 2, WR), KC_LOAD(kt + 3))                                      \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                           \
            KC_DBG_FRAGS(if (s + 1 < NS) KC_FRAGS((s + 1) & 1, CUR, s + 1) else KC_FRAGS(0, NXT, 0)) \
            KC_MFMAS(s & 1)                                                                        \
        }                                                                                          \
        /* pin the issue order (the scheduler otherwise regroups the LDS reads behind the MFMAs and every step   */ \
        /* starts with an exposed lgkmcnt wait); NS is even, so the look-ahead of the last step lands in set 0    */ \
        __builtin_amdgcn_sched_group_barrier(0x200, NLA + NLB, 0);   /* LDS writes of W(kt+2)   */  \
        __builtin_amdgcn_sched_group_barrier(0x020, NLA + NLB, 0);   /* buffer loads of G(kt+3) */  \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);                           \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        KC_DBG_BARRIER(__syncthreads();)                                                           \
    }
    // The loop is unrolled by three so that the stages are compile-time constants (every LDS address = base register +
    // immediate: no VALU at all in the body) and the 0..2 remaining k-tiles run through a rolled copy with runtime stages.
    // 8192^3: 142.7 -> 147.6 TF.  (Two earlier forms of the same idea LOST: if-regions for the remainder inside the
    // unrolled body, -4 %; breaks out of it, accumulators renamed per exit and 170 registers spilled.)
    int kt = 0;
    for (; kt + 2 < nkt; kt += 3) {                                  // compile-time stages: LDS addresses are base + immediate
        KC_ITER(0, 1, 2)
        { const int kt0_ = kt; (void)kt0_; }
        { ++kt; KC_ITER(1, 2, 0) }
        { ++kt; KC_ITER(2, 0, 1) }
        kt -= 2;
    }
    int cur = 0, nxt = 1, wr = 2;                                    // (kt is a multiple of 3 here) remaining 0..2 k-tiles
    for (; kt < nkt; ++kt) {
        KC_ITER(cur, nxt, wr)
        const int t_ = cur; cur = nxt; nxt = wr; wr = t_;
    }
#undef KC_ITER
#undef KC_LOAD
#undef KC_LOAD_INTO
#undef KC_STORE_FROM
#undef KC_LD_
#undef KC_ST_
#undef KC_STORE
#undef KC_FRAGS
#undef KC_MFMAS

#if KC_DBG & 32
    {   // ablation: no epilogue at all (one conditional store keeps the accumulators alive)
        float ssum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += acc[i][j][r];
        if (ssum == 12345.678f) C[0] = ssum;
        return;
    }
#endif
    // epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) inside each 32x32 tile.
    // The epilogue kind is a TEMPLATE parameter and interior tiles take a path without bounds checks: with a runtime
    // switch and per-element guards the 32 outputs of a lane cost ~2,800 instructions (ten branches each) -- at K = 512
    // that, not the stores themselves, was the 8 % between a launch (130 TF) and the k-loop asymptote (143).
    // (Measured and dropped earlier: float4 stores through swapped MFMA operands; two-iteration load prefetch.)
    const bool interior = (m0 + BM <= g.M) & (n0 + BN <= g.N);       // wave-uniform
    if (interior) {
        // buffer stores / loads from the tile's origin: the lane's byte offset is computed once (voffset), the
        // (compile-time row) * ld part lives on the scalar unit (soffset) -- one VALU instruction per element (the bias add)
        const int lane_c = ((wm * WM + 4 * lh) * g.ldc + wn * WN + li) * 4;
        const int lane_r = ((wm * WM + 4 * lh) * g.ldr + wn * WN + li) * 4;
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            C + (long long)m0 * g.ldc + n0, 0, ((BM - 1) * g.ldc + BN) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(EPI == EPI_RESIDUAL ? residg + (long long)m0 * g.ldr + n0 : Ag), 0,
            EPI == EPI_RESIDUAL ? ((BM - 1) * g.ldr + BN) * 4 : 0, 0x00020000);
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
                    f32x2k v2 = {acc[i][j][r2] + bias, acc[i][j][r2 + 1] + bias};
                    if (EPI == EPI_GELU) v2 = gelu_new_kc2(v2);
                    else if (EPI == EPI_RESIDUAL) { v2.x += res[r2]; v2.y += res[r2 + 1]; }
                    else if (EPI == EPI_SCALE_DIV) { v2.x = v2.x / g.scale_div; v2.y = v2.y / g.scale_div; }
                    else if (EPI == EPI_HALF_PLUS) { v2.x = (v2.x + 1.0f) / 2.0f; v2.y = (v2.y + 1.0f) / 2.0f; }
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = r2 + h2;
                        const float v = h2 ? v2.y : v2.x;
#if KC_DBG & 16
                        if (v == 12345.678f)                             // ablation: (almost) never true, keeps v alive
#endif
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
            if (EPI == EPI_RESIDUAL) {
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
                if (EPI == EPI_GELU) v = gelu_new_kc(v);
                else if (EPI == EPI_RESIDUAL) v += res[r];
                else if (EPI == EPI_SCALE_DIV) v = v / g.scale_div;
                else if (EPI == EPI_HALF_PLUS) v = (v + 1.0f) / 2.0f;
#if KC_DBG & 16
                if (v == 12345.678f)
#endif
                if (row < g.M && col_ok) C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

struct KcTile { int bm, bn, cls, blocks_per_cu, waves_per_block; double eff; };
static const KcTile kKc[] = {   // eff = measured k-loop asymptote / 157.3 TF (8192^3, tools/gemm_bench.py shape 14)
    {128, 128, PK_GEMM_KC_128x128x32, 1, 8, 0.0},    // BK 32, 108 KB, one workgroup per CU: 136 TF asymptote; the BK 16 form
                                                     // wins at every size, so eff 0 = only via R4D_GEMM_KC_TILE=0
    {128, 128, PK_GEMM_KC_128x128x16, 2, 8, 0.94},   // BK 16, three stages = 60 KB: two per CU
    {128, 64, PK_GEMM_KC_128x64x16, 3, 4, 0.88},     // BK 16, 45 KB: three per CU
    {64, 64, PK_GEMM_KC_64x64x32, 2, 4, 0.82},       // BK 32, 54 KB: two per CU
};
// (measured and dropped: 4 waves with 64x64 wave tiles at BK 16 -- 140 TF asymptote but 121 TF at K = 512; 4 waves at
// BK 32, one workgroup per CU -- 127 / 101 TF; BK 32 with TWO stages, two workgroups per CU -- 140 / 125 TF; a 256x128
// macro-tile with 64x64 wave tiles, 8 waves, one workgroup per CU -- 138 / 113 TF: two co-resident workgroups are worth
// more than fewer LDS reads per MFMA)
constexpr int kNumKc = 4;

template <int BM, int BN, int BK, int WGM, int WGN, int MINW>
static int launch_kc(const GemmArgs& g, int cls, hipStream_t stream) {
    constexpr int NTHREADS = 64 * WGM * WGN;
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    const double flop = (g.causal ? 1.0 : 2.0) * (double)g.M * g.N * g.K * g.nbatch;
    ProfScope prof(cls, flop, stream);
    KcShape sh;
    sh.M = g.M; sh.N = g.N; sh.K = g.K; sh.lda = g.lda; sh.ldb = g.ldb; sh.ldc = g.ldc; sh.ldr = g.ldr;
    sh.nb1 = g.nb1; sh.epilogue = g.epilogue; sh.causal = g.causal;
    sh.sA0 = g.sA0; sh.sA1 = g.sA1; sh.sB0 = g.sB0; sh.sB1 = g.sB1; sh.sC0 = g.sC0; sh.sC1 = g.sC1;
    sh.scale_div = g.scale_div;
#define KC_LAUNCH_(E)                                                                              \
    hipLaunchKernelGGL((gemm_f32_kc_kernel<BM, BN, BK, WGM, WGN, MINW, E>), dim3(tiles, 1, g.nbatch), dim3(NTHREADS), 0, \
                       stream, g.A, g.B, g.C, g.bias, g.resid, sh)
    switch (g.epilogue) {
        case EPI_NONE: KC_LAUNCH_(EPI_NONE); break;
        case EPI_GELU: KC_LAUNCH_(EPI_GELU); break;
        case EPI_RESIDUAL: KC_LAUNCH_(EPI_RESIDUAL); break;
        case EPI_SCALE_DIV: KC_LAUNCH_(EPI_SCALE_DIV); break;
        case EPI_HALF_PLUS: KC_LAUNCH_(EPI_HALF_PLUS); break;
        default: set_error("gemm_f32_kc: unknown epilogue %d", g.epilogue); return R4D_ERR_INVALID;
    }
#undef KC_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_f32_kc");
    return R4D_OK;
}

// cost of one launch of `rows` x N with tile t, in (tile elements / efficiency) units: a CU works through its
// ceil(blocks / 256) tiles at the tile's measured efficiency, derated when too few waves are resident
static double kc_cost(const GemmArgs& g, int rows, int t, long long* per_cu_out = nullptr) {
    const KcTile& c = kKc[t];
    long long blocks = (long long)cdiv(rows, c.bm) * cdiv(g.N, c.bn) * g.nbatch;
    if (g.causal == CAUSAL_QK) blocks = blocks / 2 + (long long)cdiv(rows, c.bm) * g.nbatch / 2;
    const long long per_cu = (blocks + 255) / 256;
    const double conc = (double)(per_cu < c.blocks_per_cu ? per_cu : c.blocks_per_cu);
    const double wps = conc * c.waves_per_block / 4.0;
    // one 4-wave workgroup alone on a CU (one wave per SIMD) runs at ~0.8 of the two-per-SIMD rate (measured: a lone
    // 64x64x2048 tile 43 us, a lone 128x64 80 us, a lone 8-wave 128x128 133 us)
    const double eff = c.eff * (wps >= 2.0 ? 1.0 : 0.6 + 0.2 * wps);
    if (per_cu_out) *per_cu_out = per_cu;
    return (double)per_cu * c.bm * c.bn / eff;
}

static int kc_best_tile(const GemmArgs& g, int rows, double* cost_out) {
    int best = kNumKc - 1;
    double best_cost = 1e300;
    for (int t = 0; t < kNumKc; ++t) {
        if (kKc[t].eff <= 0.0) continue;
        const double c = kc_cost(g, rows, t);
        if (c < best_cost) { best_cost = c; best = t; }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

static int kc_launch_tile(const GemmArgs& g, int t, hipStream_t stream) {
    switch (t) { case 0: R4D_BRANCH(KC_128x128x32); break; case 1: R4D_BRANCH(KC_128x128x16); break; case 2: R4D_BRANCH(KC_128x64x16); break; default: R4D_BRANCH(KC_64x64x32); }
    switch (t) {
        case 0: return launch_kc<128, 128, 32, 4, 2, 2>(g, kKc[0].cls, stream);
        case 1: return launch_kc<128, 128, 16, 4, 2, 4>(g, kKc[1].cls, stream);
        case 2: return launch_kc<128, 64, 16, 2, 2, 3>(g, kKc[2].cls, stream);
        default: return launch_kc<64, 64, 32, 2, 2, 2>(g, kKc[3].cls, stream);
    }
}

// k-contiguous GEMM: requires g.b_trans (B is [N,K]) and no causal P.V trimming.
//
// Tile quantisation: a launch takes ceil(tiles / 256) tile-times per CU, so 4.1 tiles per CU cost as much as 5
// (measured: M = 33,600..38,400, N = 512 all take 625 us with the 128x128 tile).  When the last round is mostly empty
// the rows are SPLIT: a main launch whose tile count fills whole rounds, and a second launch over the remaining rows
// with whatever tile is cheapest for them (usually a smaller one that spreads over more CUs).  Plain (unbatched,
// non-causal) GEMMs only; rows are independent, so the results are bit-identical to a single launch.  (Measured and
// dropped: running the remainder on a side stream, forked/joined with events, so that it overlaps the main launch --
// 3 % SLOWER on the bench step; the event barriers cost more than the idle tail they remove.)
int launch_gemm_f32_kc(const GemmArgs& g, hipStream_t stream) {
    static int forced = -2, split = -1;
    if (forced == -2) {
        const char* e = getenv("R4D_GEMM_KC_TILE");   // tuning aid: 0..3 forces a tile shape (and no row split)
        forced = e ? atoi(e) : -1;
        const char* sp = getenv("R4D_GEMM_KC_SPLIT"); // tuning aid: 0 disables the row split
        split = sp ? atoi(sp) : 1;
    }
    if (forced >= 0 && forced < kNumKc) return kc_launch_tile(g, forced, stream);
    double single_cost;
    const int single = kc_best_tile(g, g.M, &single_cost);
    if (split && g.nbatch == 1 && g.causal == CAUSAL_NONE) {
        // a second launch costs ~5 us of ramp-up/drain: in cost units (time * K), relative to a 128x128 tile at K = 512
        // taking ~27 us alone on a CU (16384 / 0.91 units)
        const double launch_overhead = 0.2 * (16384.0 / 0.91) * 512.0 / (double)g.K;
        double best_cost = single_cost;
        int best_t1 = -1, best_t2 = -1, best_m1 = 0;
        for (int t1 = 0; t1 < kNumKc; ++t1) {
            const KcTile& c = kKc[t1];
            if (c.eff <= 0.0) continue;
            const long long tiles_n = cdiv(g.N, c.bn);
            const long long rounds = (long long)cdiv(g.M, c.bm) * tiles_n / 256;          // whole rounds available
            if (rounds < 1) continue;
            const long long tiles_m1 = rounds * 256 / tiles_n;                             // rows that fill them
            const int m1 = (int)(tiles_m1 * c.bm);
            if (m1 <= 0 || m1 >= g.M) continue;
            double c2;
            const int t2 = kc_best_tile(g, g.M - m1, &c2);
            const double total = kc_cost(g, m1, t1) + c2 + launch_overhead;
            if (total < best_cost) { best_cost = total; best_t1 = t1; best_t2 = t2; best_m1 = m1; }
        }
        if (best_t1 >= 0 && best_cost < 0.97 * single_cost) {
            R4D_BRANCH(KC_ROWSPLIT);
            GemmArgs a = g, b = g;
            a.M = best_m1;
            b.M = g.M - best_m1;
            b.A = g.A + (long long)best_m1 * g.lda;
            b.C = g.C + (long long)best_m1 * g.ldc;
            if (g.resid) b.resid = g.resid + (long long)best_m1 * g.ldr;
            const int rc = kc_launch_tile(a, best_t1, stream);
            if (rc) return rc;
            return kc_launch_tile(b, best_t2, stream);
        }
    }
    return kc_launch_tile(g, single, stream);
}

}  // namespace r4d

namespace r4d { int dbgflag_kc() { return KC_DBG != 0; } }
