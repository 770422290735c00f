// "f16x2" GEMM with BOTH operands pre-split (round 5): C[M,N] = epilogue(A . W^T + bias), the arithmetic of
// gemm_h2.hip instruction for instruction -- same fragments, same three products per 16 k, same two accumulator sets, same epilogues --
// but the A operand arrives ALREADY as f16x2 lines, written by its producer (LayerNorm: encoder_ops.hip; the c_fc GELU epilogue:
// this file) in the layout r4d_split2_planes_f16 gives the weights:
//     [rows][K/32][2][32] fp16 -- per row and 32 consecutive k ONE 128-byte line of 32 hi = RN16(x/4) then 32 lo' = RN16((x/4 - hi) 2^11)
// (the same 4 bytes per element as fp32: nothing more is read or written).  Both tiles then travel global -> LDS by DMA
// (global_load_lds_dwordx4): no VGPR staging, no split arithmetic and no ds_write in the k-loop -- what is left beside the MFMAs is
// the fragment reads.  Round 4 measured this form once (402 / 427 us against 430 / 454 for c_attn / c_fc, bit-identical) and dropped
// it for want of producers; round 5 has them.
//
// Staging: one DMA instruction = 64 lanes x 16 bytes = 8 WHOLE lines (8 rows x 128 bytes) -> 1 KB of LDS, lane-linear: lane l lands
// in row l >> 3, 16-byte slot l & 7 of that row's 128 bytes.  The image wants logical chunk c (0..3 hi, 4..7 lo') of row r in slot
// c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments: score.hip's image), so the lane READS chunk (l & 7) ^ ((r >> 1) & 7):
// the permutation is on the global side.  A stage = (BM + BN) rows x 128 bytes = 48 KB at 128 x 256; three stages; wavefront w
// copies A row groups 2w, 2w+1 and W row groups 4w .. 4w+3 of every k-tile (6 instructions), k-tile kt + 2 while kt is multiplied.
// ONE barrier per k-tile, in the middle of the iteration: behind `s_waitcnt vmcnt` for the wave's own copies of k-tile kt + 1, in
// front of the first fragment read of that k-tile.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "h2.h"

namespace r4d {

typedef float f32x16p __attribute__((ext_vector_type(16)));
typedef float f32x2p __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));

int g_gemm_h2p = 1;               // r4d_set_gemm_h2p: 0 keeps the register-staged gemm_h2 kernels and fp32 activations everywhere

__device__ __forceinline__ f32x2p gelu_new_h2p(f32x2p x) {           // gemm_h2.hip's, same instructions
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    const f32x2p a = x * x * k1 + k0;
    const f32x2p w = x * a;
    f32x2p e;
    e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
    e = e + 1.0f;
    f32x2p r;
    r.x = __builtin_amdgcn_rcpf(e.x); r.y = __builtin_amdgcn_rcpf(e.y);
    return x * r;
}
__device__ __forceinline__ float gelu_new_h2p_1(float x) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, k1, k0)));
}

struct H2PShape {
    int M, N, K, ldc, ldr;
    unsigned* kblk; int kb_hd;        // EPI_H2WORDS with N = 3 d: the K columns [d, 2d) go to the key-blocked image (attention_h2.hip) instead of C
};

// NI DMA instructions of 1 KB each: global address = base + 32-bit lane offset + i KB, LDS address = M0 + i KB + 16 lane (the
// instruction offset moves BOTH sides, so the lane offsets of instruction i are built i KB short); M0 restored (the compiler owns it)
__device__ __forceinline__ void h2p_dma2(const void* base, unsigned v0, unsigned v1, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void h2p_dma4(const void* base, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(lds_dst) : "memory");
}
template <int N_> __device__ __forceinline__ void h2p_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }

// OUT_LINES (with EPI_GELU): C is the f16x2-line image of the result, [M][N/32][2][32] fp16 (the A operand of the next GEMM)
// WGM x WGN wavefronts: 2 x 4 (wave tile 64 x 64 at 128 x 256, two wavefronts per SIMD).  The 2 x 2 grid (wave tile 64 x 128, ONE
// wavefront per SIMD with the whole register file, 256 VGPRs + 256 AGPRs: 12 fragment reads per 24 MFMAs instead of 8 per 12, a
// quarter fewer LDS bytes per flop) instantiates from the same template and was measured in round 5: bit-identical and 6-19 %
// SLOWER (c_attn 389 vs 367 us, c_fc 568 vs 513, K 2048 484 vs 448 on the stand-alone probe): with one wavefront per SIMD nothing
// runs under its barrier, its fragment waits and its epilogue.  Not launched.
template <int BM, int BN, int EPI, bool OUT_LINES, int WGM = 2, int WGN = 4>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4) void gemm_h2p_kernel(const unsigned short* __restrict__ Al, const unsigned short* __restrict__ Bp,
                                                          float* __restrict__ Cg, const float* __restrict__ biasg,
                                                          const float* __restrict__ residg, const H2PShape g) {
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int STAGE = (BM + BN) * 8;                              // uint4 units: a row = 8 slots of 16 bytes (hi 0..3, lo' 4..7, XOR-ed)
    constexpr int NBUF = 3;
    constexpr int NGA = BM / 8 / NW, NGB = BN / 8 / NW;               // 8-row groups per wavefront and k-tile: 2 of A, 4 (2) of W with 8 wavefronts
    constexpr int NDMA = NGA + NGB;
    static_assert(BM == 128 && (BN == 256 || BN == 128) && TM >= 1 && TN >= 1 && (NGA == 2 || NGA == 4) && (NGB == 2 || NGB == 4 || NGB == 8), "tile");
    __shared__ u32x4p lds[NBUF * STAGE];

    // XCD-aware grouped tile order (gemm_h2.hip)
    const int nblk = gridDim.x, xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;
    const int bid = xcd * xq + min(xcd, xr) + (blockIdx.x >> 3);
    constexpr int GROUP_M = 8;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int per_group = GROUP_M * tiles_n;
    const int grp = bid / per_group, first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (bid % per_group) % gsz, tile_n = (bid % per_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = g.K / 32;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, lh = lane >> 5;

    // DMA lane offsets (bytes from the operand's base; the k-tile is added to the scalar base): group j = rows 8j .. 8j+7 of the tile
    unsigned va[NGA], vb[NGB];
    const int lrow = lane >> 3, lslot = lane & 7;
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
        const int r = (wid * NGA + i) * 8 + lrow;                     // row of the A tile
        va[i] = (unsigned)min(m0 + r, g.M - 1) * (unsigned)(g.K * 4) + 16u * (unsigned)(lslot ^ ((r >> 1) & 7)) + 3072u - 1024u * (i & 3);
    }
#pragma unroll
    for (int i = 0; i < NGB; ++i) {
        const int r = (wid * NGB + i) * 8 + lrow;                     // row of the W tile
        vb[i] = (unsigned)min(n0 + r, g.N - 1) * (unsigned)(g.K * 4) + 16u * (unsigned)(lslot ^ ((r >> 1) & 7)) + 3072u - 1024u * (i & 3);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const unsigned wid_s = __builtin_amdgcn_readfirstlane((unsigned)wid);     // wave-uniform: the DMA destinations live in SGPRs (M0)
    const unsigned dst_a = lds0 + (wid_s * NGA) * 1024u, dst_b = lds0 + ((unsigned)(BM / 8) + wid_s * NGB) * 1024u;
    const char* abase = reinterpret_cast<const char*>(Al) - 3072;     // (the lane offsets carry + 3 KB - i KB: never negative, whatever the shape)
    const char* bbase = reinterpret_cast<const char*>(Bp) - 3072;
#define H2P_ISSUE(KT, STG)                                                                         \
    {                                                                                              \
        const unsigned so_ = (unsigned)(STG) * (unsigned)(STAGE * 16);                             \
        if constexpr (NGA == 4) h2p_dma4(abase + (size_t)(KT) * 128, va[0], va[1], va[2], va[3], dst_a + so_); \
        else h2p_dma2(abase + (size_t)(KT) * 128, va[0], va[1], dst_a + so_);                      \
        if constexpr (NGB == 8) {                                                                  \
            h2p_dma4(bbase + (size_t)(KT) * 128, vb[0], vb[1], vb[2], vb[3], dst_b + so_);          \
            h2p_dma4(bbase + (size_t)(KT) * 128, vb[4], vb[5], vb[6], vb[7], dst_b + so_ + 4096u);  /* same global base: only M0 moves */ \
        } else if constexpr (NGB == 4) h2p_dma4(bbase + (size_t)(KT) * 128, vb[0], vb[1], vb[2], vb[3], dst_b + so_); \
        else h2p_dma2(bbase + (size_t)(KT) * 128, vb[0], vb[1], dst_b + so_);                       \
    }

    // fragment addresses: lane (row li of its 32-row tile, half lh), k-step s, plane p -> slot (4p + 2s + lh) ^ ((li >> 1) & 7)
    const int fsw = (li >> 1) & 7;
    int f_off[2][2];                                                 // [k-step][plane]
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int p = 0; p < 2; ++p) f_off[s_][p] = li * 8 + ((4 * p + 2 * s_ + lh) ^ fsw);
    const int fa_base = wm * WM * 8, fb_base = BM * 8 + wn * WN * 8;

    f32x16p acc0[TM][TN], acc1[TM][TN];                  // hi.hi  /  the two cross terms (factor 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

#define H2P_MFMA(ACC, A_, B_, I_, J_) \
    ACC[I_][J_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, A_), __builtin_bit_cast(f16x8p, B_), ACC[I_][J_], 0, 0, 0);
    u32x4p fa[2][TM][2], fb[2][TN][2];                   // two fragment sets: the reads of k-step s+1 travel under the MFMAs of k-step s
#define H2P_FRAGS(SET, STG, S)                                                                     \
    {                                                                                              \
        const u32x4p* st_ = lds + (STG) * STAGE;                                                   \
        /* in the order the MFMAs want them: lo(A) . hi(B) first */                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][1] = st_[fa_base + i * 256 + f_off[S][1]]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][0] = st_[fb_base + j * 256 + f_off[S][0]]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i][0] = st_[fa_base + i * 256 + f_off[S][0]]; \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j][1] = st_[fb_base + j * 256 + f_off[S][1]]; \
    }
    /* gemm_h2.hip's order: consecutive MFMAs go to different accumulators; the two writes of one acc1 tile are TM TN instructions apart */
#define H2P_MFMAS(SET)                                                                             \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2P_MFMA(acc1, fa[SET][i][1], fb[SET][j][0], i, j) \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2P_MFMA(acc0, fa[SET][i][0], fb[SET][j][0], i, j) \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) H2P_MFMA(acc1, fa[SET][i][0], fb[SET][j][1], i, j) \
    }

    // prologue: k-tiles 0 and 1 on their way, k-tile 0 landed for everybody, k-step 0 of it in fragment set 0
    H2P_ISSUE(0, 0)
    if (nkt > 1) { H2P_ISSUE(1, 1) h2p_wait_vm<NDMA>(); } else { h2p_wait_vm<0>(); }
    __syncthreads();
    H2P_FRAGS(0, 0, 0)

    constexpr int NMF = 3 * TM * TN, NFR = 2 * (TM + TN);
    static_assert(NMF >= NFR, "interleave");
    // iteration kt (stage CUR = kt % 3 landed and visible, fragment set 0 = its k-step 0):
    //   the DMA of k-tile kt + 2 into stage (kt + 2) % 3 -- last read in iteration kt - 1, in front of that iteration's barrier;
    //   k-step 0's MFMAs with the reads of k-step 1 (set 1) between them;
    //   wait for the wave's own copies of k-tile kt + 1 (all but the NDMA just issued), barrier: k-tile kt + 1 is in LDS for everybody
    //   and nobody reads stage CUR's k-step 1 any more ... (the fragments are in registers: lgkmcnt(0) in front of the barrier);
    //   k-step 1's MFMAs with the reads of k-step 0 of stage NXT between them.
#define H2P_ITER(CUR, NXT, WR)                                                                     \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        const bool more_ = kt + 2 < nkt;                                                           \
        if (more_) H2P_ISSUE(kt + 2, WR)                                                           \
        H2P_FRAGS(1, CUR, 1)                                                                       \
        H2P_MFMAS(0)                                                                               \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (more_) h2p_wait_vm<NDMA>(); else h2p_wait_vm<0>();                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                            \
        H2P_FRAGS(0, NXT, 0)                                                                       \
        H2P_MFMAS(1)                                                                               \
        _Pragma("unroll") for (int m_ = 0; m_ < NMF; ++m_) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
            if (m_ < NFR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    // (in the last iteration stage NXT holds nothing new: its k-step-0 reads fetch stale bytes nobody multiplies)
    int kt = 0;
    for (; kt + 2 < nkt; kt += 3) {                                   // compile-time stages
        H2P_ITER(0, 1, 2)
        { ++kt; H2P_ITER(1, 2, 0) }
        { ++kt; H2P_ITER(2, 0, 1) }
        kt -= 2;
    }
    if (kt < nkt) {                                                   // kt is a multiple of 3 here: one or two k-tiles left
        H2P_ITER(0, 1, 2)
        if (kt + 1 < nkt) { ++kt; H2P_ITER(1, 2, 0) }
    }
#undef H2P_ITER
#undef H2P_FRAGS
#undef H2P_MFMAS
#undef H2P_MFMA
#undef H2P_ISSUE

    // epilogue: gemm_h2.hip's, instruction for instruction, on value = (acc0 + 2^-11 acc1) * unscale
    constexpr float UNS = H2_A_UNSCALE;
    float* __restrict__ C = Cg;
    const bool interior = (m0 + BM <= g.M) & (n0 + BN <= g.N);       // wave-uniform
    if constexpr (OUT_LINES) {
        // The result as f16x2 lines [M][N/32][2][32] (the A operand of the next GEMM; value / 4 split as split2_pair<true>).  A lane
        // holds ONE column of a 32-column block (= one line) for 16 rows; the two fp16 of a packed word are NEIGHBOURING columns:
        // lanes 2c and 2c+1 swap one value per pair of rows (one DPP move), the even lane then packs row r, the odd lane row r+1 --
        // columns (2c, 2c+1) of its row -- and stores one hi word and one lo' word: as many 4-byte stores as the fp32 epilogue.
        static_assert(EPI == EPI_GELU, "line output: the c_fc epilogue only");
        unsigned char* __restrict__ L = reinterpret_cast<unsigned char*>(Cg);
        const long long row_bytes = (long long)g.N * 4;              // N/32 lines of 128 bytes
        const bool odd = li & 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int colb = n0 + wn * WN + j * 32;                   // first column of the block: line colb / 32
            const bool col_ok = colb + li < g.N;                      // (N % 32 == 0: a block is in or out as a whole)
            const float bias = (biasg && col_ok) ? biasg[colb + li] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r2 = 0; r2 < 16; r2 += 2) {
                    f32x2p v2 = {__builtin_fmaf(acc1[i][j][r2], H2_LO_UNSCALE, acc0[i][j][r2]) * UNS + bias,
                                 __builtin_fmaf(acc1[i][j][r2 + 1], H2_LO_UNSCALE, acc0[i][j][r2 + 1]) * UNS + bias};
                    v2 = gelu_new_h2p(v2);
                    const float vx = v2.x, vy = v2.y;                 // rows r2 and r2 + 1 of this lane's column
                    const float send = odd ? vx : vy;
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    const float c0 = odd ? recv : vx, c1 = odd ? vy : recv;      // columns (2c, 2c+1) of row r2 (even lane) / r2 + 1 (odd lane)
                    unsigned h, l;
                    split2_pair<true>(c0, c1, h, l);
                    const int r = r2 + (odd ? 1 : 0);
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (interior || (row < g.M && col_ok)) {
                        unsigned* dst = reinterpret_cast<unsigned*>(L + (long long)row * row_bytes + (long long)(colb >> 5) * 128 + (li >> 1) * 4);
                        dst[0] = h;
                        dst[16] = l;                                  // + 64 bytes: the lo' half of the line
                    }
                }
            }
        }
        return;
    } else {
        if (interior) {
            const int lane_c = ((wm * WM + 4 * lh) * g.ldc + wn * WN + li) * 4;
            const int lane_r = ((wm * WM + 4 * lh) * g.ldr + wn * WN + li) * 4;
            const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                C + (long long)m0 * g.ldc + n0, 0, ((BM - 1) * g.ldc + BN) * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(EPI == EPI_RESIDUAL ? residg + (long long)m0 * g.ldr + n0 : Cg), 0,
                EPI == EPI_RESIDUAL ? ((BM - 1) * g.ldr + BN) * 4 : 0, 0x00020000);
            // EPI_H2WORDS with a key-blocked K image: a 32-column block of the K third is (one head, four steps u0 .. u0+3); the MFMA
            // tile's 32 rows ARE one key block (m0, WM and the tile are multiples of 32): element (row, col) -> chunk (block, head, u),
            // slot [half = (e >> 2) & 1][row & 31][e & 3], e = the column within the head
            const int kd = g.N / 3;
            const __amdgpu_buffer_rsrc_t kb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (EPI == EPI_H2WORDS && g.kb_hd) ? g.kblk : reinterpret_cast<unsigned*>(Cg), 0,
                (EPI == EPI_H2WORDS && g.kb_hd) ? ((g.M + 31) / 32) * (kd * 128) : 0, 0x00020000);
            const int lane_kb = (li >> 3) * 1024 + ((li >> 2) & 1) * 512 + ((li & 3) + 4 * lh) * 16;       // chunk (step), half, row slot of the lane after the quad transpose
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float bias = biasg ? biasg[n0 + wn * WN + j * 32 + li] : 0.f;
                const int colb = n0 + wn * WN + j * 32;
                const bool to_kb = EPI == EPI_H2WORDS && g.kb_hd && colb >= kd && colb < 2 * kd;        // wave-uniform
                const int kb_chunk = to_kb ? ((colb - kd) / g.kb_hd) * (g.kb_hd / 8) + ((colb - kd) % g.kb_hd) / 8 : 0;   // head * NSTEP + u0
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int kb_soff = (((m0 + wm * WM + i * 32) >> 5) * (kd / 8) + kb_chunk) * 1024;     // bytes: block * (H * NSTEP) chunks of 1 KB
                    float res[16];
                    if (EPI == EPI_RESIDUAL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                r_rsrc, lane_r, ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldr + j * 32) * 4, 0));
                    }
                    unsigned o[16];
#pragma unroll
                    for (int r2 = 0; r2 < 16; r2 += 2) {
                        f32x2p v2 = {__builtin_fmaf(acc1[i][j][r2], H2_LO_UNSCALE, acc0[i][j][r2]) * UNS + bias,
                                     __builtin_fmaf(acc1[i][j][r2 + 1], H2_LO_UNSCALE, acc0[i][j][r2 + 1]) * UNS + bias};
                        if (EPI == EPI_GELU) v2 = gelu_new_h2p(v2);
                        else if (EPI == EPI_RESIDUAL) { v2.x += res[r2]; v2.y += res[r2 + 1]; }
                        const float vx = v2.x, vy = v2.y;     // (copies first: __builtin_bit_cast on an ext-vector ELEMENT reads element 0)
                        o[r2] = __builtin_bit_cast(unsigned int, vx); o[r2 + 1] = __builtin_bit_cast(unsigned int, vy);
                        if (EPI == EPI_H2WORDS) h2_words<true>(v2.x, v2.y, o[r2], o[r2 + 1]);     // C is the uint32 word image of the result (attention_h2.hip)
                    }
                    if (EPI == EPI_H2WORDS && to_kb) {
                        // registers 4a .. 4a+3 of the four lanes of a quad = rows 8a + (0..3) + 4 lh x the four words of one 16-byte
                        // slot: a 4 x 4 transpose inside the quad (two DPP exchange stages) gives every lane ONE row's slot, and the
                        // wavefront's store covers eight whole 128-byte lines (8 consecutive rows of 8 (step, half) chunks) -- as 4-byte
                        // stores to 16 scattered pieces per instruction the c_attn launch was 80 us longer.  (The same transpose for the
                        // ROW-MAJOR stores of every epilogue -- 16 bytes per lane, four times fewer store instructions -- was measured too:
                        // c_attn 295 -> 307 us, residual 258 -> 261: the VALU work costs more than the stores save.  Not kept.)
                        const bool o1 = li & 1, o2 = li & 2;
#pragma unroll
                        for (int a4 = 0; a4 < 16; a4 += 4) {
                            unsigned x0 = o[a4], x1 = o[a4 + 1], x2 = o[a4 + 2], x3 = o[a4 + 3];
                            {   // stage 1: lanes t ^ 1, registers r ^ 1
                                const unsigned s01 = o1 ? x0 : x1, s23 = o1 ? x2 : x3;
                                const unsigned r01 = (unsigned)__builtin_amdgcn_mov_dpp((int)s01, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
                                const unsigned r23 = (unsigned)__builtin_amdgcn_mov_dpp((int)s23, 0xB1, 0xF, 0xF, true);
                                if (o1) { x0 = r01; x2 = r23; } else { x1 = r01; x3 = r23; }
                            }
                            {   // stage 2: lanes t ^ 2, registers r ^ 2
                                const unsigned s02 = o2 ? x0 : x2, s13 = o2 ? x1 : x3;
                                const unsigned r02 = (unsigned)__builtin_amdgcn_mov_dpp((int)s02, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
                                const unsigned r13 = (unsigned)__builtin_amdgcn_mov_dpp((int)s13, 0x4E, 0xF, 0xF, true);
                                if (o2) { x0 = r02; x1 = r13; } else { x2 = r02; x3 = r13; }
                            }
                            const u32x4p slot = {x0, x1, x2, x3};
                            __builtin_amdgcn_raw_buffer_store_b128(slot, kb_rsrc, lane_kb, kb_soff + 2 * a4 * 16, 0);      // rows 8 (a4 / 4) ..
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(o[r], c_rsrc, lane_c, ((i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + j * 32) * 4, 0);
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
                    if (EPI == EPI_GELU) v = gelu_new_h2p_1(v);
                    else if (EPI == EPI_RESIDUAL) v += residg[(long long)min(row, g.M - 1) * g.ldr + colc];
                    else if (EPI == EPI_H2WORDS) { unsigned w0, w1; h2_words<true>(v, 0.f, w0, w1); v = __builtin_bit_cast(float, w0); }
                    if (EPI == EPI_H2WORDS && g.kb_hd && col >= g.N / 3 && col < 2 * (g.N / 3)) {      // the K third: key-blocked image
                        const int kd = g.N / 3, e = (col - kd) % g.kb_hd, chunk = ((col - kd) / g.kb_hd) * (g.kb_hd / 8) + (e >> 3);
                        if (row < g.M)
                            g.kblk[((long long)(row >> 5) * (kd / 8) + chunk) * 256 + (((e >> 2) & 1) * 32 + (row & 31)) * 4 + (e & 3)] =
                                __builtin_bit_cast(unsigned, v);
                        continue;
                    }
                    if (row < g.M && col_ok) C[(long long)row * g.ldc + col] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- fp32 rows -> f16x2 lines
// (what the producers write, as a kernel of its own: tests, and callers whose A comes from somewhere else)
__global__ __launch_bounds__(256) void split2_lines_kernel(const float* __restrict__ x, long long n4, uint2* __restrict__ lines, int K) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // float4 index
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    unsigned h0, l0, h1, l1;
    split2_pair<true>(v.x, v.y, h0, l0);
    split2_pair<true>(v.z, v.w, h1, l1);
    const long long e = i * 4, row = e / K;
    const int k = (int)(e - row * K);
    unsigned char* dst = reinterpret_cast<unsigned char*>(lines) + row * (long long)K * 4 + (k >> 5) * 128 + (k & 31) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(l0, l1);
}

// ---------------------------------------------------------------------------------------------- host side
bool gemm_h2p_supported(int M, int K, int N) {
    return g_gemm_h2p != 0 && gemm_h2_supported(M, K, N) && (long long)M * K * 4 + 4096 < (1ll << 32) && (long long)N * K * 4 + 4096 < (1ll << 32);
}

template <int BN>
static int launch_h2p(const S3Args& a, const unsigned short* a_lines, bool out_lines, int cls, hipStream_t stream) {
    const int tiles = cdiv(a.M, 128) * cdiv(a.N, BN);
    ProfScope prof(cls, 2.0 * (double)a.M * a.N * a.K, stream);
    H2PShape sh;
    sh.M = a.M; sh.N = a.N; sh.K = a.K; sh.ldc = a.ldc; sh.ldr = a.ldr;
    sh.kblk = a.kblk; sh.kb_hd = a.kblk ? a.kb_hd : 0;
#define H2P_LAUNCH_(E, OL) hipLaunchKernelGGL((gemm_h2p_kernel<128, BN, E, OL>), dim3(tiles), dim3(512), 0, stream, a_lines, a.planes, a.C, a.bias, a.resid, sh)
    if (out_lines) {
        if (a.epilogue != EPI_GELU) { set_error("gemm_h2p: line output exists for the GELU epilogue only"); return R4D_ERR_INVALID; }
        H2P_LAUNCH_(EPI_GELU, true);
    } else {
        switch (a.epilogue) {
            case EPI_NONE: H2P_LAUNCH_(EPI_NONE, false); break;
            case EPI_GELU: H2P_LAUNCH_(EPI_GELU, false); break;
            case EPI_RESIDUAL: H2P_LAUNCH_(EPI_RESIDUAL, false); break;
            case EPI_H2WORDS: H2P_LAUNCH_(EPI_H2WORDS, false); break;
            default: set_error("gemm_h2p: epilogue %d has no instantiation", a.epilogue); return R4D_ERR_INVALID;
        }
    }
#undef H2P_LAUNCH_
    R4D_CHECK_LAUNCH("gemm_h2p");
    return R4D_OK;
}

// a.A is ignored: the A operand is `a_lines` ([M][K/32][2][32] fp16); out_lines: a.C receives [M][N/32][2][32] fp16 instead of fp32
int launch_gemm_h2p(const S3Args& a, const unsigned short* a_lines, bool out_lines, hipStream_t stream) {
    R4D_REQUIRE(a_lines && a.planes && a.C, "gemm_h2p: null pointer");
    R4D_REQUIRE(gemm_h2p_supported(a.M, a.K, a.N), "gemm_h2p: unsupported shape M=%d K=%d N=%d", a.M, a.K, a.N);
    R4D_REQUIRE(((uintptr_t)a_lines % 16) == 0 && ((uintptr_t)a.planes % 16) == 0, "gemm_h2p: alignment");
    R4D_REQUIRE(!out_lines || a.N % 32 == 0, "gemm_h2p: line output needs N %% 32 == 0 (N = %d)", a.N);
    R4D_REQUIRE(a.epilogue != EPI_RESIDUAL || a.resid, "gemm_h2p: the residual epilogue needs the second buffer");
    R4D_REQUIRE(!a.kblk || (a.epilogue == EPI_H2WORDS && a.N % 3 == 0 && a.kb_hd >= 32 && a.kb_hd % 32 == 0 && (a.N / 3) % a.kb_hd == 0),
                "gemm_h2p: the key-blocked K image goes with the h2-word epilogue, N = 3 d and a head_dim that is a multiple of 32");
    // fewest tile waves; the wide tile wins ties (gemm_h2.hip's rule: a row's result never depends on the tile either way)
    const long long b0 = (long long)cdiv(a.M, 128) * cdiv(a.N, 256), b1 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128);
    const double c0 = (double)((b0 + 255) / 256) * 128 * 256, c1 = (double)((b1 + 255) / 256) * 128 * 128 / 0.9;
    if (c1 < c0) { R4D_BRANCH(H2P_128x128); return launch_h2p<128>(a, a_lines, out_lines, PK_GEMM_H2_128x128, stream); }
    R4D_BRANCH(H2P_128x256);
    return launch_h2p<256>(a, a_lines, out_lines, PK_GEMM_H2_128x256, stream);
}

int launch_split2_lines(const float* x, long long rows, int K, unsigned short* lines, hipStream_t s) {
    R4D_REQUIRE(x && lines && rows >= 0 && K >= 32 && K % 32 == 0, "split2_lines: bad arguments (K %% 32 == 0 wanted)");
    R4D_REQUIRE((((uintptr_t)x | (uintptr_t)lines) & 15) == 0, "split2_lines: alignment");
    const long long n4 = rows * K / 4;
    if (n4 == 0) return R4D_OK;
    hipLaunchKernelGGL(split2_lines_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, n4, reinterpret_cast<uint2*>(lines), K);
    R4D_CHECK_LAUNCH("split2_lines");
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_set_gemm_h2p(int32_t on) {
    const int prev = g_gemm_h2p;
    g_gemm_h2p = on != 0;
    return prev;
}

int r4d_split2_lines_f16(const float* x_d, int64_t rows, int32_t K, uint16_t* lines_d, void* stream) {
    return launch_split2_lines(x_d, rows, K, lines_d, (hipStream_t)stream);
}

int r4d_conv1d_h2p_f32(const uint16_t* x_lines_d, const uint16_t* planes_d, const float* bias_d, const float* residual_d, int32_t M,
                       int32_t K, int32_t N, int32_t epilogue, int32_t out_lines, void* y_d, void* stream) {
    R4D_REQUIRE(epilogue >= 0 && epilogue <= 2, "conv1d_h2p: epilogue %d not in {0,1,2}", epilogue);
    S3Args a;
    memset(&a, 0, sizeof(a));
    a.planes = planes_d; a.C = (float*)y_d; a.bias = bias_d; a.resid = residual_d;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N; a.ldr = N; a.epilogue = epilogue;
    return launch_gemm_h2p(a, x_lines_d, out_lines != 0, (hipStream_t)stream);
}

}  // extern "C"
