// Retrieval scoring on gfx950: row normalisation and the query-vs-pool cosine scan, handing the score rows to the one-launch
// top-k of topk.hip.  Three forms of the scan, bit-identical in their scores: pool_scan_ks_kernel (operands straight from
// global memory into MFMA registers: the exact-f32 mode and d <= 128), pool_scan_dma_kernel (bf16x3 operands, LDS-DMA staged:
// shards of up to 64 rows per CU) and pool_scan_ring_kernel (the same staging as a stream: longer shards).
// Reference: train/train_retriever.py:433-438 (normalise, matmul, (x+1)/2) and :357-358,461-467 (argsort).
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "common.h"

#ifndef SCAN_RING_P
#define SCAN_RING_P 3   // ring slots per wavefront of the long-shard form (tuning aid: 4 = deeper ring, single-buffered partial tiles)
#endif
#ifndef SCAN_DBG
#define SCAN_DBG 0   // tuning aid (tools/kc_ablate.sh score.hip SCAN_DBG n): bit 0 skips the MFMAs (and the split), bit 1 the score stores, bit 2 the cross-wave reduction (one-tile form), bit 3 writes s_memrealtime stamps (100 MHz) of workgroup phases over the score rows of queries >= 16 (tools/scan_timeline.py), bit 4 skips the bf16x3 split arithmetic only, bit 5 makes every query load read ONE line, bit 6 makes every pool load read 8 whole lines (timing of other access patterns)
#endif

namespace r4d {

int dbgflag_scan() { return SCAN_DBG != 0; }

// ------------------------------------------------------------------------------------ normalise rows
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, int n, int d,
                                                             float* __restrict__ out, unsigned* __restrict__ range_flag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = x + (long long)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += p[c] * p[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float nrm = sqrtf(s);                       // x / x.norm(dim=1, keepdim=True): no eps
    // range guard (r4d_set_range_flag): a row that cannot be normalised (NaN / inf element, overflowed or zero norm) would reach
    // the scan as NaN scores, which the selection orders below everything -- an ordinary-looking top-k of garbage
    if (range_flag && !(nrm > 0.f && nrm < __builtin_inff()) && lane == 0) atomicOr(range_flag, R4D_RANGE_BAD_NORM);
    for (int c = lane; c < d; c += 64) out[(long long)row * d + c] = p[c] / nrm;
}

// ------------------------------------------------------------------------------------ pool scan (Q <= 64)
// S[q, n] = (q_hat[q] . pool_hat[n] + 1) / 2 for one block of <= 32 queries against the whole pool shard:
// train_retriever.py:437-438 at the reference's query batch (32).
//
// At 32 queries the scan sits ON the fp32 ridge of the chip: 16 flop per pool byte is 21 us of exact-f32 MFMA issue at
// peak for a 100k x 512 pool against 25.6 us of HBM at peak, so the kernel has to keep BOTH pipes busy, which comes
// down to balance and to what shares the issue port with the MFMAs:
//   * one workgroup = KW wavefronts that SPLIT THE CONTRACTION of one 32-row pool tile (wave w owns the 128-byte lines
//     w, w + KW, ... of every row): a tile is d / (2 KW) MFMAs per SIMD instead of d / 2 on one, so the 391 tiles of a
//     12.5k-row shard become 1,564 wave-tiles and 3,125 tiles spread over 256 CUs to within one tile (the previous
//     tile-per-wave form left SIMDs with 4 tiles next to SIMDs with 2);
//   * the wave's slice of the 32 normalised queries lives in REGISTERS for the whole kernel (NG x 16 VGPRs): no LDS read
//     in front of the MFMAs at all -- the loop body is 16 MFMAs + 4 global loads per 128-byte line group;
//   * lane (j, h) streams row j of the tile with 16-byte loads, the lane pair of a row consuming each 128-byte line at
//     once; a group's registers are refilled with the NEXT tile's data right after its MFMAs (program order, no copies);
//   * the KW partial 32 x 32 tiles meet in LDS (double-buffered, ONE barrier per tile), wave w finishing 16 / KW of the
//     accumulator registers in a fixed order (deterministic sums); the epilogue applies (x+1)/2 and writes 128-byte row
//     segments of the score matrix, which topk_chunk_kernel then reduces from the Infinity Cache in one launch.
// Measured (MI355X, 100k x 512, 32 queries; tools/bench_components.py, SCAN_DBG builds): 45-46 us = 4.7 TB/s = 0.59-0.61 of
// the 8 TB/s spec; the same kernel WITHOUT its MFMAs streams the pool in 34.5 us (6.3 TB/s, the measured HBM ceiling of
// this part), the MFMAs alone are 21 us: the two pipes overlap to ~80 %.  PMC: FETCH_SIZE == the algorithmic bytes (no
// re-reads), MFMA pipe 46 % busy.  Tried without gain: two tiles in flight per wave (0.58), 16-byte pieces 64 bytes apart
// per load instead of whole 32-byte sectors (0.56-0.59), 4-way split with three workgroups per CU (0.56; 0.60 with two).
// Fusing the selection into this epilogue was priced and NOT done: a score costs 16 SIMD-cycles of MFMA here, an exact
// per-workgroup top-k costs ~2.3 more on the same issue port (+14 %), which buys nothing over the separate launch while
// the raw rows (6 % of the pool bytes) stay cache-resident.
//
// S3 (the default, r4d_set_gemm_split3): the same kernel on the bf16 matrix cores at fp32 accuracy -- both operands written
// as exact sums of three bf16 numbers (the queries once, into registers; a pool row's 8 consecutive k of a lane right before
// its MFMAs: 11 VALU instructions per two elements), six v_mfma_f32_32x32x16_bf16 per 16 k instead of eight
// v_mfma_f32_32x32x2_f32: 2.67x fewer matrix-pipe cycles per pool byte, which takes the scan OFF the fp32 ridge (the timeline
// of tools/scan_timeline.py at a 12.5k-row shard: 4 of 11 us were exact-f32 MFMA on two tiles per CU).  The arithmetic of a
// score does not depend on the tile, the workgroup or the shard the row falls in (shard merge == single GPU bit for bit).
typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned scan_cvt_pk(float a, float b) {   // v_cvt_pk_bf16_f32: low half = bf16(a), RNE
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
// two fp32 -> packed (hi, hi), (mid, mid), (lo, lo): x == hi + mid + lo exactly (gemm_s3.hip)
__device__ __forceinline__ void scan_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = scan_cvt_pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = scan_cvt_pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = scan_cvt_pk(s0, s1);
}
// a lane's 8 k of one bf16 MFMA step (two 16-byte pieces) -> three bf16x8 operands
__device__ __forceinline__ void scan_split8(const float4& a, const float4& b, u32x4v& h, u32x4v& m, u32x4v& l) {
    unsigned hh[4], mm[4], ll[4];
    scan_split_pair(a.x, a.y, hh[0], mm[0], ll[0]); scan_split_pair(a.z, a.w, hh[1], mm[1], ll[1]);
    scan_split_pair(b.x, b.y, hh[2], mm[2], ll[2]); scan_split_pair(b.z, b.w, hh[3], mm[3], ll[3]);
    h = u32x4v{hh[0], hh[1], hh[2], hh[3]}; m = u32x4v{mm[0], mm[1], mm[2], mm[3]}; l = u32x4v{ll[0], ll[1], ll[2], ll[3]};
}
// the six products of one 16-k step, smallest terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi (gemm_s3.hip)
#define SCAN_STEP6(ACC_, Q_, BH_, BM_, BL_) do { SCAN_MFMA16(Q_[2], BH_, ACC_); SCAN_MFMA16(Q_[0], BL_, ACC_); SCAN_MFMA16(Q_[1], BM_, ACC_); \
    SCAN_MFMA16(Q_[1], BH_, ACC_); SCAN_MFMA16(Q_[0], BM_, ACC_); SCAN_MFMA16(Q_[0], BH_, ACC_); } while (0)
#define SCAN_MFMA16(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, A_), __builtin_bit_cast(bf16x8v, B_), ACC_, 0, 0, 0)

template <int KW, int NG, bool TWO, bool S3>            // d == 32 * KW * NG; TWO: two tiles of a workgroup in flight; S3: bf16x3
__global__ __launch_bounds__(64 * KW, (KW == 4 && NG == 4) ? 3 : 1) void pool_scan_ks_kernel(const float* __restrict__ qhat, const float* __restrict__ pool,
                                                              int Q, int N, int rpw, float* __restrict__ scores,
                                                              unsigned* __restrict__ zero_d, int nzero) {
    constexpr int D = 32 * KW * NG;
    constexpr int R = 16 / KW;                          // accumulator registers a wave finishes
    __shared__ float red[KW > 1 ? 2 * KW * KW * 64 * R : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.y * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0)             // ticket counters of the top-k launch that follows on this stream
        for (int i = tid; i < nzero; i += 64 * KW) zero_d[i] = 0u;
    // rpw > 0: workgroup b owns the CONTIGUOUS rows [b rpw, (b+1) rpw) of the shard, walked in 32-row tiles (the last one
    // partial): rows, i.e. bytes, are dealt evenly -- a 12,500-row shard is 49 rows (100 KB) per CU instead of 391 whole
    // tiles over 256 CUs (two tiles on half of them, one on the rest).  rpw == 0 (long shards): the 32-row tiles are dealt
    // round-robin (tile b, b + G, ...), so that at any moment the workgroups read ADJACENT lines (DRAM page locality:
    // contiguous ranges measured 48.9 us against 43-45 us at 100k x 512).
    const int G = (int)gridDim.x;
    const int r0 = rpw > 0 ? blockIdx.x * rpw : blockIdx.x * 32;
    const int r1 = rpw > 0 ? min(N, r0 + rpw) : N;
    const int tstep = rpw > 0 ? 32 : 32 * G;                 // rows between consecutive tiles of this workgroup
    if (r0 >= N) return;
#if SCAN_DBG & 8
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(scores + (long long)16 * N) + blockIdx.x * 16;
    if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); stamps[5] = __builtin_amdgcn_s_memtime(); stamps[10] = gridDim.x; }
    unsigned long long* wst = reinterpret_cast<unsigned long long*>(scores + (long long)16 * N) + gridDim.x * 16 + (blockIdx.x * KW + w) * 8;   // per wavefront
    if (lane == 0) wst[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int ntiles = rpw > 0 ? (r1 - r0 + 31) / 32 : ((N + 31) / 32 - (int)blockIdx.x + G - 1) / G;
    // k order: load u of lane half h of row j is the 16-byte piece 2u + h of the wave's 128-byte line g, so ONE load
    // instruction touches a whole 32-byte sector of each of its 32 rows; component c of that load is
    // k = 32 (g KW + w) + 8 u + 4 h + c on BOTH operands.
#if SCAN_DBG & 64                                       // WRONG results: every load instruction reads 8 whole 128-byte lines (timing of the coalesced pattern)
#define SCAN_TIDX(g, u) (8 * KW * (g) + (u) * 8 * (D / 4))
    auto tile_ptr = [&](int tt) {
        return reinterpret_cast<const float4*>(pool + (long long)min(r0 + tt * tstep + (lane >> 3), N - 25) * D) + (lane & 7) + 8 * w;
    };
#else
#define SCAN_TIDX(g, u) (8 * KW * (g) + 2 * (u))
    auto tile_ptr = [&](int tt) {                       // rows past the shard are clamped (always valid); rows past r1 belong to the
        return reinterpret_cast<const float4*>(pool + (long long)min(r0 + tt * tstep + li, N - 1) * D) + lh + 8 * w;   // next workgroup: computed, not stored
    };
#endif
    auto load_tile = [&](float4 (&bb)[NG][4], int tt) {
        const float4* __restrict__ p = tile_ptr(tt);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) bb[g][u] = p[SCAN_TIDX(g, u)];
    };
    float4 b0[NG][4];
    load_tile(b0, 0);                                   // the HBM stream starts BEFORE the (L2-resident) queries pass the CU's 64 B/clk vector-memory path (0.5 us for 64 KB)
    // S3: MFMA step s of line g takes loads 2s and 2s + 1 of the lane as its 8 consecutive "k" (the same bijection on both
    // operands), q3[g][s][plane] holding the query side split once
    float4 qf[NG][4];
    u32x4v q3[S3 ? NG : 1][2][3];
    float4 b1[TWO ? NG : 1][4];
    {
        const bool ok = q0 + li < Q;
        const float4* src = reinterpret_cast<const float4*>(qhat + (long long)((SCAN_DBG & 32) ? 0 : min(q0 + li, Q - 1)) * D) + ((SCAN_DBG & 32) ? 0 : lh);   // bit 5: one line per query load (WRONG results)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) qf[g][u] = src[8 * (g * KW + w) + 2 * u];
        if constexpr (TWO) {                            // the second tile's loads go out before anything waits
            const float4* __restrict__ p1 = tile_ptr(min(1, ntiles - 1));
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int u = 0; u < 4; ++u) b1[g][u] = p1[SCAN_TIDX(g, u)];
        }
        __builtin_amdgcn_sched_barrier(0);              // ... and the compiler does not sink them below the query split
#if SCAN_DBG & 8
        if (lane == 0) wst[1] = __builtin_amdgcn_s_memrealtime();
        asm volatile("" :: "v"(qf[NG - 1][3].w));
        if (lane == 0) wst[2] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (!ok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { qf[g][u].x = 0.f; qf[g][u].y = 0.f; qf[g][u].z = 0.f; qf[g][u].w = 0.f; }
            }
            if constexpr (S3) {
                scan_split8(qf[g][0], qf[g][1], q3[g][0][0], q3[g][0][1], q3[g][0][2]);
                scan_split8(qf[g][2], qf[g][3], q3[g][1][0], q3[g][1][1], q3[g][1][2]);
            }
        }
    }
#if SCAN_DBG & 8
    if constexpr (S3) { asm volatile("" :: "v"(q3[NG - 1][1][2])); if (lane == 0) wst[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // the MFMAs of one 128-byte line group of the staged rows
    auto mfma_group = [&](f32x16s& acc, int g, const float4 (&b)[4]) {
        if constexpr ((SCAN_DBG & 1) != 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += b[u].x + b[u].y + b[u].z + b[u].w;
        } else if constexpr (S3) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4v bh, bm, bl;
                if constexpr ((SCAN_DBG & 16) != 0) {
                    bh = __builtin_bit_cast(u32x4v, b[2 * st]); bm = __builtin_bit_cast(u32x4v, b[2 * st + 1]); bl = bh ^ bm;
                } else {
                    scan_split8(b[2 * st], b[2 * st + 1], bh, bm, bl);
                }
                SCAN_MFMA16(q3[g][st][2], bh, acc);                      // smallest terms first: lo.hi, hi.lo, mid.mid,
                SCAN_MFMA16(q3[g][st][0], bl, acc);                      // mid.hi, hi.mid, hi.hi (gemm_s3.hip)
                SCAN_MFMA16(q3[g][st][1], bm, acc);
                SCAN_MFMA16(q3[g][st][1], bh, acc);
                SCAN_MFMA16(q3[g][st][0], bm, acc);
                SCAN_MFMA16(q3[g][st][0], bh, acc);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].x, b[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].y, b[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].z, b[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].w, b[u].w, acc, 0, 0, 0);
            }
        }
    };
    int buf = 0;
    // one tile: MFMAs over the staged rows `bb`, each group's registers refilled with tile `trefill` (< 0: none) right after
    // its MFMAs (program order, no copies), cross-wave reduction, (x+1)/2, store
    auto do_tile = [&](float4 (&bb)[NG][4], int tcur, int trefill) {
        const float4* __restrict__ pn = tile_ptr(trefill >= 0 ? trefill : tcur);
        f32x16s acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            mfma_group(acc, g, bb[g]);
            if (trefill >= 0) {                                          // wave-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u) bb[g][u] = pn[SCAN_TIDX(g, u)];
            }
        }
#if SCAN_DBG & 8
        if (tcur < 2) { asm volatile("" :: "v"(acc[0]), "v"(acc[15])); if (tid == 0) stamps[8 + 4 * tcur] = __builtin_amdgcn_s_memrealtime(); }
#endif
        float fin[R];
        if (KW > 1 && !(SCAN_DBG & 4)) {
            // red[buf][src wave][dst wave][lane][R]: lane-contiguous vectors, conflict-free on both sides
            float* base = red + buf * (KW * KW * 64 * R);
#pragma unroll
            for (int wd = 0; wd < KW; ++wd)
#pragma unroll
                for (int i = 0; i < R; ++i) base[((w * KW + wd) * 64 + lane) * R + i] = acc[wd * R + i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < R; ++i) fin[i] = 0.f;
#pragma unroll
            for (int p = 0; p < KW; ++p)                                 // fixed order: deterministic sums
#pragma unroll
                for (int i = 0; i < R; ++i) fin[i] += base[((p * KW + w) * 64 + lane) * R + i];
            buf ^= 1;
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) fin[i] = acc[w * R + i];
        }
#if SCAN_DBG & 8
        if (tcur < 2) { asm volatile("" :: "v"(fin[0])); if (tid == 0) stamps[9 + 4 * tcur] = __builtin_amdgcn_s_memrealtime(); }
#endif
        const int row = r0 + tcur * tstep + li;
        if (row < r1 && !(SCAN_DBG & 2)) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = w * R + i;
                const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (q < Q && (!(SCAN_DBG & 8) || q < 16)) scores[(long long)q * N + row] = (fin[i] + 1.0f) / 2.0f;
            }
        }
    };
#if SCAN_DBG & 8
    if (tid == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();              // all loads issued
    {   // wait for the first tile and the queries
        float sink = qf[0][0].x + b0[NG - 1][3].w;
        asm volatile("" :: "v"(sink));
        if (tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();          // first tile landed
    }
#endif
    if constexpr (TWO) {
        // short ranges (a shard of a few ten thousand rows: 2-3 tiles per workgroup): the second tile's loads go out together
        // with the first's -- ONE memory latency in front of the MFMAs instead of one per tile -- and the two tiles share ONE
        // cross-wave reduction (timeline of the one-tile-at-a-time form, tools/scan_timeline.py: the barrier of the first
        // tile's reduction waits 1.8 us for the wave whose loads landed last, while the matrix pipe idles)
        for (int t = 0; t < ntiles; t += 2) {
            const bool has2 = t + 1 < ntiles;                             // workgroup-uniform
            f32x16s acc2[2];
            if constexpr (S3 && !(SCAN_DBG & 17)) {
                // The split of a step's pool operand (36 VALU instructions) is issued INSIDE the six MFMAs of the step before
                // it, six per MFMA gap (sched_group_barrier): left to the compiler the stream is split, split, ..., MFMA x 6,
                // and two wavefronts of a SIMD fall into lock step -- both splitting, then both multiplying -- so that the
                // vector ALU and the matrix pipe take turns (timeline: 4,600 cycles per tile and SIMD = 48 x 32 + 704 x 4.2).
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
                u32x4v ch, cm, cl;
                scan_split8(b0[0][0], b0[0][1], ch, cm, cl);
#pragma unroll
                for (int i = 0; i < 2 * NG; ++i) {
                    const int g = i >> 1, st = i & 1, gn = (i + 1) >> 1, un = 2 * ((i + 1) & 1);
                    u32x4v nh, nm, nl;
                    if (i + 1 < 2 * NG) scan_split8(b0[gn][un], b0[gn][un + 1], nh, nm, nl);
                    else scan_split8(b1[0][0], b1[0][1], nh, nm, nl);
                    SCAN_STEP6(acc2[0], q3[g][st], ch, cm, cl);
#pragma unroll
                    for (int j = 0; j < 6; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
                    ch = nh; cm = nm; cl = nl;
                }
                if (t + 2 < ntiles) load_tile(b0, t + 2);
#if SCAN_DBG & 8
                if (t == 0) { asm volatile("" :: "v"(acc2[0][0])); if (lane == 0) wst[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
                if (has2) {
#pragma unroll
                    for (int i = 0; i < 2 * NG; ++i) {
                        const int g = i >> 1, st = i & 1, gn = (i + 1) >> 1, un = 2 * ((i + 1) & 1);
                        u32x4v nh, nm, nl;
                        if (i + 1 < 2 * NG) scan_split8(b1[gn][un], b1[gn][un + 1], nh, nm, nl);
                        SCAN_STEP6(acc2[1], q3[g][st], ch, cm, cl);
                        if (i + 1 < 2 * NG) {
#pragma unroll
                            for (int j = 0; j < 6; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
                            ch = nh; cm = nm; cl = nl;
                        }
                    }
                    if (t + 3 < ntiles) load_tile(b1, t + 3);
                }
#if SCAN_DBG & 8
                if (t == 0) { asm volatile("" :: "v"(acc2[0][0]), "v"(acc2[1][15])); if (tid == 0) { stamps[8] = __builtin_amdgcn_s_memrealtime(); stamps[12] = stamps[8]; }
                              if (lane == 0) wst[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 (&bb)[NG][4] = h ? b1 : b0;
                const int tn = t + 2 + h;
                const float4* __restrict__ pn = tile_ptr(tn < ntiles ? tn : 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[h][r] = 0.f;
                if (h == 0 || has2) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        mfma_group(acc2[h], g, bb[g]);
                        if (tn < ntiles) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) bb[g][u] = pn[SCAN_TIDX(g, u)];
                        }
                    }
                }
#if SCAN_DBG & 8
                if (t == 0) { asm volatile("" :: "v"(acc2[h][0]), "v"(acc2[h][15])); if (tid == 0) stamps[8 + 4 * h] = __builtin_amdgcn_s_memrealtime(); }
#endif
            }
            // red[tile h][src wave][dst wave][lane][R]; same fixed summation order as the one-tile form: identical bits
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* base = red + h * (KW * KW * 64 * R);
#pragma unroll
                for (int wd = 0; wd < KW; ++wd)
#pragma unroll
                    for (int i = 0; i < R; ++i) base[((w * KW + wd) * 64 + lane) * R + i] = acc2[h][wd * R + i];
            }
            __syncthreads();
#if SCAN_DBG & 8
            if (t == 0 && tid == 0) { stamps[9] = __builtin_amdgcn_s_memrealtime(); stamps[13] = stamps[9]; }
            if (t == 0 && lane == 0) wst[6] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* base = red + h * (KW * KW * 64 * R);
                float fin[R];
#pragma unroll
                for (int i = 0; i < R; ++i) fin[i] = 0.f;
#pragma unroll
                for (int p = 0; p < KW; ++p)
#pragma unroll
                    for (int i = 0; i < R; ++i) fin[i] += base[((p * KW + w) * 64 + lane) * R + i];
                const int row = r0 + (t + h) * tstep + li;
                if (row < r1 && (h == 0 || has2)) {
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int r = w * R + i;
                        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (q < Q && (!(SCAN_DBG & 8) || q < 16) && !(SCAN_DBG & 2)) scores[(long long)q * N + row] = (fin[i] + 1.0f) / 2.0f;
                    }
                }
            }
            if (t + 2 < ntiles) __syncthreads();                          // the LDS buffers are reused by the next pair
        }
    } else {
        for (int t = 0; t < ntiles; ++t) do_tile(b0, t, t + 1 < ntiles ? t + 1 : -1);
    }
#if SCAN_DBG & 8
    if (tid == 0) stamps[7] = __builtin_amdgcn_s_memrealtime();              // all stores issued
    if (lane == 0) wst[7] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { stamps[3] = __builtin_amdgcn_s_memrealtime(); stamps[6] = __builtin_amdgcn_s_memtime(); stamps[4] = (unsigned long long)__builtin_amdgcn_s_getreg(6164); }   // end; XCC id
#endif
}


// ------------------------------------------------------------------------------------ short shards: LDS-DMA staged form
// A shard of 8k-16k rows is ONE cold burst: every CU reads its <= 64 rows (two 32-row tiles) plus the 32 queries once, and
// the kernel is as long as that burst.  The per-wavefront timeline of the register-staged form above (tools/scan_timeline.py)
// showed what the burst costs there: a load instruction whose lane (row j, half h) reads 32 bytes of ITS row touches 32
// cache lines, the CU's vector-memory path takes ~40 cycles for it, and the second wavefront of every SIMD gets its loads
// issued 2.7 us after the first (stream complete at 8.3 us); with 8 whole 128-byte lines per instruction the same bytes are
// issued by 2.8 us and landed by 6 us.  Whole lines per instruction and row-per-lane MFMA operands need a transposition, so
// the rows go global -> LDS by DMA (global_load_lds_dwordx4: lane l of instruction i copies 16 bytes of row 8 i + l / 8,
// no VGPRs) and come back as ds_read_b128 fragments:
//   * wavefront w still owns the 128-byte lines w, w + KW, ... of every row, so it reads back ONLY what it copied itself:
//     its own counted s_waitcnt vmcnt orders the DMA in front of the ds_read (no barrier until the cross-wave reduction);
//   * a "piece" = 32 rows x one line = 4 KB = four DMA instructions; pieces run through a ring of P = 4 slots per
//     wavefront (16 KB; 128 KB per workgroup at KW = 8): the queries' NG pieces first, then tile 0's, then tile 1's, a
//     slot refilled as soon as its fragments are in registers (s_waitcnt lgkmcnt(0) in front of the DMA);
//   * LDS image: row r's 16-byte chunk c sits at slot c ^ ((r >> 1) & 7) of its 128 bytes -- written by permuting the
//     SOURCE chunk per lane (the DMA destination is lane-linear), read with the same XOR: both sides conflict-free;
//   * rows past the workgroup's share are CLAMPED to its last row (cache hits), so a 49-row share reads 49 rows from HBM,
//     not two whole tiles (the form above reads 64);
//   * arithmetic: the same chunk -> MFMA-slot bijection, the same six products in the same order and the same reduction
//     order as pool_scan_ks_kernel<.., S3 = true>: scores are bit-identical to the long-shard form (shard merge == one GPU).
// one piece = four DMA instructions of 1 KB: global address = base + 32-bit lane offset + i KB, LDS address = M0 + i KB + 16 lane
// (the instruction offset moves BOTH sides, so the lane offsets of instruction i are built i KB short); M0 is written in the
// statement that uses it and restored (the compiler owns it)
__device__ __forceinline__ void scan_glds_piece(const void* base, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "s"(lds_dst) : "memory");
}
template <int N_> __device__ __forceinline__ void scan_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }
__device__ __forceinline__ void scan_wait_vmcnt(int n) {                                   // n is a constant after unrolling
    switch (n) { case 0: scan_wait_vm<0>(); break; case 4: scan_wait_vm<4>(); break; case 8: scan_wait_vm<8>(); break;
                 default: scan_wait_vm<12>(); break; }
}
template <int N_> struct ScanInt { static constexpr int value = N_; };

template <int KW, int NG>
__global__ __launch_bounds__(64 * KW, 1) void pool_scan_dma_kernel(const float* __restrict__ qhat, const float* __restrict__ pool,
                                                                  int Q, int N, int rpw, float* __restrict__ scores,
                                                                  unsigned* __restrict__ zero_d, int nzero) {
    constexpr int D = 32 * KW * NG;
    constexpr int R = 16 / KW;
    constexpr int P = 4;                                  // ring slots of 4 KB per wavefront
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.y * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = tid; i < nzero; i += 64 * KW) zero_d[i] = 0u;
    const int r0 = blockIdx.x * rpw, r1 = min(N, r0 + rpw);              // rpw <= 64: one or two tiles
    if (r0 >= N) return;
#if SCAN_DBG & 8
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(scores + (long long)16 * N) + blockIdx.x * 16;
    unsigned long long* wst = reinterpret_cast<unsigned long long*>(scores + (long long)16 * N) + gridDim.x * 16 + (blockIdx.x * KW + w) * 8;
    if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); stamps[5] = __builtin_amdgcn_s_memtime(); stamps[10] = gridDim.x; }
    if (lane == 0) wst[0] = __builtin_amdgcn_s_memrealtime();
#define SCAN_WSTAMP(i) do { if (lane == 0) wst[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SCAN_WSTAMP(i) do { } while (0)
#endif
    unsigned char* ring = scan_lds + w * (P * 4096);
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
    // DMA sources: instruction i of a piece copies rows 8 i + lane / 8 of the piece, lane % 8 picking the (permuted) chunk:
    // byte offsets from (pointer - 3 KB), instruction i's built i KB short (scan_glds_piece); rows past the share / the
    // query block are clamped to its last row
    unsigned voff[3][4];                                  // queries, tile 0, tile 1
    {
        const int c0 = (lane & 7) ^ (lane >> 4);          // chunk of instruction i: c0 ^ 4 (i & 1)   [= (lane & 7) ^ ((row >> 1) & 7)]
        const int lrow = lane >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned in_row = 128u * w + 16u * (c0 ^ (4 * (i & 1))) + 3072u - 1024u * i;
            voff[0][i] = (unsigned)min(q0 + 8 * i + lrow, Q - 1) * (unsigned)(D * 4) + in_row;
            voff[1][i] = (unsigned)min(r0 + 8 * i + lrow, r1 - 1) * (unsigned)(D * 4) + in_row;
            voff[2][i] = (unsigned)min(r0 + 32 + 8 * i + lrow, r1 - 1) * (unsigned)(D * 4) + in_row;
        }
    }
    const char* qbase = reinterpret_cast<const char*>(qhat) - 3072;
    const char* pbase = reinterpret_cast<const char*>(pool) - 3072;
    auto issue = [&](int p) {                             // piece p: kind p / NG, line group p % NG -> ring slot p % P
        const int kind = p / NG;
        scan_glds_piece((kind == 0 ? qbase : pbase) + 128 * KW * (p % NG), voff[kind][0], voff[kind][1], voff[kind][2], voff[kind][3],
                        ring_lds + (p % P) * 4096);
    };
    const int fsw = (li >> 1) & 7;
    auto read_piece = [&](int p, float4 (&f)[4]) {        // lane (row li, half lh): chunks lh, 2 + lh, 4 + lh, 6 + lh of its row
        const float4* base = reinterpret_cast<const float4*>(ring + (p % P) * 4096 + li * 128);
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = base[(2 * u + lh) ^ fsw];
    };
    f32x16s acc2[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
    const bool has2 = r1 - r0 > 32;                       // workgroup-uniform
    auto body = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;       // tiles of this workgroup
        constexpr int NP = NG * (1 + NT);                 // pieces: queries, tile 0, (tile 1)
        // pieces outstanding BEHIND piece k when it is waited for: min(NP, k + P) have been issued by then
#define SCAN_BEHIND(k) (4 * (((NP) < (k) + P ? (NP) : (k) + P) - (k) - 1))
#pragma unroll
        for (int p = 0; p < P && p < NP; ++p) issue(p);
        SCAN_WSTAMP(1);
        u32x4v q3[NG][2][3];
        const bool ok = q0 + li < Q;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 f[4];
            scan_wait_vmcnt(SCAN_BEHIND(g));
            if (g == NG - 1) SCAN_WSTAMP(2);
            read_piece(g, f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (g + P < NP) issue(g + P);
            if (!ok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { f[u].x = 0.f; f[u].y = 0.f; f[u].z = 0.f; f[u].w = 0.f; }
            }
            scan_split8(f[0], f[1], q3[g][0][0], q3[g][0][1], q3[g][0][2]);
            scan_split8(f[2], f[3], q3[g][1][0], q3[g][1][1], q3[g][1][2]);
        }
#if SCAN_DBG & 8
        asm volatile("" :: "v"(q3[NG - 1][1][2]));
        SCAN_WSTAMP(3);
#endif
        constexpr int NTP = NG * NT;                      // tile pieces; piece NG + tp = (tile tp / NG, line group tp % NG)
        float4 cur[4], nxt[4];
        scan_wait_vmcnt(SCAN_BEHIND(NG));
        read_piece(NG, cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NG + P < NP) issue(NG + P);
        u32x4v ch, cm, cl;
        scan_split8(cur[0], cur[1], ch, cm, cl);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) {
            const int p = NG + tp, g = tp % NG, h = tp / NG;
            u32x4v nh, nm, nl;
            scan_split8(cur[2], cur[3], nh, nm, nl);                         // ... inside the six MFMAs of the step before it
            SCAN_STEP6(acc2[h], q3[g][0], ch, cm, cl);
#pragma unroll
            for (int j = 0; j < 6; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
            if (tp + 1 < NTP) {
                scan_wait_vmcnt(SCAN_BEHIND(p + 1));
                read_piece(p + 1, nxt);
                scan_split8(nxt[0], nxt[1], ch, cm, cl);
            }
            SCAN_STEP6(acc2[h], q3[g][1], nh, nm, nl);
            if (tp + 1 < NTP) {
                // the next piece's fragments are still on their way from LDS: two MFMAs first, then the split in the gaps
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { __builtin_amdgcn_sched_group_barrier(0x002, 9, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (p + 1 + P < NP) issue(p + 1 + P);
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            }
#if SCAN_DBG & 8
            if (tp == NG - 1) { asm volatile("" :: "v"(acc2[0][0])); SCAN_WSTAMP(4); }
#endif
        }
#if SCAN_DBG & 8
        asm volatile("" :: "v"(acc2[0][0]), "v"(acc2[1][0]));
        SCAN_WSTAMP(5);
#endif
#undef SCAN_BEHIND
    };
    if (has2) body(ScanInt<2>{}); else body(ScanInt<1>{});
    // cross-wave reduction: wavefront w's partial tiles go into ITS OWN ring (all of its pieces are consumed), 4 KB per tile,
    // [dst wave][lane][R]; same fixed summation order as pool_scan_ks_kernel
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float* base = reinterpret_cast<float*>(ring + h * 4096);
#pragma unroll
        for (int wd = 0; wd < KW; ++wd)
#pragma unroll
            for (int i = 0; i < R; ++i) base[(wd * 64 + lane) * R + i] = acc2[h][wd * R + i];
    }
    __syncthreads();
    SCAN_WSTAMP(6);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float fin[R];
#pragma unroll
        for (int i = 0; i < R; ++i) fin[i] = 0.f;
#pragma unroll
        for (int p = 0; p < KW; ++p) {
            const float* base = reinterpret_cast<const float*>(scan_lds + p * (P * 4096) + h * 4096);
#pragma unroll
            for (int i = 0; i < R; ++i) fin[i] += base[(w * 64 + lane) * R + i];
        }
        const int row = r0 + 32 * h + li;
        if (row < r1) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = w * R + i;
                const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (q < Q && (!(SCAN_DBG & 8) || q < 16) && !(SCAN_DBG & 2)) scores[(long long)q * N + row] = (fin[i] + 1.0f) / 2.0f;
            }
        }
    }
#if SCAN_DBG & 8
    SCAN_WSTAMP(7);
    if (tid == 0) stamps[7] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { stamps[3] = __builtin_amdgcn_s_memrealtime(); stamps[6] = __builtin_amdgcn_s_memtime(); stamps[4] = (unsigned long long)__builtin_amdgcn_s_getreg(6164);
                    stamps[1] = stamps[0]; stamps[2] = stamps[0]; stamps[8] = stamps[9] = stamps[12] = stamps[13] = stamps[3]; }
#endif
#undef SCAN_WSTAMP
}


// ------------------------------------------------------------------------------------ long shards: the same staging as a ring
// pool_scan_dma_kernel's pieces (32 rows x one 128-byte line, four DMA instructions of whole lines, chunk-permuted source,
// ds_read_b128 fragments, split pinned into the MFMA gaps) as a STREAM: workgroup b walks the tiles b, b + G, ... of the shard
// (round-robin: at any moment the workgroups read adjacent lines), wavefront w keeps P = 3 pieces of its own lines in flight
// in a private ring (a slot is refilled as soon as its fragments are in registers), and after the NG pieces of a tile the KW
// partial tiles meet in LDS exactly as in pool_scan_ks_kernel (double-buffered, ONE barrier per tile -- a raw s_barrier behind
// lgkmcnt(0): __syncthreads() would also wait vmcnt(0) and drain the ring).  LDS: 96 KB ring + 64 KB partial tiles at KW = 8.
// Same chunk -> MFMA-slot bijection, product order and reduction order as every other form: bit-identical scores.
template <int KW, int NG>
__global__ __launch_bounds__(64 * KW, KW == 8 ? 1 : 2) void pool_scan_ring_kernel(const float* __restrict__ qhat, const float* __restrict__ pool,
                                                                                int Q, int N, float* __restrict__ scores,
                                                                                unsigned* __restrict__ zero_d, int nzero) {
    constexpr int D = 32 * KW * NG;
    constexpr int R = 16 / KW;
    constexpr int P = SCAN_RING_P;                        // 3: partial tiles double-buffered, one barrier per tile; 4: one buffer, two barriers
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.y * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = tid; i < nzero; i += 64 * KW) zero_d[i] = 0u;
    const int G = (int)gridDim.x;
    const int ntl = ((N + 31) / 32 - (int)blockIdx.x + G - 1) / G;       // tiles of this workgroup (>= 1: G <= tiles)
    const int NPt = NG * (1 + ntl);                                       // pieces: the queries', then NG per tile
    unsigned char* ring = scan_lds + w * (P * 4096);
    float* red = reinterpret_cast<float*>(scan_lds + KW * (P * 4096));
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring);
    const int c0 = (lane & 7) ^ (lane >> 4), lrow = lane >> 3;
    unsigned in_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) in_row[i] = 128u * w + 16u * (c0 ^ (4 * (i & 1))) + 3072u - 1024u * i;
    const char* qbase = reinterpret_cast<const char*>(qhat) - 3072;
    const char* pbase = reinterpret_cast<const char*>(pool) - 3072;
    // issue side: piece `ik` goes to slot `islot`; its lane offsets are the queries' or those of tile (ik - NG) / NG
    int ik = 0, islot = 0, ig = 0, itile = (int)blockIdx.x;
    unsigned ivoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ivoff[i] = (unsigned)min(q0 + 8 * i + lrow, Q - 1) * (unsigned)(D * 4) + in_row[i];
    auto issue_next = [&]() {                             // workgroup-uniform control flow
        if (ik >= NPt) return;
        scan_glds_piece((ik < NG ? qbase : pbase) + 128 * KW * ig, ivoff[0], ivoff[1], ivoff[2], ivoff[3], ring_lds + islot * 4096);
        ++ik;
        islot = islot + 1 == P ? 0 : islot + 1;
        if (++ig == NG) {
            ig = 0;
            if (ik > NG) itile += G;
#pragma unroll
            for (int i = 0; i < 4; ++i) ivoff[i] = (unsigned)min(itile * 32 + 8 * i + lrow, N - 1) * (unsigned)(D * 4) + in_row[i];
        }
    };
    const int fsw = (li >> 1) & 7;
    int ck = 0, cslot = 0;                                // consume side
    auto wait_piece = [&]() {                             // pieces issued behind piece ck: min(NPt, ck + P) - ck - 1
        const int behind = min(NPt, ck + P) - ck - 1;
        if (behind >= 3) scan_wait_vm<12>(); else if (behind == 2) scan_wait_vm<8>(); else if (behind == 1) scan_wait_vm<4>(); else scan_wait_vm<0>();
    };
    auto read_piece = [&](float4 (&f)[4]) {
        const float4* base = reinterpret_cast<const float4*>(ring + cslot * 4096 + li * 128);
#pragma unroll
        for (int u = 0; u < 4; ++u) f[u] = base[(2 * u + lh) ^ fsw];
        ++ck;
        cslot = cslot + 1 == P ? 0 : cslot + 1;
    };
#pragma unroll
    for (int p = 0; p < P; ++p) issue_next();
    u32x4v q3[NG][2][3];
    {
        const bool ok = q0 + li < Q;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 f[4];
            wait_piece();
            read_piece(f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_next();
            if (!ok) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { f[u].x = 0.f; f[u].y = 0.f; f[u].z = 0.f; f[u].w = 0.f; }
            }
            scan_split8(f[0], f[1], q3[g][0][0], q3[g][0][1], q3[g][0][2]);
            scan_split8(f[2], f[3], q3[g][1][0], q3[g][1][1], q3[g][1][2]);
        }
    }
    float4 cur[4], nxt[4];
    wait_piece();
    read_piece(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue_next();
    u32x4v ch, cm, cl;
    scan_split8(cur[0], cur[1], ch, cm, cl);
    int buf = 0;
    for (int j = 0; j < ntl; ++j) {
        f32x16s acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            u32x4v nh, nm, nl;
#if SCAN_DBG & 1
            acc[0] += cur[0].x + cur[1].y + cur[2].z + cur[3].w; nh = ch; nm = cm; nl = cl;
#else
            scan_split8(cur[2], cur[3], nh, nm, nl);
            SCAN_STEP6(acc, q3[g][0], ch, cm, cl);
#endif
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
            if (ck < NPt) {                               // workgroup-uniform: false only at the very last piece
                wait_piece();
                read_piece(nxt);
#if !(SCAN_DBG & 1)
                scan_split8(nxt[0], nxt[1], ch, cm, cl);
                SCAN_STEP6(acc, q3[g][1], nh, nm, nl);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { __builtin_amdgcn_sched_group_barrier(0x002, 9, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_next();
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            } else {
#if !(SCAN_DBG & 1)
                SCAN_STEP6(acc, q3[g][1], nh, nm, nl);
#endif
            }
        }
        // cross-wave reduction of the tile: red[buf][src wave][dst wave][lane][R], fixed summation order
        float* base = red + (P == 3 ? buf : 0) * (KW * KW * 64 * R);
#pragma unroll
        for (int wd = 0; wd < KW; ++wd)
#pragma unroll
            for (int i = 0; i < R; ++i) base[((w * KW + wd) * 64 + lane) * R + i] = acc[wd * R + i];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float fin[R];
#pragma unroll
        for (int i = 0; i < R; ++i) fin[i] = 0.f;
#pragma unroll
        for (int p = 0; p < KW; ++p)
#pragma unroll
            for (int i = 0; i < R; ++i) fin[i] += base[((p * KW + w) * 64 + lane) * R + i];
        buf ^= 1;
        if (P != 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the one buffer is rewritten by the next tile
        const int row = ((int)blockIdx.x + j * G) * 32 + li;
        if (row < N) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = w * R + i;
                const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (q < Q && !(SCAN_DBG & 2)) scores[(long long)q * N + row] = (fin[i] + 1.0f) / 2.0f;
            }
        }
    }
}

template <int KW, int NG, bool S3>
static int launch_scan_variant_p(const float* qhat, const float* pool, int Q, int N, float* scores, unsigned* zero_d, int nzero,
                               hipStream_t s) {
    static int wgs_per_cu = 0;
    if (wgs_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pool_scan_ks_kernel<KW, NG, false, S3>, 64 * KW, 0) != hipSuccess || nb < 1) nb = 1;
        wgs_per_cu = nb;
    }
    static int force = -1, two = -1;
    if (force < 0) { const char* e = getenv("R4D_SCAN_WGS_PER_CU"); force = e ? atoi(e) : 0; }     // tuning aid
    if (two < 0) { const char* e = getenv("R4D_SCAN_TWO"); two = e ? atoi(e) : 1; }                // tuning aid: 0 = never two tiles in flight
    const int per_cu = force > 0 ? force : wgs_per_cu;
    const int ntiles = cdiv(N, 32);
    // short shards (<= 3 tiles per CU): ONE workgroup per CU owning an equal, contiguous share of the rows, two tiles in
    // flight and one joint reduction; long ones: round-robin tiles over the occupancy-sized grid
    const bool shortr = two && KW * NG >= 8 && ntiles > 256 && ntiles <= 3 * 256;
    static int dma = -1;
    if (dma < 0) { const char* e = getenv("R4D_SCAN_DMA"); dma = e ? atoi(e) : 3; }                // tuning aid: bit 0 = LDS-DMA form for short shards, bit 1 = ring form for long ones (0: register-staged forms only)
    // the DMA forms copy 16-byte pieces and address the pool with 32-bit byte offsets: 16-byte aligned operands below 4 GB, else
    // the register-staged forms (any dword-aligned pointer, 64-bit addressing) take the call
    const bool dma_ok = ((reinterpret_cast<uintptr_t>(qhat) | reinterpret_cast<uintptr_t>(pool)) & 15u) == 0 &&
                        (unsigned long long)N * (32ull * KW * NG) * 4ull < (1ull << 32) && (unsigned long long)Q * (32ull * KW * NG) * 4ull < (1ull << 32);
    if (S3 && (dma & 1) && dma_ok && KW * NG >= 8 && ntiles <= 2 * 256) { // <= 64 rows per CU: the LDS-DMA staged form
        R4D_BRANCH(SCAN_DMA);
        constexpr int lds_bytes = KW * 4 * 4096;
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_scan_dma_kernel<KW, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
                set_error("pool_scan: cannot reserve %d bytes of LDS", lds_bytes);
                return R4D_ERR_HIP;
            }
            attr = true;
        }
        const int rpw = max(8, cdiv(N, 256));            // rows dealt evenly over the CUs (a share below 32 rows leaves MFMA columns idle, not CUs)
        hipLaunchKernelGGL((pool_scan_dma_kernel<KW, NG>), dim3(cdiv(N, rpw), cdiv(Q, 32)), dim3(64 * KW), lds_bytes, s, qhat, pool, Q, N,
                           rpw, scores, zero_d, nzero);
    } else if (S3 && (dma & 2) && dma_ok && KW * NG >= 8) {   // long shards: the ring form
        R4D_BRANCH(SCAN_RING);
        constexpr int lds_bytes = KW * SCAN_RING_P * 4096 + (SCAN_RING_P == 3 ? 2 : 1) * KW * KW * 64 * (16 / KW) * 4;
        static int ring_per_cu = 0;
        if (ring_per_cu == 0) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_scan_ring_kernel<KW, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
                set_error("pool_scan: cannot reserve %d bytes of LDS", lds_bytes);
                return R4D_ERR_HIP;
            }
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pool_scan_ring_kernel<KW, NG>, 64 * KW, lds_bytes) != hipSuccess || nb < 1) nb = 1;
            ring_per_cu = nb;
        }
        const int gx = max(1, min(ntiles, 256 * (force > 0 ? force : ring_per_cu)));
        hipLaunchKernelGGL((pool_scan_ring_kernel<KW, NG>), dim3(gx, cdiv(Q, 32)), dim3(64 * KW), lds_bytes, s, qhat, pool, Q, N, scores,
                           zero_d, nzero);
    } else if (shortr) {
        R4D_BRANCH(SCAN_SHORT);
        const int rpw = cdiv(N, 256);
        hipLaunchKernelGGL((pool_scan_ks_kernel<KW, NG, (KW * NG >= 8), S3>), dim3(cdiv(N, rpw), cdiv(Q, 32)), dim3(64 * KW), 0, s, qhat,
                           pool, Q, N, rpw, scores, zero_d, nzero);
    } else {
        const int gx = max(1, min(ntiles, 256 * per_cu));
        hipLaunchKernelGGL((pool_scan_ks_kernel<KW, NG, false, S3>), dim3(gx, cdiv(Q, 32)), dim3(64 * KW), 0, s, qhat, pool, Q, N, 0,
                           scores, zero_d, nzero);
    }
    R4D_CHECK_LAUNCH("pool_scan");
    return R4D_OK;
}

template <int KW, int NG>
static int launch_scan_variant(const float* qhat, const float* pool, int Q, int N, float* scores, unsigned* zero_d, int nzero,
                               hipStream_t s) {
    if (g_gemm_split3) { R4D_BRANCH(SCAN_BF16X3); return launch_scan_variant_p<KW, NG, true>(qhat, pool, Q, N, scores, zero_d, nzero, s); }
    R4D_BRANCH(SCAN_F32);
    return launch_scan_variant_p<KW, NG, false>(qhat, pool, Q, N, scores, zero_d, nzero, s);
}

// d -> (KW, NG) with d == 32 * KW * NG; +1 = no instantiation (the tiled GEMM takes the call)
static int launch_pool_scan(const float* qhat, const float* pool, int Q, int N, int d, float* scores, unsigned* zero_d,
                            int nzero, hipStream_t s) {
    static int kw8 = -1;
    if (kw8 < 0) { const char* e = getenv("R4D_SCAN_KW8"); kw8 = e ? atoi(e) : 1; }               // tuning aid: 0 = 4-way split at d 512
    if (d != 32 && d != 64 && d != 128 && d != 256 && d != 384 && d != 512 && d != 768 && d != 1024) return 1;   // no variant
    // algorithmic bytes (SURVEY 8d B_score): pool read once per 32 queries + queries + score rows out
    ProfScope prof(PK_POOL_SCAN, 4.0 * N * d * cdiv(Q, 32) + 4.0 * Q * d + 4.0 * (double)Q * N, s);
    // Q >= 64 on the split arithmetic: 128 x 256 tiles of gemm_s3.hip in THIS file's arithmetic (same slices, same k slots, same
    // products, same order: bit-identical scores) -- the pool is read Q / 128 times instead of Q / 32 times and an element is split once
    // per tile instead of once per wavefront that touches it (round 5; R4D_SCAN_TILED=0 keeps the 32-query blocks for every Q)
    static int tiled = -1;
    if (tiled < 0) { const char* e = getenv("R4D_SCAN_TILED"); tiled = e ? atoi(e) : 1; }
    // (measured, tools/scan_q_bench.py, scan + top-10, d 512: 256 x 100k 347 -> 248 us, 2,048 x 12,512 417 -> 232 us, 128 x 100k 173 -> 128 us;
    //  below ~0.75 tiles per CU the blocks win: 64 x 12,512 is 23 us in blocks, 53 us as 49 tiles)
    if (tiled && Q >= 64 && g_gemm_split3 && (long long)cdiv(Q, 128) * cdiv(N, 256) >= 192) {
        const int ng = d == 256 ? 2 : d == 384 ? 3 : d == 512 ? (kw8 ? 2 : 4) : d == 768 ? 3 : d == 1024 ? 4 : 1;
        if (nzero > 0) R4D_HIP(hipMemsetAsync(zero_d, 0, (size_t)nzero * sizeof(unsigned), s));     // the scan kernels clear the top-k ticket counters themselves
        const int rc = launch_gemm_s3_scan_order(qhat, pool, scores, Q, N, d, ng, s);
        if (rc <= 0) return rc;
    }
    switch (d) {
        case 32:   R4D_BRANCH(SCAN_1_1); return launch_scan_variant<1, 1>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 64:   R4D_BRANCH(SCAN_2_1); return launch_scan_variant<2, 1>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 128:  R4D_BRANCH(SCAN_4_1); return launch_scan_variant<4, 1>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 256:  R4D_BRANCH(SCAN_4_2); return launch_scan_variant<4, 2>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 384:  R4D_BRANCH(SCAN_4_3); return launch_scan_variant<4, 3>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 512:  if (kw8) { R4D_BRANCH(SCAN_8_2); return launch_scan_variant<8, 2>(qhat, pool, Q, N, scores, zero_d, nzero, s); }   // default
                   R4D_BRANCH(SCAN_4_4); return launch_scan_variant<4, 4>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 768:  R4D_BRANCH(SCAN_8_3); return launch_scan_variant<8, 3>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        case 1024: R4D_BRANCH(SCAN_8_4); return launch_scan_variant<8, 4>(qhat, pool, Q, N, scores, zero_d, nzero, s);
        default:   return 1;
    }
}

}  // namespace r4d

using namespace r4d;

extern "C" {

int r4d_normalize_rows_f32(const float* x_d, int32_t n, int32_t d, float* out_d, void* stream) {
    R4D_REQUIRE(x_d && out_d && n >= 0 && d >= 1, "normalize_rows: bad arguments");
    if (n == 0) return R4D_OK;
    ProfScope prof(PK_NORMALIZE, 8.0 * n * d, (hipStream_t)stream);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x_d, n, d, out_d, g_range_flag);
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
    const size_t skip = align_up((size_t)Q * N * sizeof(float), 256);
    char* ws = (char*)workspace_d + skip;                                        // starts with the top-k ticket counters
    int rc = 1;
    bool zeroed = false;
    // ONE scoring arithmetic: whenever d has a scan variant EVERY call goes through the scan kernels, 32 queries
    // per block of grid.y -- a (query, pool row) score is its own fixed chain of products and sums, independent of which block or
    // call the query sits in, so scores and top-k do not depend on how queries are batched into calls (Q = 1, 32, 33, 65, 256:
    // tests/test_gpu_ops.py).  Beyond one block the pool is re-read from the Infinity Cache / L2, once per block; the tiled GEMM
    // below (another summation order: 2 of 256 top-10 lists differed inside the 2e-6 band) serves the other d only --
    // R4D_SCORE_GEMM=1 restores it for Q > 64 (tuning aid).
    static int gemm_route = -1;
    if (gemm_route < 0) { const char* e = getenv("R4D_SCORE_GEMM"); gemm_route = e ? atoi(e) : 0; }
    if (Q <= 64 || !gemm_route) {
        rc = launch_pool_scan(q_hat_d, pool_hat_d, Q, N, d, scores, (unsigned*)ws, Q, s);
        zeroed = rc == R4D_OK;
    }
    if (rc > 0 && g_gemm_split3 && gemm_s3_f32b_supported(Q, d, N) && ((reinterpret_cast<uintptr_t>(q_hat_d) | reinterpret_cast<uintptr_t>(pool_hat_d)) & 15u) == 0) {
        // MFMA-bound regime on the bf16 matrix cores: queries AND pool rows split on the fly (no plane copy of the pool)
        R4D_BRANCH(SCAN_GEMM);
        rc = launch_gemm_s3_f32b(q_hat_d, pool_hat_d, scores, Q, N, d, d, N, EPI_HALF_PLUS, s);
    }
    if (rc > 0) {                                                                // MFMA-bound regime / other d: tiled GEMM
        R4D_BRANCH(SCAN_GEMM);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = q_hat_d; g.B = pool_hat_d; g.C = scores;
        g.M = Q; g.N = N; g.K = d; g.lda = d; g.ldb = d; g.ldc = N;
        g.b_trans = 1; g.b_rows = N; g.nbatch = 1; g.nb1 = 1; g.epilogue = EPI_HALF_PLUS; g.scale_div = 1.f;
        rc = launch_gemm_f32(g, s);
    }
    if (rc) return rc;
    return topk_rows<float>(scores, nullptr, Q, N, N, k, index_offset, out_val_d, (long long*)out_idx_d, ws,
                            workspace_bytes - skip, zeroed, s);
}

}  // extern "C"
