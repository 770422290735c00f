// Retrieval scoring on gfx950: row normalisation and the query-vs-pool cosine scan (exact-f32 MFMA), handing the
// score rows to the one-launch top-k of topk.hip.
// Reference: train/train_retriever.py:433-438 (normalise, matmul, (x+1)/2) and :357-358,461-467 (argsort).
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "common.h"

#ifndef SCAN_DBG
#define SCAN_DBG 0   // tuning aid (tools/kc_ablate.sh score.hip SCAN_DBG n): bit 0 skips the MFMAs, bit 1 the score stores, bit 2 the cross-wave reduction, bit 3 writes s_memrealtime stamps (100 MHz) of workgroup phases over the score rows of queries >= 16 (tools/scan_timeline.py)
#endif

namespace r4d {

int dbgflag_scan() { return SCAN_DBG != 0; }

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
typedef float f32x16s __attribute__((ext_vector_type(16)));

template <int KW, int NG, bool TWO>                     // d == 32 * KW * NG; TWO: two tiles of a workgroup in flight
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
    if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); stamps[5] = __builtin_amdgcn_s_memtime(); }
#endif
    const int ntiles = rpw > 0 ? (r1 - r0 + 31) / 32 : ((N + 31) / 32 - (int)blockIdx.x + G - 1) / G;
    // k order: load u of lane half h of row j is the 16-byte piece 2u + h of the wave's 128-byte line g, so ONE load
    // instruction touches a whole 32-byte sector of each of its 32 rows; component c of that load is
    // k = 32 (g KW + w) + 8 u + 4 h + c on BOTH operands.
    float4 qf[NG][4];
    {
        const bool ok = q0 + li < Q;
        const float4* src = reinterpret_cast<const float4*>(qhat + (long long)min(q0 + li, Q - 1) * D) + lh;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 v = src[8 * (g * KW + w) + 2 * u];
                if (!ok) { v.x = 0.f; v.y = 0.f; v.z = 0.f; v.w = 0.f; }
                qf[g][u] = v;
            }
    }
    auto tile_ptr = [&](int tt) {                       // rows past the shard are clamped (always valid); rows past r1 belong to the
        return reinterpret_cast<const float4*>(pool + (long long)min(r0 + tt * tstep + li, N - 1) * D) + lh + 8 * w;   // next workgroup: computed, not stored
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
            if (!(SCAN_DBG & 1)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].x, bb[g][u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].y, bb[g][u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].z, bb[g][u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].w, bb[g][u].w, acc, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += bb[g][u].x + bb[g][u].y + bb[g][u].z + bb[g][u].w;
            }
            if (trefill >= 0) {                                          // wave-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u) bb[g][u] = pn[8 * KW * g + 2 * u];
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
    auto load_tile = [&](float4 (&bb)[NG][4], int tt) {
        const float4* __restrict__ p = tile_ptr(tt);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) bb[g][u] = p[8 * KW * g + 2 * u];
    };
    float4 b0[NG][4];
    load_tile(b0, 0);
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
        float4 b1[NG][4];
        load_tile(b1, min(1, ntiles - 1));
        for (int t = 0; t < ntiles; t += 2) {
            const bool has2 = t + 1 < ntiles;                             // workgroup-uniform
            f32x16s acc2[2];
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
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc2[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].x, bb[g][u].x, acc2[h], 0, 0, 0);
                            acc2[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].y, bb[g][u].y, acc2[h], 0, 0, 0);
                            acc2[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].z, bb[g][u].z, acc2[h], 0, 0, 0);
                            acc2[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[g][u].w, bb[g][u].w, acc2[h], 0, 0, 0);
                        }
                        if (tn < ntiles) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) bb[g][u] = pn[8 * KW * g + 2 * u];
                        }
                    }
                }
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
                        if (q < Q) scores[(long long)q * N + row] = (fin[i] + 1.0f) / 2.0f;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { stamps[3] = __builtin_amdgcn_s_memrealtime(); stamps[6] = __builtin_amdgcn_s_memtime(); stamps[4] = (unsigned long long)__builtin_amdgcn_s_getreg(6164); }   // end; XCC id
#endif
}

template <int KW, int NG>
static int launch_scan_variant(const float* qhat, const float* pool, int Q, int N, float* scores, unsigned* zero_d, int nzero,
                               hipStream_t s) {
    static int wgs_per_cu = 0;
    if (wgs_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pool_scan_ks_kernel<KW, NG, false>, 64 * KW, 0) != hipSuccess || nb < 1) nb = 1;
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
    if (shortr) {
        R4D_BRANCH(SCAN_SHORT);
        const int rpw = cdiv(N, 256);
        hipLaunchKernelGGL((pool_scan_ks_kernel<KW, NG, (KW * NG >= 8)>), dim3(cdiv(N, rpw), cdiv(Q, 32)), dim3(64 * KW), 0, s, qhat,
                           pool, Q, N, rpw, scores, zero_d, nzero);
    } else {
        const int gx = max(1, min(ntiles, 256 * per_cu));
        hipLaunchKernelGGL((pool_scan_ks_kernel<KW, NG, false>), dim3(gx, cdiv(Q, 32)), dim3(64 * KW), 0, s, qhat, pool, Q, N, 0,
                           scores, zero_d, nzero);
    }
    R4D_CHECK_LAUNCH("pool_scan");
    return R4D_OK;
}

// d -> (KW, NG) with d == 32 * KW * NG; +1 = no instantiation (the tiled GEMM takes the call)
static int launch_pool_scan(const float* qhat, const float* pool, int Q, int N, int d, float* scores, unsigned* zero_d,
                            int nzero, hipStream_t s) {
    static int kw8 = -1;
    if (kw8 < 0) { const char* e = getenv("R4D_SCAN_KW8"); kw8 = e ? atoi(e) : 1; }               // tuning aid: 0 = 4-way split at d 512
    if (d != 32 && d != 64 && d != 128 && d != 256 && d != 384 && d != 512 && d != 768 && d != 1024) return 1;   // no variant
    // algorithmic bytes (SURVEY 8d B_score): pool read once per 32 queries + queries + score rows out
    ProfScope prof(PK_POOL_SCAN, 4.0 * N * d * cdiv(Q, 32) + 4.0 * Q * d + 4.0 * (double)Q * N, s);
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
    const size_t skip = align_up((size_t)Q * N * sizeof(float), 256);
    char* ws = (char*)workspace_d + skip;                                        // starts with the top-k ticket counters
    int rc = 1;
    bool zeroed = false;
    if (Q <= 64) {                                                               // HBM-bound regime: stream the pool once
        rc = launch_pool_scan(q_hat_d, pool_hat_d, Q, N, d, scores, (unsigned*)ws, Q, s);
        zeroed = rc == R4D_OK;
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
