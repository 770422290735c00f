// Jaccard pool annotation on gfx950.  Replaces the pure-python double loop occurrence_matrix /
// co_occurrence_ratio (retrieval_data_annotation.py:36-41, :5-15): out[i,j] = |A_i & B_j| / |A_i | B_j|
// as an IEEE double (python float), 0.0 when either set is empty, optional zeroed diagonal (:172-173).
//
// The pass is HBM-WRITE bound (8 bytes per pair out, a few bytes per SET in), so it is integer/bit work
// laid out for coalesced 512-byte row segments -- no GEMM reshaping.  One workgroup owns 64 B-sets (one
// per lane) and a chunk of A rows.  It first builds, in LDS, the transposed incidence table
//     mask[token] = 64-bit lane mask of the B-sets in this column tile that contain `token`
// (ds_or_b64 atomics; 8*vocab bytes, <= 152 KB of the 160 KB LDS).  Then each wavefront takes one A row at
// a time: its tokens are wave-uniform, every token is ONE broadcast LDS read of mask[token], and lane j
// adds bit j -- the popcount of the intersection accumulates across the wave's 64 pairs at 2 VALU ops per
// token.  |A|B| = |A| + |B| - |A&B|; the f64 division is correctly rounded == python's int/int.
// DENSE TOKENS.  The input sets of the annotation pass (get_input_seq, retrieval_data_annotation.py:17-20) keep their
// <|timeK|> tokens, so a dozen tokens occur in most sets and dominate the token walk (14 tokens per set on hepth, 5 of them
// sparse).  The host may therefore move up to 32 tokens of its choice out of the CSR lists into one 32-bit membership word
// per set (a_dense / b_dense): their share of |A & B| is popcount(a_dense & b_dense) -- two vector instructions per row
// and 64 pairs instead of 4.3 per token -- and the walk only visits the sparse remainder.  Same integers, same quotient.
// (Measured and dropped: tables in GLOBAL memory so that tokens and masks come through scalar loads and a token costs one
// v_addc_co_u32 with the mask as carry-in -- correct, but 2x SLOWER: random 8-byte lookups in a 95 KB table miss the scalar
// cache and the dependent s_load chains (token -> mask) are latency-bound.)
#include "common.h"

#ifndef JAC_DBG
#define JAC_DBG 0   // tuning aid (tools/kc_ablate.sh jaccard.hip JAC_DBG n): bit 0 skips the token loop, bit 1 the division, bit 2 the store
#endif

namespace r4d {

constexpr int JAC_MAX_VOCAB_LDS = 19455;           // 8 B * (19455 + 1) = 152 KB
constexpr int JAC_RCP = 512;                       // reciprocal table entries (4 KB of LDS after the mask table)

// Correctly rounded a / b for small integers without the 13-instruction IEEE f64 division sequence (which, not the
// HBM writes, bounded this kernel: ablation 1149 us -> 555 us on 20k x 20k when the division is removed):
//   r = RN(1/b) from an LDS table (filled once per workgroup with the true division), q0 = RN(a*r) is within 1 ulp,
//   rem = a - b*q0 exactly (one FMA), q1 = RN(q0 + rem*r) is the correctly rounded quotient (Markstein's theorem:
//   r correctly rounded, b's significand not all ones).  tests/test_host_cpu.py proves q1 == a/b for every
//   1 <= a <= b < 512 with exact rational arithmetic, tests/test_gpu_ops.py checks the kernel on sets that produce
//   every such (a, b).  Unions >= 512 take the IEEE division (wave-uniform branch).
// Tried and dropped (MI355X, round 2): a 64 x 64 LDS table of the quotients a / b themselves (one address + one ds_read_b64
// per pair instead of the reciprocal sequence): hepth input sets 71 -> 82 us, output sets 39 -> 48 us (the 4096 true
// divisions per workgroup and the 32 KB of LDS cost more than the lookup saves), synthetic 20k x 20k unchanged.  NOTE on
// ablations of this kernel: JAC_DBG bit 1 ("no division") also lets the compiler delete the token walk, whose only use is
// the quotient -- bit 0 (walk skipped, division kept) is the meaningful one: hepth input sets 71 -> 38 us, store-only 26 us.
__device__ __forceinline__ double small_int_div(int a, int b, const double* __restrict__ rcp) {
    const double af = (double)a, bf = (double)b, r = rcp[b];
    const double q0 = af * r;
    const double rem = __builtin_fma(-bf, q0, af);
    return __builtin_fma(rem, r, q0);
}

__global__ __launch_bounds__(1024) void jaccard_lds_kernel(const int32_t* __restrict__ a_ptr,
                                                          const int32_t* __restrict__ a_idx, int na, int a_nnz,
                                                          const int32_t* __restrict__ b_ptr,
                                                          const int32_t* __restrict__ b_idx, int nb, int vocab,
                                                          int zero_diag, int rows_per_block,
                                                          const int32_t* __restrict__ a_order,
                                                          const uint32_t* __restrict__ a_dense,
                                                          const uint32_t* __restrict__ b_dense,
                                                          const int32_t* __restrict__ a_len,
                                                          const int32_t* __restrict__ b_len, double* __restrict__ out) {
    extern __shared__ unsigned long long mask[];    // [vocab + 1]
    // a_len / b_len (optional, the device-side preparation below): a set's list is idx[ptr[i] .. ptr[i] + len[i]) -- the dense
    // tokens were squeezed out of the list IN PLACE, so the lists keep their offsets and no prefix sum is needed
    auto a_end = [&](int row, int s) { return a_len ? s + a_len[row] : a_ptr[row + 1]; };
    // the wavefront index is made PROVABLY wave-uniform: everything per A row (CSR pointers, loop control, the row's
    // output base address) then lives in scalar registers / scalar loads instead of vector instructions
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col0 = blockIdx.x * 64;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    double* rcp = reinterpret_cast<double*>(mask + vocab + 1);           // [JAC_RCP]
    for (int t = tid; t <= vocab; t += nthreads) mask[t] = 0ull;          // slot [vocab] stays zero
    for (int t = tid; t < JAC_RCP; t += nthreads) rcp[t] = t ? 1.0 / (double)t : 0.0;
    __syncthreads();
    if (tid < 256) {   // scatter the 64 B-sets of this tile: 4 threads per set
        const int j = tid >> 2, sub = tid & 3, cj = col0 + j;
        if (cj < nb) {
            const int s = b_ptr[cj], e = b_len ? s + b_len[cj] : b_ptr[cj + 1];
            for (int p = s + sub; p < e; p += 4) {
                const int tok = b_idx[p];
                if ((unsigned)tok < (unsigned)vocab) atomicOr(&mask[tok], 1ull << j);
            }
        }
    }
    __syncthreads();
    const int col = col0 + lane;
    const uint32_t bd = (b_dense && col < nb) ? b_dense[col] : 0u;         // this column's dense-token word
    const int lb = (col < nb) ? (b_len ? b_len[col] : b_ptr[col + 1] - b_ptr[col]) + __builtin_popcount(bd) : 0;
    const int row_begin = blockIdx.y * rows_per_block;
    const int row_end = min(na, row_begin + rows_per_block);
    // R = 4 independent A rows per wavefront iteration (rows i, i+nwaves, ..., i+3*nwaves).  With the big table only
    // one workgroup (4 waves per SIMD) fits a CU and one row is a serial chain of dependent latencies (token ->
    // LDS word -> count -> reciprocal from LDS -> 3 f64 ops -> store), ~1000 cycles per row and wave: four chains
    // side by side fill them.  Software pipeline: the CSR pointers of the group after next (scalar loads) and the
    // first 64 tokens of the next group's rows are in flight while the current group is counted and stored.  All
    // loads use clamped, always-valid indices; validity is applied to the VALUE (tokens past a row read as -1,
    // which selects the always-zero slot mask[vocab]).
    constexpr int R = 4;
    const int last = na - 1, nz1 = a_nnz - 1;
    const unsigned int* mask32 = reinterpret_cast<const unsigned int*>(mask);
    const int half = lane >> 5, bit = lane & 31;
    const int stride = R * nwaves;
    // a_order (optional): the A rows are VISITED in this order (longest set first) while results still land in their own
    // rows.  The four rows of an iteration walk their tokens jointly, to the longest of the four: in file order that is
    // 1.7x the token steps the sets hold (input sets: log-normal lengths), sorted it is 1.04x.
    int i = row_begin + wid;                        // POSITION in the visiting order
    int sA[R], eA[R], sB[R], eB[R], tokA[R], rowA[R], rowB[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pa = min(i + r * nwaves, last), pb = min(i + stride + r * nwaves, last);
        rowA[r] = a_order ? a_order[pa] : pa;
        rowB[r] = a_order ? a_order[pb] : pb;
        sA[r] = a_ptr[rowA[r]]; eA[r] = a_end(rowA[r], sA[r]);
        sB[r] = a_ptr[rowB[r]]; eB[r] = a_end(rowB[r], sB[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = a_idx[min(sA[r] + lane, nz1)];
        tokA[r] = (sA[r] + lane < eA[r]) ? t : -1;
    }
    for (; i < row_end; i += stride) {
        int sC[R], eC[R], tokB[R], rowC[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int pc = min(i + 2 * stride + r * nwaves, last);
            rowC[r] = a_order ? a_order[pc] : pc;
            sC[r] = a_ptr[rowC[r]]; eC[r] = a_end(rowC[r], sC[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = a_idx[min(sB[r] + lane, nz1)];
            tokB[r] = (sB[r] + lane < eB[r]) ? t : -1;
        }
        int la[R], lw[R], cnt[R];                       // |A| (dense tokens included), tokens to walk, |A & B|
        int mmax = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t ad = a_dense ? a_dense[rowA[r]] : 0u;         // wave-uniform (scalar load)
            lw[r] = eA[r] - sA[r];
            la[r] = lw[r] + __builtin_popcount(ad);
            cnt[r] = __builtin_popcount(ad & bd);
            mmax = max(mmax, min(lw[r], 64));
        }
        if (JAC_DBG & 1) mmax = 0;
        // joint walk over the first 64 tokens of the four rows: four independent broadcast LDS reads per step
        // TU token positions per trip: 4 x TU independent LDS reads in flight before the first is consumed (lanes past
        // a row's end hold -1, which selects the zero slot, so the odd tail needs no special case)
        constexpr int TU = 2;           // (4: no gain on long sets, -5 % on the one- and two-token output sets)
        for (int t = 0; t < mmax; t += TU) {
            unsigned int bits[TU][R];
#pragma unroll
            for (int h2 = 0; h2 < TU; ++h2)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int tok = __builtin_amdgcn_readlane(tokA[r], (t + h2) & 63);   // scalar broadcast
                    const int tkk = ((unsigned)tok < (unsigned)vocab) ? tok : vocab;
                    bits[h2][r] = mask32[2 * tkk + half];
                }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int h2 = 0; h2 < TU; ++h2) cnt[r] += (int)((bits[h2][r] >> bit) & 1u);
        }
        // rows longer than 64 tokens (rare): the remaining chunks, one row at a time
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((JAC_DBG & 1) || lw[r] <= 64) continue;
            for (int p0 = sA[r] + 64; p0 < eA[r]; p0 += 64) {
                int mytok = a_idx[min(p0 + lane, nz1)];
                mytok = (p0 + lane < eA[r]) ? mytok : -1;
                const int m = min(64, eA[r] - p0);
                for (int t = 0; t < m; t += 4) {
                    unsigned int bits[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int tok = __builtin_amdgcn_readlane(mytok, (t + u) & 63);
                        const int tkk = ((unsigned)tok < (unsigned)vocab) ? tok : vocab;   // lanes past m hold -1
                        bits[u] = mask32[2 * tkk + half];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) cnt[r] += (int)((bits[u] >> bit) & 1u);
                }
            }
        }
        double q[R];
        int any_cnt = 0, any_big = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            any_cnt |= cnt[r];
            any_big |= (la[r] + lb - cnt[r] >= JAC_RCP);
            q[r] = 0.0;
        }
        if (!(JAC_DBG & 2) && __any(any_cnt != 0)) {     // most 4 x 64-pair groups have only empty intersections: skip
            if (__any(any_big)) {                        // some union >= table size: IEEE division for the whole group
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (la[r] > 0 && lb > 0) q[r] = (double)cnt[r] / (double)(la[r] + lb - cnt[r]);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)         // rcp[0] = 0: an empty set gives 0.0
                    q[r] = small_int_div(cnt[r], (la[r] > 0 && lb > 0) ? la[r] + lb - cnt[r] : 0, rcp);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = rowA[r];                         // wave-uniform
            if (i + r * nwaves < row_end) {
                double v = q[r];
                if (zero_diag && row == col) v = 0.0;
                if ((JAC_DBG & 4) ? (v == 12345.0) : (col < nb)) (out + (long long)row * nb)[(unsigned)col] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            sA[r] = sB[r]; eA[r] = eB[r]; tokA[r] = tokB[r]; rowA[r] = rowB[r];
            sB[r] = sC[r]; eB[r] = eC[r]; rowB[r] = rowC[r];
        }
    }
}

// Fallback for vocabularies whose incidence table does not fit LDS: one thread per pair, sorted-list merge.
__global__ __launch_bounds__(256) void jaccard_merge_kernel(const int32_t* __restrict__ a_ptr,
                                                            const int32_t* __restrict__ a_idx, int na,
                                                            const int32_t* __restrict__ b_ptr,
                                                            const int32_t* __restrict__ b_idx, int nb, int zero_diag,
                                                            const uint32_t* __restrict__ a_dense,
                                                            const uint32_t* __restrict__ b_dense,
                                                            const int32_t* __restrict__ a_len,
                                                            const int32_t* __restrict__ b_len, double* __restrict__ out) {
    const int col_blocks = (nb + 255) / 256;
    const int col = (blockIdx.x % col_blocks) * 256 + threadIdx.x, i = blockIdx.x / col_blocks;
    if (col >= nb) return;
    int p = a_ptr[i], pe = a_len ? p + a_len[i] : a_ptr[i + 1], q = b_ptr[col], qe = b_len ? q + b_len[col] : b_ptr[col + 1];
    const uint32_t ad = a_dense ? a_dense[i] : 0u, bd = b_dense ? b_dense[col] : 0u;
    const int la = pe - p + __builtin_popcount(ad), lb = qe - q + __builtin_popcount(bd);
    int cnt = __builtin_popcount(ad & bd);
    while (p < pe && q < qe) {
        const int x = a_idx[p], y = b_idx[q];
        cnt += (x == y);
        p += (x <= y);
        q += (y <= x);
    }
    double r = 0.0;
    if (la > 0 && lb > 0) r = (double)cnt / (double)(la + lb - cnt);
    if (zero_diag && i == col) r = 0.0;
    out[(long long)i * nb + col] = r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Set preparation ON THE DEVICE (round 5; until then a dozen torch index ops per call -- bincount, topk, repeat_interleave,
// index_add_, boolean-mask compaction, cumsum, argsort -- that together took more GPU time than the Jaccard kernel itself on
// the hepth matrices).  Three small launches in front of the kernel, none of them changes a value:
//   jac_pick_dense   one workgroup: token frequencies over a strided SAMPLE of <= 2 x 512 sets (an LDS histogram), then the 32
//                    most frequent tokens (count descending, token ascending) by a threshold on the histogram of the counts.
//                    Which tokens are picked only decides how much of the token walk turns into a popcount; a sample finds the
//                    <|timeK|> tokens that sit in most sets as surely as a full count does.
//   jac_split        16 lanes per set: dense tokens -> bits of the set's membership word, the others squeezed to the front of
//                    the set's own segment in a copy of idx (offsets unchanged -> no prefix sum; the kernel takes a_len / b_len).
//   jac_order        one workgroup: the A rows by list length, longest first (counting sort over the lengths; ties in no
//                    particular order -- the order is a schedule, results land in their own rows).
constexpr int JAC_SAMPLE_SETS = 512;               // per family: one sampled set per thread of the picking workgroup
constexpr int JAC_PICK_LDS_VOCAB = 36864;          // int32 histogram in LDS (144 KB); larger vocabularies count in global memory
constexpr int JAC_LEN_BINS = 1024;                 // lists of >= 1023 tokens share the first (longest) bin

// The 32 most frequent tokens of the sample, count descending, token id ascending on a tie -- without 32 argmax rounds:
//   1. hist[token] = number of sampled sets holding it (<= 1024);  2. chist[c] = number of tokens with count c;
//   3. a suffix sum over c gives the threshold T = the largest c with  #{count >= c} >= 32;  4. the (< 32) tokens above T are
//   collected and ranked among themselves;  5. the remaining slots go to the tokens with count == T in token order (one ballot
//   prefix per 1,024 tokens, stopping when the slots are full).  Deterministic; ~15 us where the rounds took 80.
template <bool LDS>
__global__ __launch_bounds__(1024) void jac_pick_dense_kernel(const int32_t* __restrict__ a_ptr, const int32_t* __restrict__ a_idx,
                                                             int na, const int32_t* __restrict__ b_ptr,
                                                             const int32_t* __restrict__ b_idx, int nb, int vocab,
                                                             int* __restrict__ cnt_g, signed char* __restrict__ rank) {
    extern __shared__ int hist_l[];                 // [vocab] when LDS
    __shared__ int chist[1025];
    __shared__ int wtot[2][16];
    __shared__ unsigned long long cand[32];
    __shared__ int n_cand, thr, n_above_s;
    int* hist;
    if constexpr (LDS) hist = hist_l; else hist = cnt_g;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    auto count_of = [&](int t) {                    // global counts were made by L2 atomics: read them there, not in this CU's L1
        int c;
        if constexpr (LDS) c = hist[t]; else c = __hip_atomic_load(&hist[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return min(c, 1024);
    };
    for (int t = tid; t < vocab; t += 1024) { hist[t] = 0; rank[t] = (signed char)-1; }
    chist[tid] = 0;
    if (tid == 0) { chist[1024] = 0; n_cand = 0; thr = 0; n_above_s = 0; }
    __syncthreads();
    const int stepA = max(1, (na + JAC_SAMPLE_SETS - 1) / JAC_SAMPLE_SETS), stepB = max(1, (nb + JAC_SAMPLE_SETS - 1) / JAC_SAMPLE_SETS);
    const int nsA = (na + stepA - 1) / stepA, nsB = (nb + stepB - 1) / stepB;      // <= 512 each
    if (tid < nsA + nsB) {
        const bool fb = tid >= nsA;
        const int row = fb ? (tid - nsA) * stepB : tid * stepA;
        const int32_t* ptr = fb ? b_ptr : a_ptr;
        const int32_t* idx = fb ? b_idx : a_idx;
        for (int p = ptr[row], e = ptr[row + 1]; p < e; p += 4) {          // four independent loads in flight per trip
            int tok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) tok[u] = idx[min(p + u, e - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u < e && (unsigned)tok[u] < (unsigned)vocab) atomicAdd(&hist[tok[u]], 1);
        }
    }
    __syncthreads();
    // chist: counts 1..3 (most tokens) are tallied per wavefront by ballots, the rest by LDS atomics
    int c1 = 0, c2 = 0, c3 = 0;
    for (int base = 0; base < vocab; base += 1024) {
        const int t = base + tid;
        const int c = t < vocab ? count_of(t) : 0;
        c1 += __popcll(__ballot(c == 1)); c2 += __popcll(__ballot(c == 2)); c3 += __popcll(__ballot(c == 3));
        if (c >= 4) atomicAdd(&chist[c], 1);
    }
    if (lane == 0) { atomicAdd(&chist[1], c1); atomicAdd(&chist[2], c2); atomicAdd(&chist[3], c3); }
    __syncthreads();
    // thread tid stands for count c = 1024 - tid: an inclusive scan over tid is S(c) = #{tokens with count >= c}
    const int c = 1024 - tid, v = chist[c];
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[0][wid] = incl;
    __syncthreads();
    int S = incl;
    for (int k = 0; k < wid; ++k) S += wtot[0][k];
    if (S >= 32 && S - v < 32) { thr = c; n_above_s = S - v; }       // exactly one c has this property when >= 32 tokens were seen
    if (c == 1 && S < 32) { thr = 0; n_above_s = S; }               // fewer: all of them are "above"
    __syncthreads();
    const int T = thr, n_above = n_above_s;
    for (int base = 0; base < vocab; base += 1024) {
        const int t = base + tid;
        if (t < vocab) {
            const int ct = count_of(t);
            if (ct > T) cand[atomicAdd(&n_cand, 1)] = ((unsigned long long)(unsigned)ct << 32) | (unsigned)(~t);    // < 32 of them
        }
    }
    __syncthreads();
    if (tid < n_cand) {
        const unsigned long long my = cand[tid];
        int r = 0;
        for (int j = 0; j < n_cand; ++j) r += cand[j] > my;
        rank[(int)~(unsigned)(my & 0xFFFFFFFFull)] = (signed char)r;
    }
    if (T >= 1) {
        const int m = 32 - n_above;
        int found = 0, it = 0;
        for (int base = 0; base < vocab && found < m; base += 1024, ++it) {
            const int t = base + tid;
            const bool flag = t < vocab && count_of(t) == T;
            const unsigned long long bal = __ballot(flag);
            if (lane == 0) wtot[it & 1][wid] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int w = wtot[it & 1][k]; total += w; before += k < wid ? w : 0; }
            const int pos = found + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (flag && pos < m) rank[t] = (signed char)(n_above + pos);
            found += total;
        }
    }
}

// 16 lanes per set, four sets per wavefront: a set's tokens are looked up side by side, the kept ones keep their order (ballot prefix)
__global__ __launch_bounds__(256) void jac_split_kernel(const int32_t* __restrict__ a_ptr, const int32_t* __restrict__ a_idx, int na,
                                                       const int32_t* __restrict__ b_ptr, const int32_t* __restrict__ b_idx, int nb,
                                                       int vocab, const signed char* __restrict__ rank,
                                                       int32_t* __restrict__ a_idx2, int32_t* __restrict__ a_len,
                                                       uint32_t* __restrict__ a_dense, int32_t* __restrict__ b_idx2,
                                                       int32_t* __restrict__ b_len, uint32_t* __restrict__ b_dense) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int g = (int)((blockIdx.x * 256u + threadIdx.x) >> 4);
    const bool valid = g < na + nb, fb = valid && g >= na;
    const int row = fb ? g - na : g;
    const int32_t* ptr = fb ? b_ptr : a_ptr;
    const int32_t* idx = fb ? b_idx : a_idx;
    int32_t* idx2 = fb ? b_idx2 : a_idx2;
    const int s = valid ? ptr[row] : 0, e = valid ? ptr[row + 1] : 0;
    int k = 0;
    uint32_t word = 0u;
    for (int p0 = s; __any(p0 < e); p0 += 16) {
        const int p = p0 + sub;
        const bool in = p < e;
        const int tok = in ? idx[p] : -1;
        const int r = (in && (unsigned)tok < (unsigned)vocab) ? (int)rank[tok] : -1;
        const bool keep = in && r < 0;
        if (in && r >= 0) word |= 1u << r;
        const unsigned gm = (unsigned)(__ballot(keep) >> (grp * 16)) & 0xFFFFu;
        if (keep) idx2[s + k + __popc(gm & ((1u << sub) - 1u))] = tok;
        k += __popc(gm);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) word |= (uint32_t)__shfl_xor((int)word, o, 64);
    if (valid && sub == 0) {
        (fb ? b_len : a_len)[row] = k;
        (fb ? b_dense : a_dense)[row] = word;
    }
}

__global__ __launch_bounds__(1024) void jac_order_kernel(const int32_t* __restrict__ a_ptr, const int32_t* __restrict__ a_len, int na,
                                                        int32_t* __restrict__ order) {
    __shared__ int bins[JAC_LEN_BINS];              // bin 0 = the longest lists
    const int tid = threadIdx.x;
    auto bin_of = [&](int row) {
        const int len = a_len ? a_len[row] : a_ptr[row + 1] - a_ptr[row];
        return JAC_LEN_BINS - 1 - min(max(len, 0), JAC_LEN_BINS - 1);
    };
    for (int t = tid; t < JAC_LEN_BINS; t += 1024) bins[t] = 0;
    __syncthreads();
    for (int row = tid; row < na; row += 4096) {                         // four independent length loads per trip
        int bn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bn[u] = bin_of(min(row + 1024 * u, na - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (row + 1024 * u < na) atomicAdd(&bins[bn[u]], 1);
    }
    __syncthreads();
    // exclusive prefix over the 1024 bins: one bin per thread, wave scans + the 16 wave totals
    __shared__ int wsum[16];
    const int lane = tid & 63, wid = tid >> 6;
    const int c = bins[tid];
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wid; ++k) base += wsum[k];
    bins[tid] = base + incl - c;
    __syncthreads();
    for (int row = tid; row < na; row += 4096) {
        int bn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bn[u] = bin_of(min(row + 1024 * u, na - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (row + 1024 * u < na) order[atomicAdd(&bins[bn[u]], 1)] = row + 1024 * u;
    }
}

static inline size_t jac_al(size_t b) { return (b + 255) & ~(size_t)255; }

static int launch_jaccard(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz, const int32_t* b_ptr_d,
                          const int32_t* b_idx_d, int32_t nb, int32_t b_nnz, int32_t vocab, int32_t zero_diag,
                          const int32_t* a_order_d, const uint32_t* a_dense_d, const uint32_t* b_dense_d, const int32_t* a_len_d,
                          const int32_t* b_len_d, double* out_d, hipStream_t s) {
    // algorithmic bytes (SURVEY 8d B_jac): the f64 matrix out + both CSR inputs read once
    ProfScope prof(PK_JACCARD, 8.0 * na * (double)nb + 4.0 * (na + nb + 2) + 4.0 * ((double)a_nnz + b_nnz), s);
    if (vocab <= JAC_MAX_VOCAB_LDS) {
        const int col_tiles = cdiv(nb, 64);
        // enough row chunks for >= ~8 workgroups per CU, but >= 64 rows each so the table build amortises
        int chunks = max(1, min(cdiv(na, 64), cdiv(2048, col_tiles)));
        const size_t lds = ((size_t)vocab + 1) * sizeof(unsigned long long) + JAC_RCP * sizeof(double);
        if (lds > 64 * 1024) {
            static bool raised = false;     // opt in to > 64 KB dynamic LDS once
            if (!raised) {
                if (hipFuncSetAttribute((const void*)jaccard_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (JAC_MAX_VOCAB_LDS + 1) * 8 + JAC_RCP * 8) != hipSuccess) {
                    set_error("jaccard: cannot raise dynamic LDS limit");
                    return R4D_ERR_HIP;
                }
                raised = true;
            }
        }
        // a big table leaves room for one workgroup per CU only: give it 16 wavefronts instead of 4
        const int threads = lds > 80 * 1024 ? 1024 : (lds > 40 * 1024 ? 512 : 256);
        const int rows_per_block = cdiv(na, chunks);
        chunks = cdiv(na, rows_per_block);
        R4D_BRANCH(JAC_LDS);
        hipLaunchKernelGGL(jaccard_lds_kernel, dim3(col_tiles, chunks), dim3(threads), lds, s, a_ptr_d, a_idx_d, na, a_nnz,
                           b_ptr_d, b_idx_d, nb, vocab, zero_diag, rows_per_block, a_order_d, a_dense_d, b_dense_d, a_len_d,
                           b_len_d, out_d);
        R4D_CHECK_LAUNCH("jaccard_lds");
    } else {
        R4D_BRANCH(JAC_MERGE);
        hipLaunchKernelGGL(jaccard_merge_kernel, dim3((unsigned)((long long)cdiv(nb, 256) * na)), dim3(256), 0, s, a_ptr_d, a_idx_d, na, b_ptr_d,
                           b_idx_d, nb, zero_diag, a_dense_d, b_dense_d, a_len_d, b_len_d, out_d);
        R4D_CHECK_LAUNCH("jaccard_merge");
    }
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

static int jac_check_args(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t& a_nnz, const int32_t* b_ptr_d,
                          const int32_t* b_idx_d, int32_t nb, int32_t& b_nnz, int32_t vocab, const double* out_d) {
    R4D_REQUIRE(a_ptr_d && b_ptr_d && a_idx_d && b_idx_d && out_d, "jaccard: null pointer");
    R4D_REQUIRE(a_nnz >= 0 && b_nnz >= 0, "jaccard: negative nnz");
    if (a_nnz < 1) a_nnz = 1;                       // idx buffers hold >= 1 element by contract
    if (b_nnz < 1) b_nnz = 1;
    R4D_REQUIRE(na >= 0 && nb >= 0 && vocab >= 1, "jaccard: bad sizes na=%d nb=%d vocab=%d", na, nb, vocab);
    return R4D_OK;
}

extern "C" int r4d_jaccard_ordered_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                                       const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz,
                                       int32_t vocab, int32_t zero_diag, const int32_t* a_order_d,
                                       const uint32_t* a_dense_d, const uint32_t* b_dense_d, double* out_d, void* stream) {
    if (int rc = jac_check_args(a_ptr_d, a_idx_d, na, a_nnz, b_ptr_d, b_idx_d, nb, b_nnz, vocab, out_d)) return rc;
    R4D_REQUIRE((a_dense_d != nullptr) == (b_dense_d != nullptr), "jaccard: a_dense_d and b_dense_d go together");
    if (na == 0 || nb == 0) return R4D_OK;
    return launch_jaccard(a_ptr_d, a_idx_d, na, a_nnz, b_ptr_d, b_idx_d, nb, b_nnz, vocab, zero_diag, a_order_d, a_dense_d, b_dense_d,
                          nullptr, nullptr, out_d, (hipStream_t)stream);
}

extern "C" int r4d_jaccard_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                               const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz, int32_t vocab,
                               int32_t zero_diag, double* out_d, void* stream) {
    return r4d_jaccard_ordered_f64(a_ptr_d, a_idx_d, na, a_nnz, b_ptr_d, b_idx_d, nb, b_nnz, vocab, zero_diag, nullptr, nullptr,
                                   nullptr, out_d, stream);
}

// workspace: rank[vocab] | counts[vocab] | a_idx2 | b_idx2 | a_len | b_len | a_dense | b_dense | order
extern "C" size_t r4d_jaccard_prepared_workspace_bytes(int32_t na, int32_t a_nnz, int32_t nb, int32_t b_nnz, int32_t vocab) {
    if (na < 0 || nb < 0 || a_nnz < 0 || b_nnz < 0 || vocab < 1) return 0;
    const size_t na_ = (size_t)(na > 0 ? na : 1), nb_ = (size_t)(nb > 0 ? nb : 1);
    return jac_al((size_t)vocab) + jac_al(4 * (size_t)vocab) + jac_al(4 * (size_t)(a_nnz > 0 ? a_nnz : 1)) +
           jac_al(4 * (size_t)(b_nnz > 0 ? b_nnz : 1)) + 3 * jac_al(4 * na_) + 2 * jac_al(4 * nb_);
}

extern "C" int r4d_jaccard_prepared_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                                        const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz,
                                        int32_t vocab, int32_t zero_diag, int32_t dense_split, int32_t sort_rows,
                                        double* out_d, void* workspace_d, size_t workspace_bytes, void* stream) {
    if (int rc = jac_check_args(a_ptr_d, a_idx_d, na, a_nnz, b_ptr_d, b_idx_d, nb, b_nnz, vocab, out_d)) return rc;
    if (na == 0 || nb == 0) return R4D_OK;
    hipStream_t s = (hipStream_t)stream;
    const int32_t* order = nullptr;
    const int32_t *a_len = nullptr, *b_len = nullptr, *a_use = a_idx_d, *b_use = b_idx_d;
    const uint32_t *a_dense = nullptr, *b_dense = nullptr;
    if (dense_split || sort_rows) {
        R4D_REQUIRE(workspace_d && workspace_bytes >= r4d_jaccard_prepared_workspace_bytes(na, a_nnz, nb, b_nnz, vocab),
                    "jaccard_prepared: workspace too small");
        char* w = (char*)workspace_d;
        signed char* rank = (signed char*)w;           w += jac_al((size_t)vocab);
        int* counts = (int*)w;                         w += jac_al(4 * (size_t)vocab);
        int32_t* a_idx2 = (int32_t*)w;                 w += jac_al(4 * (size_t)a_nnz);
        int32_t* b_idx2 = (int32_t*)w;                 w += jac_al(4 * (size_t)b_nnz);
        int32_t* a_len_w = (int32_t*)w;                w += jac_al(4 * (size_t)na);
        int32_t* b_len_w = (int32_t*)w;                w += jac_al(4 * (size_t)nb);
        uint32_t* a_dense_w = (uint32_t*)w;            w += jac_al(4 * (size_t)na);
        uint32_t* b_dense_w = (uint32_t*)w;            w += jac_al(4 * (size_t)nb);
        int32_t* order_w = (int32_t*)w;
        ProfScope prof(PK_JACCARD_PREP, 4.0 * ((double)a_nnz + b_nnz) * (dense_split ? 2.0 : 0.0) + 8.0 * ((double)na + nb), s);
        if (dense_split) {
            const int use_lds = vocab <= JAC_PICK_LDS_VOCAB;
            const size_t lds = use_lds ? 4 * (size_t)vocab : 0;
            if (lds > 48 * 1024) {
                static bool raised = false;
                if (!raised) {
                    if (hipFuncSetAttribute((const void*)jac_pick_dense_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            4 * JAC_PICK_LDS_VOCAB) != hipSuccess) {
                        set_error("jaccard_prepared: cannot raise dynamic LDS limit");
                        return R4D_ERR_HIP;
                    }
                    raised = true;
                }
            }
            if (use_lds) R4D_BRANCH(JAC_PREP_DENSE_LDS); else R4D_BRANCH(JAC_PREP_DENSE_GLOBAL);
            if (use_lds)
                hipLaunchKernelGGL(jac_pick_dense_kernel<true>, dim3(1), dim3(1024), lds, s, a_ptr_d, a_idx_d, na, b_ptr_d, b_idx_d, nb,
                                   vocab, counts, rank);
            else
                hipLaunchKernelGGL(jac_pick_dense_kernel<false>, dim3(1), dim3(1024), 0, s, a_ptr_d, a_idx_d, na, b_ptr_d, b_idx_d, nb,
                                   vocab, counts, rank);
            R4D_CHECK_LAUNCH("jac_pick_dense");
            hipLaunchKernelGGL(jac_split_kernel, dim3(cdiv(na + nb, 16)), dim3(256), 0, s, a_ptr_d, a_idx_d, na, b_ptr_d, b_idx_d, nb,
                               vocab, rank, a_idx2, a_len_w, a_dense_w, b_idx2, b_len_w, b_dense_w);
            R4D_CHECK_LAUNCH("jac_split");
            a_use = a_idx2; b_use = b_idx2; a_len = a_len_w; b_len = b_len_w; a_dense = a_dense_w; b_dense = b_dense_w;
        }
        if (sort_rows && na > 1) {
            R4D_BRANCH(JAC_PREP_ORDER);
            hipLaunchKernelGGL(jac_order_kernel, dim3(1), dim3(1024), 0, s, a_ptr_d, a_len, na, order_w);
            R4D_CHECK_LAUNCH("jac_order");
            order = order_w;
        }
    }
    return launch_jaccard(a_ptr_d, a_use, na, a_nnz, b_ptr_d, b_use, nb, b_nnz, vocab, zero_diag, order, a_dense, b_dense, a_len, b_len,
                          out_d, s);
}

namespace r4d { int dbgflag_jac() { return JAC_DBG != 0; } }
