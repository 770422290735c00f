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
#include "common.h"

namespace r4d {

constexpr int JAC_MAX_VOCAB_LDS = 19455;           // 8 B * (19455 + 1) = 152 KB

__global__ __launch_bounds__(1024) void jaccard_lds_kernel(const int32_t* __restrict__ a_ptr,
                                                          const int32_t* __restrict__ a_idx, int na, int a_nnz,
                                                          const int32_t* __restrict__ b_ptr,
                                                          const int32_t* __restrict__ b_idx, int nb, int vocab,
                                                          int zero_diag, int rows_per_block, double* __restrict__ out) {
    extern __shared__ unsigned long long mask[];    // [vocab + 1]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int col0 = blockIdx.x * 64;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    for (int t = tid; t <= vocab; t += nthreads) mask[t] = 0ull;          // slot [vocab] stays zero
    __syncthreads();
    if (tid < 256) {   // scatter the 64 B-sets of this tile: 4 threads per set
        const int j = tid >> 2, sub = tid & 3, col = col0 + j;
        if (col < nb) {
            const int s = b_ptr[col], e = b_ptr[col + 1];
            for (int p = s + sub; p < e; p += 4) {
                const int tok = b_idx[p];
                if ((unsigned)tok < (unsigned)vocab) atomicOr(&mask[tok], 1ull << j);
            }
        }
    }
    __syncthreads();
    const int col = col0 + lane;
    const int lb = (col < nb) ? (b_ptr[col + 1] - b_ptr[col]) : 0;
    const int row_begin = blockIdx.y * rows_per_block;
    const int row_end = min(na, row_begin + rows_per_block);
    // Software pipeline over this wavefront's rows (i, i+nwaves, ...): the CSR pointers of the row after next
    // and the first 64 tokens of the next row are in flight while the current row is counted and stored, so the
    // two dependent global round trips per row (ptr -> tokens) are off the critical path.  All loads use clamped,
    // always-valid indices; validity is applied to the VALUE.
    const int last = na - 1, nz1 = a_nnz - 1;
    const unsigned int* mask32 = reinterpret_cast<const unsigned int*>(mask);
    const int half = lane >> 5, bit = lane & 31;
    int i = row_begin + wid;
    int sA = a_ptr[min(i, last)], eA = a_ptr[min(i, last) + 1];
    int sB = a_ptr[min(i + nwaves, last)], eB = a_ptr[min(i + nwaves, last) + 1];
    int tokA = a_idx[min(sA + lane, nz1)];
    tokA = (sA + lane < eA) ? tokA : -1;
    for (; i < row_end; i += nwaves) {
        const int i2 = min(i + 2 * nwaves, last);
        const int sC = a_ptr[i2], eC = a_ptr[i2 + 1];
        int tokB = a_idx[min(sB + lane, nz1)];
        tokB = (sB + lane < eB) ? tokB : -1;
        const int la = eA - sA;
        int cnt = 0;
        int mytok = tokA;
        for (int p0 = sA; p0 < eA; p0 += 64) {
            if (p0 != sA) {
                mytok = a_idx[min(p0 + lane, nz1)];
                mytok = (p0 + lane < eA) ? mytok : -1;
            }
            const int m = min(64, eA - p0);
            // 4 tokens per trip: four independent broadcast LDS reads in flight (out-of-range / padding tokens
            // are redirected to the always-zero slot mask[vocab], so there is no branch in the loop)
            int t = 0;
            if (m <= 2) {                             // typical output set: one or two tokens, no unroll overhead
                for (; t < m; ++t) {
                    const int tok = __builtin_amdgcn_readlane(mytok, t);
                    const int tkk = ((unsigned)tok < (unsigned)vocab) ? tok : vocab;
                    cnt += (int)((mask32[2 * tkk + half] >> bit) & 1u);
                }
            }
            for (; t < m; t += 4) {
                int tk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tok = __builtin_amdgcn_readlane(mytok, (t + u) & 63);     // scalar broadcast
                    tk[u] = ((unsigned)tok < (unsigned)vocab && t + u < m) ? tok : vocab;
                }
                unsigned int bits[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) bits[u] = mask32[2 * tk[u] + half];
#pragma unroll
                for (int u = 0; u < 4; ++u) cnt += (int)((bits[u] >> bit) & 1u);
            }
        }
        double r = 0.0;
        if (__any(cnt != 0)) {                       // most 64-pair segments have an empty intersection: skip the
            if (la > 0 && lb > 0) r = (double)cnt / (double)(la + lb - cnt);    // f64 division for the whole wave
            if (zero_diag && i == col) r = 0.0;
        }
        if (col < nb) out[(long long)i * nb + col] = r;
        sA = sB; eA = eB; tokA = tokB; sB = sC; eB = eC;
    }
}

// Fallback for vocabularies whose incidence table does not fit LDS: one thread per pair, sorted-list merge.
__global__ __launch_bounds__(256) void jaccard_merge_kernel(const int32_t* __restrict__ a_ptr,
                                                            const int32_t* __restrict__ a_idx, int na,
                                                            const int32_t* __restrict__ b_ptr,
                                                            const int32_t* __restrict__ b_idx, int nb, int zero_diag,
                                                            double* __restrict__ out) {
    const int col_blocks = (nb + 255) / 256;
    const int col = (blockIdx.x % col_blocks) * 256 + threadIdx.x, i = blockIdx.x / col_blocks;
    if (col >= nb) return;
    int p = a_ptr[i], pe = a_ptr[i + 1], q = b_ptr[col], qe = b_ptr[col + 1];
    const int la = pe - p, lb = qe - q;
    int cnt = 0;
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

}  // namespace r4d

using namespace r4d;

extern "C" int r4d_jaccard_f64(const int32_t* a_ptr_d, const int32_t* a_idx_d, int32_t na, int32_t a_nnz,
                               const int32_t* b_ptr_d, const int32_t* b_idx_d, int32_t nb, int32_t b_nnz, int32_t vocab,
                               int32_t zero_diag, double* out_d, void* stream) {
    R4D_REQUIRE(a_ptr_d && b_ptr_d && a_idx_d && b_idx_d && out_d, "jaccard: null pointer");
    R4D_REQUIRE(a_nnz >= 0 && b_nnz >= 0, "jaccard: negative nnz");
    if (a_nnz < 1) a_nnz = 1;                       // idx buffers hold >= 1 element by contract
    R4D_REQUIRE(na >= 0 && nb >= 0 && vocab >= 1, "jaccard: bad sizes na=%d nb=%d vocab=%d", na, nb, vocab);
    if (na == 0 || nb == 0) return R4D_OK;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes (SURVEY 8d B_jac): the f64 matrix out + both CSR inputs read once
    ProfScope prof(PK_JACCARD, 8.0 * na * (double)nb + 4.0 * (na + nb + 2) + 4.0 * ((double)a_nnz + b_nnz), s);
    if (vocab <= JAC_MAX_VOCAB_LDS) {
        const int col_tiles = cdiv(nb, 64);
        // enough row chunks for >= ~8 workgroups per CU, but >= 64 rows each so the table build amortises
        int chunks = max(1, min(cdiv(na, 64), cdiv(2048, col_tiles)));
        const int rows_per_block = cdiv(na, chunks);
        chunks = cdiv(na, rows_per_block);
        const size_t lds = ((size_t)vocab + 1) * sizeof(unsigned long long);
        if (lds > 64 * 1024) {
            static bool raised = false;     // opt in to > 64 KB dynamic LDS once
            if (!raised) {
                if (hipFuncSetAttribute((const void*)jaccard_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (JAC_MAX_VOCAB_LDS + 1) * 8) != hipSuccess) {
                    set_error("jaccard: cannot raise dynamic LDS limit");
                    return R4D_ERR_HIP;
                }
                raised = true;
            }
        }
        // a big table leaves room for one workgroup per CU only: give it 16 wavefronts instead of 4
        const int threads = lds > 80 * 1024 ? 1024 : (lds > 40 * 1024 ? 512 : 256);
        hipLaunchKernelGGL(jaccard_lds_kernel, dim3(col_tiles, chunks), dim3(threads), lds, s, a_ptr_d, a_idx_d, na, a_nnz,
                           b_ptr_d, b_idx_d, nb, vocab, zero_diag, rows_per_block, out_d);
        R4D_CHECK_LAUNCH("jaccard_lds");
    } else {
        hipLaunchKernelGGL(jaccard_merge_kernel, dim3((unsigned)((long long)cdiv(nb, 256) * na)), dim3(256), 0, s, a_ptr_d, a_idx_d, na, b_ptr_d,
                           b_idx_d, nb, zero_diag, out_d);
        R4D_CHECK_LAUNCH("jaccard_merge");
    }
    return R4D_OK;
}
