// Row / element kernels of the retriever TRAINING step's backward pass on gfx950: LayerNorm backward,
// gelu_new forward / backward, causal softmax backward, column sums (bias gradients), batched transpose (the A^T operand
// of the weight-gradient GEMMs), embedding scatter-add, mean-pool broadcast, squared-norm accumulation and the AdamW update.
// The reference gets all of these from torch autograd (loss.backward(), train/train_retriever.py:196-214) and
// transformers.AdamW (utils/model.py:80-102); what is computed here is the same calculus on the tensors of
// models/modeling_gpt2.py:140-235 -- checked against the reference's own autograd gradients (tests/golden/g8_training_step.npz).
#include <math.h>
#include "common.h"

namespace r4d {

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------ LayerNorm backward
// y = (x - mean) * rstd * w + b per row.  Given dy:  g = dy * w,  xh = (x - mean) * rstd,
//   dx = rstd * (g - mean_c(g) - xh * mean_c(g * xh))   [+ add[m,:] when `add` is given: the residual branch's gradient]
//   dw = sum_rows dy * xh,  db = sum_rows dy      (two deterministic stages: per-workgroup partial rows, then a column pass)
// One wavefront per row (d = 64 * nv <= 64 * NV), statistics recomputed from x (two-pass like the forward kernel).  A
// workgroup owns `rows_per_wg` consecutive rows (its four waves interleaved), so the partial count is bounded by the grid
// (<= LNB_MAX_WG), and the NEXT row's x / dy are in flight while the current row reduces: HBM-bound (3 reads + 1 write per
// element), not latency-bound.  NV is a template parameter so the per-lane arrays stay in registers at 4+ waves per SIMD.
constexpr int LNB_MAX_WG = 2048;
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ dy, const float* add, int rows, int rows_per_wg,
                                                     int d, float eps, float* dx, float* __restrict__ part_w,
                                                     float* __restrict__ part_b) {          // `add` may alias `dx` (in-place residual)
    extern __shared__ float red[];                               // [2][4][d]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = d >> 6;                                       // d % 64 == 0
    float aw[NV], ab[NV], wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { aw[i] = 0.f; ab[i] = 0.f; wv[i] = i < nv ? w[lane + 64 * i] : 0.f; }
    const long long rbeg = (long long)blockIdx.x * rows_per_wg, rend = min((long long)rows, rbeg + rows_per_wg);
    long long row = rbeg + wid;
    float xn[NV], dn[NV];
    if (row < rend) {
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i < nv) { xn[i] = x[row * d + lane + 64 * i]; dn[i] = dy[row * d + lane + 64 * i]; }
    }
    for (; row < rend; row += 4) {                               // wave-uniform
        float xv[NV], dv[NV], av[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i < nv) { xv[i] = xn[i]; dv[i] = dn[i]; s += xv[i]; }
        if (add) {
#pragma unroll
            for (int i = 0; i < NV; ++i) if (i < nv) av[i] = add[row * d + lane + 64 * i];
        }
        if (row + 4 < rend) {                                    // next row of this wave: in flight under the reductions
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (i < nv) { xn[i] = x[(row + 4) * d + lane + 64 * i]; dn[i] = dy[(row + 4) * d + lane + 64 * i]; }
        }
        const float mean = wave_sum_t(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i < nv) { xv[i] -= mean; q += xv[i] * xv[i]; }
        const float rstd = rsqrtf(wave_sum_t(q) / (float)d + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i < nv) {
                xv[i] *= rstd;                                   // xh
                const float gi = dv[i] * wv[i];
                sg += gi; sgx += gi * xv[i];
                aw[i] += dv[i] * xv[i]; ab[i] += dv[i];
                dv[i] = gi;
            }
        const float mg = wave_sum_t(sg) / (float)d, mgx = wave_sum_t(sgx) / (float)d;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i < nv) {
                float v = rstd * (dv[i] - mg - xv[i] * mgx);
                if (add) v += av[i];
                dx[row * d + lane + 64 * i] = v;
            }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) if (i < nv) { red[wid * d + lane + 64 * i] = aw[i]; red[(4 + wid) * d + lane + 64 * i] = ab[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {                 // waves combined in a fixed order
        part_w[(long long)blockIdx.x * d + c] = (red[c] + red[d + c]) + (red[2 * d + c] + red[3 * d + c]);
        part_b[(long long)blockIdx.x * d + c] = (red[4 * d + c] + red[5 * d + c]) + (red[6 * d + c] + red[7 * d + c]);
    }
}

// out[c] (+)= sum over rows of x[r, c] (partials of ln_bwd_kernel / colsum_partial_kernel): a workgroup owns 64 columns,
// its 16 waves take rows w, w+16, ... (independent loads), combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ x, int rows, int n, float* __restrict__ out,
                                                            int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int r = wid;
        for (; r + 16 < rows; r += 32) { s0 += x[(long long)r * n + c]; s1 += x[(long long)(r + 16) * n + c]; }
        if (r < rows) s0 += x[(long long)r * n + c];
    }
    red[wid][lane] = s0 + s1;
    __syncthreads();
    if (wid == 0 && c < n) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][lane];
        out[c] = accumulate ? out[c] + s : s;
    }
}

// partial[b, c] = sum of x[r, c] over the rows_per_block rows of block b (4 waves x a quarter each, coalesced across c)
constexpr int CS_MAX_BLOCKS = 256;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int rows, int n, int ld, int rows_per_block,
                                                             float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int q = rows_per_block >> 2;                           // rows_per_block % 4 == 0
    const long long r0 = (long long)blockIdx.y * rows_per_block + (long long)wid * q;
    const long long r1 = min((long long)rows, r0 + q);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < n) {
        long long r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += x[r * ld + c]; s1 += x[(r + 1) * ld + c]; s2 += x[(r + 2) * ld + c]; s3 += x[(r + 3) * ld + c];
        }
        for (; r < r1; ++r) s0 += x[r * ld + c];
    }
    red[wid][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wid == 0 && c < n) partial[(long long)blockIdx.y * n + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------------------------ gelu_new
__device__ __forceinline__ float gelu_new_t(float x) {           // modeling_gpt2.py:25,206 (tanh form)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_new_grad(float x) {
    const float c = 0.7978845608028654f;
    const float u = c * (x + 0.044715f * x * x * x), t = tanhf(u);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
}
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ pre, long long n, float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = gelu_new_t(pre[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ pre, const float* dy, long long n,
                                                       float* dx) {                              // dx may alias dy
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * gelu_new_grad(pre[i]);
}

// ------------------------------------------------------------------------------------ causal softmax backward
// P = softmax of the causal row (zeros right of the diagonal), logits were divided by sqrt(hd) BEFORE the softmax
// (modeling_gpt2.py:143), so dLogit_raw = dS / scale_div with dS_ij = P_ij (dP_ij - sum_k P_ik dP_ik).  In place on dP;
// every column right of the diagonal up to ld is WRITTEN as zero (the dP buffer holds unwritten memory there).
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long long nrows_total,
                                                          int T, int ld, float scale_div) {
    const int lane = threadIdx.x & 63;
    const long long gr = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= nrows_total) return;
    const int i = (int)(gr % T);
    const long long base = (gr / T) * (long long)T * ld + (long long)i * ld;
    const int n = i + 1;
    float p[16], g[16];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = lane + 64 * c;
        const bool ok = j < n;
        p[c] = ok ? P[base + min(j, ld - 1)] : 0.f;
        g[c] = ok ? dP[base + min(j, ld - 1)] : 0.f;
        s += p[c] * g[c];
    }
    s = wave_sum_t(s);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = lane + 64 * c;
        if (j < ld) dP[base + j] = (j < n) ? p[c] * (g[c] - s) / scale_div : 0.f;
    }
}

// ------------------------------------------------------------------------------------ batched transpose
// out[z][c][r] = in[z][r][c] for r < rows, c < cols; columns r in [rows, ld_out) of every out row are written as zero
// (they are the padded contraction range of the GEMM that reads `out` as its A operand).  32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int rows, int cols, long long ld_in,
                                                        long long stride_in, float* __restrict__ out, long long ld_out,
                                                        long long stride_out) {
    __shared__ float tile[32][33];
    const float* src = in + (long long)blockIdx.z * stride_in;
    float* dst = out + (long long)blockIdx.z * stride_out;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int r = r0 + ty + k, c = c0 + tx;
        tile[ty + k][tx] = (r < rows && c < cols) ? src[(long long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int c = c0 + ty + k, r = r0 + tx;                  // out row = c, out column = r
        if (c < cols && r < ld_out) dst[(long long)c * ld_out + r] = tile[tx][ty + k];
    }
}

// ------------------------------------------------------------------------------------ embedding backward
// x[m,:] = wte[ids[m]] + wpe[t(m)]  (modeling_gpt2.py:463-469)  =>  dwte[ids[m]] += dx[m], dwpe[t(m)] += dx[m].
// torch's embedding backward on a GPU uses unordered float atomics (run-to-run differences in the last bits); here both sums
// are DETERMINISTIC:
//   * dwpe: the rows of position t are row t of every sequence of a batch -- a plain column sum over the sequences in a fixed
//     order, one thread per (t, column), the batches of a step one launch after the other;
//   * dwte: which rows share a token is data dependent, so the contributions are added as 64-bit FIXED-POINT integers
//     (value * 2^S, round to nearest): integer addition is associative, the atomics can land in any order and the sum is the
//     same bits every time; a second kernel converts the touched table back to fp32.
//     THE SCALE FOLLOWS THE DATA (round 5, second form): a pre-pass takes max |dx| over the step's token rows (a maximum does not
//     depend on the order either: same bits every run) and S = 62 - ceil(log2 M) - e with max < 2^e, M = the token rows that add
//     into the table: every term is below 2^(62 - ceil(log2 M)), at most M of them meet in one element, so the sum stays below
//     2^62 and CANNOT wrap -- whatever the magnitudes (a fixed 2^17 bound with a 2^44 scale let four same-sign terms
//     wrap unnoticed).  Resolution 2^-(61 - ceil(log2 M)) of the largest contribution: 2^-44 at 131,072 rows, twenty bits below
//     fp32's own.  (The first form of this round kept a 2^40 scale and bounded a term by 2^22 / M = 50-140, poisoning the step
//     beyond it: correct, but a bound the reference does not have.  This form has none.)
//     A NON-FINITE contribution has no fixed-point image: the maximum is then Inf / NaN, the POISON word behind the table is set and
//     the conversion writes NaN into the whole gradient -- a diverged step stays as loud as with float atomics (NaN gradient norm,
//     NaN parameters after the clip) instead of turning into a finite, wrong update.
constexpr int EMB_SUM_BITS = 62;
__device__ __forceinline__ int emb_scale_exp(unsigned max_bits, int lg_rows) {          // S: |g| 2^S M < 2^62 for every |g| <= max
    const int e = (int)((max_bits >> 23) & 255u) - 126;                                  // max < 2^e (exponent field 0: a subnormal)
    return EMB_SUM_BITS - lg_rows - e;
}
// max |x| as the bit pattern of a non-negative float (NaN counts as Inf); *max_bits zeroed by the caller
__global__ __launch_bounds__(256) void embedding_absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ max_bits) {
    unsigned m = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = fabsf(x[i]);
        const unsigned b = (v <= 3.402823466e38f) ? __float_as_uint(v) : 0x7f800000u;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(max_bits, m);
}
// acc_wte: [vocab * d] sums, then the poison word, then the max word (low 32 bits)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                            long long rows, int d, int vocab, unsigned long long* __restrict__ acc_wte,
                                                            int lg_rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const long long n = (long long)vocab * d;
    const unsigned mb = (unsigned)acc_wte[n + 1];
    if (mb >= 0x7f800000u) {                                                       // some contribution of this step is NaN / Inf
        if (row == 0 && lane == 0) atomicOr(acc_wte + n, 1ull);
        return;
    }
    const long long id = ids[row];
    if (id < 0 || id >= vocab) return;
    const int S = emb_scale_exp(mb, lg_rows);
    for (int c = lane; c < d; c += 64) {
        const long long q = __double2ll_rn(ldexp((double)dx[row * d + c], S));
        atomicAdd(acc_wte + id * d + c, (unsigned long long)q);
    }
}
// dwpe[t, c] (+)= sum over the B sequences of dx[b * T + t, c], b ascending; `first`: overwrite instead of add
__global__ __launch_bounds__(256) void wpe_bwd_kernel(const float* __restrict__ dx, int B, int T, int d, int first, float* __restrict__ dwpe) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * d) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[(long long)b * T * d + i];
    dwpe[i] = first ? s : dwpe[i] + s;
}
__global__ __launch_bounds__(256) void embedding_fix_to_f32_kernel(const unsigned long long* __restrict__ acc, long long n, int lg_rows,
                                                                   float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned mb = (unsigned)acc[n + 1];
    out[i] = (acc[n] || mb >= 0x7f800000u) ? __builtin_nanf("") : (float)ldexp((double)(long long)acc[i], -emb_scale_exp(mb, lg_rows));
}

// dh[row, :] = d_pool[seq(row), :] / T   (torch.mean(h, dim=1) backward, train_retriever.py:181-183)
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ d_pool, long long rows, int T, int d,
                                                           float* __restrict__ dh) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * d) return;
    const long long row = i / d;
    dh[i] = d_pool[(row / T) * d + i % d] / (float)T;
}

// ------------------------------------------------------------------------------------ dropout
// nn.Dropout of the training forward (modeling_gpt2.py:114-115,153,194,207-212,337,427): keep with probability 1 - p, scale the
// kept values by 1 / (1 - p).  The reference draws its masks from torch's device RNG stream, which no other implementation can
// reproduce; this build draws them from a COUNTER-BASED generator -- Philox-4x32-10 (Salmon et al., SC'11) keyed by the run's
// seed, counter = (element index / 4, site, step low, step high) -- so a mask is a pure function of (seed, step, site, index):
// nothing is stored between forward and backward (the backward regenerates it), the result does not depend on the launch
// geometry, and the oracle restates the generator in numpy (oracle/train_ref.py philox_keep) to check forward AND gradients.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
// out[i] = (resid ? resid[i] : 0) + keep(base + i) * x[i] / (1 - p), four elements (one Philox block) per thread; n, base % 4 == 0
__global__ __launch_bounds__(256) void dropout_kernel(const float* x, const float* resid, long long n4, float* out,
                                                      unsigned threshold, float scale, DropKey key, unsigned site,
                                                      unsigned long long base4) {   // resid / x may alias out
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const unsigned long long c = base4 + (unsigned long long)i;
    const uint4 r = philox4x32_10(make_uint4((unsigned)c, site + ((unsigned)(c >> 32) << 16), key.step_lo, key.step_hi),
                                  make_uint2(key.seed_lo, key.seed_hi));
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o = make_float4(r.x >= threshold ? v.x * scale : 0.f, r.y >= threshold ? v.y * scale : 0.f,
                           r.z >= threshold ? v.z * scale : 0.f, r.w >= threshold ? v.w * scale : 0.f);
    if (resid) {
        const float4 q = reinterpret_cast<const float4*>(resid)[i];
        o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
    }
    reinterpret_cast<float4*>(out)[i] = o;
}

int launch_dropout(const float* x, const float* resid, long long n, float* out, float p, DropKey key, unsigned site,
                   unsigned long long base, hipStream_t s) {
    R4D_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%g outside [0, 1)", (double)p);
    R4D_REQUIRE(n % 4 == 0 && base % 4 == 0 && site < 65536u, "dropout: n=%lld base=%llu must be multiples of 4", n, base);
    if (n == 0) return R4D_OK;
    const unsigned threshold = (unsigned)((double)p * 4294967296.0);               // keep iff u32 >= threshold
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, resid, n / 4, out, threshold,
                       1.0f / (1.0f - p), key, site, base / 4);
    R4D_CHECK_LAUNCH("dropout");
    return R4D_OK;
}

// ------------------------------------------------------------------------------------ optimizer
// accum[0] += sum x^2 (the total feeds clip_grad_norm_, train_retriever.py:210).  Two stages, no atomics: the value must be
// the same bits on every data-parallel rank (same gradients after the all-reduce), or the clip coefficient -- and with it the
// weights -- drift apart between ranks.  part[b] = workgroup b's grid-strided share; then one workgroup adds them in order.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum_t(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ accum) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = wave_sum_t(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) accum[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

// transformers.AdamW.step (the optimizer utils/model.py:80-93 builds; third-party, restated from its published update):
//   g <- g * min(1, max_norm / (sqrt(sumsq) + 1e-6))                      (torch.nn.utils.clip_grad_norm_)
//   m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2
//   p <- p - lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps);   p <- p - lr * wd * p        (decoupled decay, after)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr_step, float b1, float omb1, float b2,
                                                    float omb2, float eps, float lr_wd, const float* __restrict__ sumsq,
                                                    float max_norm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float coef = 1.f;
    if (sumsq && max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + omb1 * gi;                    // scalars rounded to f32 like torch's mul_ / add_(alpha=)
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    float pi = p[i] - lr_step * (mi / (sqrtf(vi) + eps));
    if (lr_wd > 0.f) pi -= lr_wd * pi;
    p[i] = pi;
}

// ------------------------------------------------------------------------------------ launchers (used by train.hip)
static inline void ln_bwd_grid(int rows, int& nwg, int& rows_per_wg) {
    rows_per_wg = cdiv(cdiv(rows, LNB_MAX_WG), 4) * 4;               // every wave of a workgroup gets whole rows
    if (rows_per_wg < 8) rows_per_wg = 8;
    nwg = cdiv(rows, rows_per_wg);
}
size_t ln_bwd_scratch_floats(int rows, int d) {
    int nwg, rpw;
    ln_bwd_grid(rows > 0 ? rows : 1, nwg, rpw);
    return (size_t)2 * nwg * d;
}

template <int NV>
static int launch_ln_bwd_nv(const float* x, const float* w, const float* dy, const float* add, int rows, int rpw, int nwg, int d,
                            float eps, float* dx, float* pw, float* pb, hipStream_t s) {
    const size_t lds = (size_t)8 * d * sizeof(float);
    if (lds > 48 * 1024) {
        static bool raised = false;
        if (!raised) {
            R4D_HIP(hipFuncSetAttribute((const void*)ln_bwd_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 4));
            raised = true;
        }
    }
    hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3(nwg), dim3(256), lds, s, x, w, dy, add, rows, rpw, d, eps, dx, pw, pb);
    R4D_CHECK_LAUNCH("ln_bwd");
    return R4D_OK;
}

int launch_ln_bwd(const float* x, const float* w, const float* dy, const float* add, int rows, int d, float eps, float* dx,
                  float* dw, float* db, float* scratch, int accumulate, hipStream_t s) {
    R4D_REQUIRE(d % 64 == 0 && d <= 2048, "layernorm_bwd: d=%d must be a multiple of 64 and <= 2048", d);
    if (rows <= 0) return R4D_OK;
    int nwg, rpw;
    ln_bwd_grid(rows, nwg, rpw);
    float* pw = scratch;
    float* pb = scratch + (size_t)nwg * d;
    const int nv = d / 64;
    int rc;
    if (nv <= 4) rc = launch_ln_bwd_nv<4>(x, w, dy, add, rows, rpw, nwg, d, eps, dx, pw, pb, s);
    else if (nv <= 8) rc = launch_ln_bwd_nv<8>(x, w, dy, add, rows, rpw, nwg, d, eps, dx, pw, pb, s);
    else if (nv <= 12) rc = launch_ln_bwd_nv<12>(x, w, dy, add, rows, rpw, nwg, d, eps, dx, pw, pb, s);
    else if (nv <= 16) rc = launch_ln_bwd_nv<16>(x, w, dy, add, rows, rpw, nwg, d, eps, dx, pw, pb, s);
    else rc = launch_ln_bwd_nv<32>(x, w, dy, add, rows, rpw, nwg, d, eps, dx, pw, pb, s);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(d, 64)), dim3(1024), 0, s, pw, nwg, d, dw, accumulate);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(d, 64)), dim3(1024), 0, s, pb, nwg, d, db, accumulate);
    R4D_CHECK_LAUNCH("ln_bwd colsum");
    return R4D_OK;
}

static inline void colsum_grid(long long rows, int& nb, int& rows_per_block) {
    rows_per_block = (int)cdiv((int)cdiv((int)rows, CS_MAX_BLOCKS), 4) * 4;
    if (rows_per_block < 64) rows_per_block = 64;
    nb = cdiv((int)rows, rows_per_block);
}
size_t colsum_scratch_floats(long long rows, int n) {
    int nb, rpb;
    colsum_grid(rows > 0 ? rows : 1, nb, rpb);
    return (size_t)nb * n;
}

int launch_colsum(const float* x, long long rows, int n, int ld, float* out, float* scratch, int accumulate, hipStream_t s) {
    if (rows <= 0 || n <= 0) return R4D_OK;
    int nb, rpb;
    colsum_grid(rows, nb, rpb);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(n, 64), nb), dim3(256), 0, s, x, (int)rows, n, ld, rpb, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, s, scratch, nb, n, out, accumulate);
    R4D_CHECK_LAUNCH("colsum");
    return R4D_OK;
}

int launch_gelu_fwd(const float* pre, long long n, float* y, hipStream_t s) {
    if (n <= 0) return R4D_OK;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, n, y);
    R4D_CHECK_LAUNCH("gelu_fwd");
    return R4D_OK;
}
int launch_gelu_bwd(const float* pre, const float* dy, long long n, float* dx, hipStream_t s) {
    if (n <= 0) return R4D_OK;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pre, dy, n, dx);
    R4D_CHECK_LAUNCH("gelu_bwd");
    return R4D_OK;
}
int launch_softmax_bwd(const float* P, float* dP, int nbh, int T, int ld, float scale_div, hipStream_t s) {
    R4D_REQUIRE(T >= 1 && T <= 1024 && ld >= T && ld <= 1024, "softmax_bwd: T=%d ld=%d out of range", T, ld);
    const long long rows = (long long)nbh * T;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, P, dP, rows, T, ld, scale_div);
    R4D_CHECK_LAUNCH("softmax_bwd");
    return R4D_OK;
}
int launch_transpose(const float* in, int rows, int cols, long long ld_in, long long stride_in, float* out, long long ld_out,
                     long long stride_out, int nbatch, hipStream_t s) {
    R4D_REQUIRE(rows >= 1 && cols >= 1 && ld_in >= cols && ld_out >= rows && nbatch >= 1 && nbatch <= 65535,
                "transpose: bad shape rows=%d cols=%d", rows, cols);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv((int)ld_out, 32), nbatch), dim3(256), 0, s, in, rows, cols,
                       ld_in, stride_in, out, ld_out, stride_out);
    R4D_CHECK_LAUNCH("transpose");
    return R4D_OK;
}
static int emb_lg_rows(long long table_rows) {
    int lg = 0;
    while ((1LL << lg) < table_rows) ++lg;
    return lg;
}
// once per backward call, before the batches: max |dx| over ALL token rows of the step into the max word behind the table
// (`acc_wte` [vocab * d] + poison word + max word, zeroed by the caller)
int launch_embedding_absmax(const float* dx, long long n, unsigned long long* acc_wte, long long table_elems, hipStream_t s) {
    if (n <= 0) return R4D_OK;
    const unsigned grid = (unsigned)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
    hipLaunchKernelGGL(embedding_absmax_kernel, dim3(grid), dim3(256), 0, s, dx, n, reinterpret_cast<unsigned*>(acc_wte + table_elems + 1));
    R4D_CHECK_LAUNCH("embedding_absmax");
    return R4D_OK;
}
// one batch [B, T] of a step: token part into the fixed-point table `acc_wte` [vocab, d] (zeroed by the caller once per step),
// position part into dwpe rows [0, T) (`first_group`: the step's first batch overwrites, later ones add -- in call order)
// `table_rows`: ALL token rows that add into this table before it is converted (every batch of the backward call)
int launch_embedding_bwd(const float* dx, const int64_t* ids, int B, int T, int d, int vocab, unsigned long long* acc_wte, float* dwpe,
                         int first_group, long long table_rows, hipStream_t s) {
    const long long rows = (long long)B * T;
    if (rows <= 0) return R4D_OK;
    R4D_REQUIRE(table_rows >= rows, "embedding_bwd: table_rows %lld < rows %lld", table_rows, rows);
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, dx, ids, rows, d, vocab, acc_wte,
                       emb_lg_rows(table_rows));
    R4D_CHECK_LAUNCH("embedding_bwd");
    hipLaunchKernelGGL(wpe_bwd_kernel, dim3((unsigned)(((long long)T * d + 255) / 256)), dim3(256), 0, s, dx, B, T, d, first_group, dwpe);
    R4D_CHECK_LAUNCH("wpe_bwd");
    return R4D_OK;
}
int launch_embedding_fix_to_f32(const unsigned long long* acc, long long n, long long table_rows, float* out, hipStream_t s) {
    if (n <= 0) return R4D_OK;
    hipLaunchKernelGGL(embedding_fix_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, n, emb_lg_rows(table_rows), out);
    R4D_CHECK_LAUNCH("embedding_fix_to_f32");
    return R4D_OK;
}
int launch_meanpool_bwd(const float* d_pool, long long rows, int T, int d, float* dh, hipStream_t s) {
    if (rows <= 0) return R4D_OK;
    hipLaunchKernelGGL(meanpool_bwd_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, s, d_pool, rows, T, d, dh);
    R4D_CHECK_LAUNCH("meanpool_bwd");
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

size_t r4d_layernorm_bwd_workspace_bytes(int32_t rows, int32_t d) {
    if (rows <= 0 || d <= 0) return 0;
    return ln_bwd_scratch_floats(rows, d) * sizeof(float) + 256;
}
int r4d_layernorm_bwd_f32(const float* x_d, const float* w_d, const float* dy_d, const float* add_d, int32_t rows, int32_t d,
                          float eps, float* dx_d, float* dw_d, float* db_d, void* workspace_d, size_t workspace_bytes,
                          void* stream) {
    R4D_REQUIRE(x_d && w_d && dy_d && dx_d && dw_d && db_d && workspace_d, "layernorm_bwd: null pointer");
    R4D_REQUIRE(workspace_bytes >= r4d_layernorm_bwd_workspace_bytes(rows, d), "layernorm_bwd: workspace too small");
    return launch_ln_bwd(x_d, w_d, dy_d, add_d, rows, d, eps, dx_d, dw_d, db_d, (float*)workspace_d, 0, (hipStream_t)stream);
}
int r4d_gelu_new_f32(const float* pre_d, int64_t n, float* y_d, void* stream) {
    R4D_REQUIRE(pre_d && y_d && n >= 0, "gelu_new: bad arguments");
    return launch_gelu_fwd(pre_d, n, y_d, (hipStream_t)stream);
}
int r4d_gelu_new_bwd_f32(const float* pre_d, const float* dy_d, int64_t n, float* dx_d, void* stream) {
    R4D_REQUIRE(pre_d && dy_d && dx_d && n >= 0, "gelu_new_bwd: bad arguments");
    return launch_gelu_bwd(pre_d, dy_d, n, dx_d, (hipStream_t)stream);
}
int r4d_causal_softmax_bwd_f32(const float* p_d, float* dp_d, int32_t nbh, int32_t T, int32_t ld, float scale_div, void* stream) {
    R4D_REQUIRE(p_d && dp_d && nbh >= 1, "causal_softmax_bwd: bad arguments");
    return launch_softmax_bwd(p_d, dp_d, nbh, T, ld, scale_div, (hipStream_t)stream);
}
int r4d_dropout_f32(const float* x_d, const float* resid_d, int64_t n, float* out_d, float p, uint64_t seed, uint64_t step,
                    uint32_t site, uint64_t index_base, void* stream) {
    R4D_REQUIRE(x_d && out_d && n >= 0, "dropout: bad arguments");
    const DropKey key{(unsigned)seed, (unsigned)(seed >> 32), (unsigned)step, (unsigned)(step >> 32)};
    return launch_dropout(x_d, resid_d, n, out_d, p, key, site, index_base, (hipStream_t)stream);
}
int r4d_sumsq_accumulate_f32(const float* x_d, int64_t n, float* accum_d, void* stream) {
    R4D_REQUIRE(x_d && accum_d && n >= 0, "sumsq: bad arguments");
    if (n == 0) return R4D_OK;
    const unsigned grid = (unsigned)((n + 255) / 256 < SUMSQ_BLOCKS ? (n + 255) / 256 : SUMSQ_BLOCKS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_d, n, accum_d + 1);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, accum_d + 1, (int)grid, accum_d);
    R4D_CHECK_LAUNCH("sumsq");
    return R4D_OK;
}
int r4d_adamw_step_f32(float* p_d, const float* g_d, float* m_d, float* v_d, int64_t n, double lr, double beta1, double beta2,
                       double eps, double weight_decay, int32_t step, const float* grad_sumsq_d, float max_grad_norm,
                       void* stream) {
    R4D_REQUIRE(p_d && g_d && m_d && v_d && n >= 0 && step >= 1, "adamw: bad arguments");
    if (n == 0) return R4D_OK;
    // python-float (double) hyper-parameters, every derived scalar formed in double and rounded to f32 once, like the
    // reference's optimizer does through torch's scalar arguments
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float lr_step = (float)(lr * sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p_d, g_d, m_d, v_d, n,
                       lr_step, (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr * weight_decay), grad_sumsq_d, max_grad_norm);
    R4D_CHECK_LAUNCH("adamw");
    return R4D_OK;
}

}  // extern "C"
