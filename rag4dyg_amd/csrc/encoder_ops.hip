// HBM-bound row kernels of the encoder (gfx950): embedding gather + LayerNorm, LayerNorm, causal
// softmax, ln_f + mean-pool.  One 64-lane wavefront owns one row; reductions are wavefront shuffles
// (no LDS, no atomics -> bitwise reproducible run to run).
#include <string.h>
#include "common.h"
#include "h2.h"

namespace r4d {

constexpr int MAXV = 32;   // values per lane: d <= 64*32 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Normalise the row held in v[0..nv) (element c = lane + 64*i), nn.LayerNorm semantics
// (biased variance, eps inside the sqrt) -- modeling_gpt2.py:219,221,339.
__device__ __forceinline__ void ln_row(float (&v)[MAXV], int nv, int d, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (i < nv) s += v[i];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (i < nv) { const float t = v[i] - mean; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (i < nv) v[i] = (v[i] - mean) * rstd;
}

// y[m,:] = LayerNorm(x[m,:]), one wavefront per row
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x_in, int rows, int d, const float* __restrict__ w,
                                                 const float* __restrict__ b, float eps, float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = d >> 6;
    float v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) if (i < nv) v[i] = x_in[(long long)row * d + lane + 64 * i];
    ln_row(v, nv, d, eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) y_out[(long long)row * d + lane + 64 * i] = v[i] * w[lane + 64 * i] + b[lane + 64 * i];
}

// Same, 16 bytes per lane and instruction (d % 256 == 0): lane owns columns 4*lane + 256*i .. +3
// y as f16x2 LINES (gemm_h2p.hip; round 5): the four values a lane holds are four consecutive k of one 128-byte line -- 8 bytes of its hi
// half, 8 bytes of its lo' half (64 bytes further): the same bytes as the fp32 row, split where they are produced
__device__ __forceinline__ void store_lines4(float* y_rows, long long row, int d, int col, float a0, float a1, float a2, float a3) {
    unsigned h0, l0, h1, l1;
    split2_pair<true>(a0, a1, h0, l0);
    split2_pair<true>(a2, a3, h1, l1);
    unsigned char* dst = reinterpret_cast<unsigned char*>(y_rows) + row * (long long)d * 4 + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(l0, l1);
}

template <int NQ, bool LINES = false>
__global__ __launch_bounds__(256) void ln4_kernel(const float* __restrict__ x_in, int rows, int d, const float* __restrict__ w,
                                                  const float* __restrict__ b, float eps, float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nq = d >> 8;
    const float4* xr = reinterpret_cast<const float4*>(x_in + (long long)row * d) + lane;
    float4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = (i < nq) ? xr[64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) s_ += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s_) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if (i < nq) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float4* yr = reinterpret_cast<float4*>(y_out + (long long)row * d) + lane;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if (i < nq) {
            const float4 ww = reinterpret_cast<const float4*>(w)[lane + 64 * i], bb = reinterpret_cast<const float4*>(b)[lane + 64 * i];
            const float4 y4 = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                                          (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
            if constexpr (LINES) store_lines4(y_out, row, d, 4 * lane + 256 * i, y4.x, y4.y, y4.z, y4.w);
            else yr[64 * i] = y4;
        }
}

// LayerNorm whose output is the f16x2-line image of the normalised rows (the A operand of gemm_h2p.hip); d % 256 == 0
bool layernorm_lines_supported(int d) { return d % 256 == 0 && d <= 64 * MAXV; }
int launch_layernorm_lines(const float* x, const float* w, const float* b, int rows, int d, float eps, unsigned short* y_lines, hipStream_t s) {
    R4D_REQUIRE(layernorm_lines_supported(d) && (((uintptr_t)x | (uintptr_t)y_lines | (uintptr_t)w | (uintptr_t)b) & 15) == 0,
                "layernorm_lines: d=%d must be a multiple of 256 and the pointers 16-byte aligned", d);
    if (rows <= 0) return R4D_OK;
    ProfScope prof(PK_LAYERNORM, 8.0 * rows * d, s);            // bytes: read x + write the lines (4 bytes per element, as fp32)
    float* y = reinterpret_cast<float*>(y_lines);
    if (d <= 512) { R4D_BRANCH(LN4_2); hipLaunchKernelGGL((ln4_kernel<2, true>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y); }
    else if (d <= 1024) { R4D_BRANCH(LN4_4); hipLaunchKernelGGL((ln4_kernel<4, true>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y); }
    else { R4D_BRANCH(LN4_8); hipLaunchKernelGGL((ln4_kernel<8, true>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y); }
    R4D_CHECK_LAUNCH("layernorm_lines");
    return R4D_OK;
}

int launch_layernorm(const float* x, const float* w, const float* b, int rows, int d, float eps, float* y,
                     hipStream_t s) {
    R4D_REQUIRE(d % 64 == 0 && d <= 64 * MAXV, "layernorm: d=%d must be a multiple of 64 and <= %d", d, 64 * MAXV);
    if (rows <= 0) return R4D_OK;
    ProfScope prof(PK_LAYERNORM, 8.0 * rows * d, s);            // bytes: read x + write y
    const bool vec = d % 256 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)b) & 15) == 0;
    if (vec && d <= 512) R4D_BRANCH(LN4_2); else if (vec && d <= 1024) R4D_BRANCH(LN4_4); else if (vec) R4D_BRANCH(LN4_8); else R4D_BRANCH(LN_GENERIC);
    if (vec && d <= 512) hipLaunchKernelGGL(ln4_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y);
    else if (vec && d <= 1024) hipLaunchKernelGGL(ln4_kernel<4>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y);
    else if (vec) hipLaunchKernelGGL(ln4_kernel<8>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y);
    else
    hipLaunchKernelGGL(ln_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, w, b, eps, y);
    R4D_CHECK_LAUNCH("layernorm");
    return R4D_OK;
}

// Prefix tables of a RowGroups launch (kernel argument, by value; the batch of a row / sequence is found by a short scan).
struct RowTable {
    int n;
    int T[ATT_MAXG];
    int seq_prefix[ATT_MAXG + 1];            // first sequence of each batch
    long long row_prefix[ATT_MAXG + 1];      // first token row of each batch
    long long part_prefix[ATT_MAXG + 1];     // first partial row (B * chunks) of each batch in the mean-pool scratch
    __device__ __forceinline__ long long row0_of(int g, int bidx) const { return row_prefix[g] + (long long)bidx * T[g]; }
};
struct RowInputs { const int64_t* ids[ATT_MAXG]; const float* emb[ATT_MAXG]; };

static RowTable make_table(const RowGroups& G) {
    RowTable t;
    t.n = G.n;
    t.seq_prefix[0] = 0; t.row_prefix[0] = 0; t.part_prefix[0] = 0;
    for (int g = 0; g < ATT_MAXG; ++g) {
        const int B = g < G.n ? G.B[g] : 0, T = g < G.n ? G.T[g] : 0;
        t.T[g] = T;
        t.seq_prefix[g + 1] = t.seq_prefix[g] + B;
        t.row_prefix[g + 1] = t.row_prefix[g] + (long long)B * T;
        t.part_prefix[g + 1] = t.part_prefix[g] + (long long)B * cdiv(T > 0 ? T : 1, LNF_ROWS_PER_CHUNK) * (B > 0);
    }
    return t;
}

// x_out[m,:] = (ids ? wte[ids[m]] : inputs_embeds[m]) + wpe[position of m in ITS batch]   (modeling_gpt2.py:463-469)
// y_out[m,:] = LayerNorm(x_out[m,:])   (ln_1 of block 0) -- all batches of a fused call in one launch
__global__ __launch_bounds__(256) void embed_ln_groups_kernel(const RowTable G, const RowInputs in,
                                                              const float* __restrict__ wte, const float* __restrict__ wpe,
                                                              int vocab, int d, const float* __restrict__ w,
                                                              const float* __restrict__ b, float eps,
                                                              float* __restrict__ x_out, float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= G.row_prefix[G.n]) return;
    int g = 0;
    while (g + 1 < G.n && row >= G.row_prefix[g + 1]) ++g;
    const long long local = row - G.row_prefix[g];
    const int nv = d >> 6;
    float v[MAXV];
    const float* src;
    bool bad = false;
    if (in.ids[g]) {
        const long long id = in.ids[g][local];
        bad = id < 0 || id >= vocab;                     // out-of-vocabulary id: poison the row, never fault
        src = wte + (bad ? 0 : id) * (long long)d;
    } else {
        src = in.emb[g] + local * d;
    }
    const float* pe = wpe + (local % G.T[g]) * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const float e = src[lane + 64 * i] + pe[lane + 64 * i];
            v[i] = bad ? __builtin_nanf("") : e;
            x_out[row * d + lane + 64 * i] = v[i];
        }
    ln_row(v, nv, d, eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) y_out[row * d + lane + 64 * i] = v[i] * w[lane + 64 * i] + b[lane + 64 * i];
}

// the same with 16-byte lanes (d % 256 == 0, every pointer 16-byte aligned): a quarter of the memory instructions of the 4-byte form,
// which held this write-bound kernel at 2.5 TB/s (ln4_kernel's layout and reduction order)
template <int NQ, bool LINES = false>
__global__ __launch_bounds__(256) void embed_ln4_groups_kernel(const RowTable G, const RowInputs in,
                                                               const float* __restrict__ wte, const float* __restrict__ wpe,
                                                               int vocab, int d, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps,
                                                               float* __restrict__ x_out, float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= G.row_prefix[G.n]) return;
    int g = 0;
    while (g + 1 < G.n && row >= G.row_prefix[g + 1]) ++g;
    const long long local = row - G.row_prefix[g];
    const int nq = d >> 8;
    const float* src;
    bool bad = false;
    if (in.ids[g]) {
        const long long id = in.ids[g][local];
        bad = id < 0 || id >= vocab;                     // out-of-vocabulary id: poison the row, never fault
        src = wte + (bad ? 0 : id) * (long long)d;
    } else {
        src = in.emb[g] + local * d;
    }
    const float4* s4 = reinterpret_cast<const float4*>(src) + lane;
    const float4* p4 = reinterpret_cast<const float4*>(wpe + (local % G.T[g]) * d) + lane;
    float4* xr = reinterpret_cast<float4*>(x_out + row * d) + lane;
    float4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nq) {
            const float4 e = s4[64 * i], q = p4[64 * i];
            const float nan = __builtin_nanf("");
            v[i] = bad ? make_float4(nan, nan, nan, nan) : make_float4(e.x + q.x, e.y + q.y, e.z + q.z, e.w + q.w);
            xr[64 * i] = v[i];
        }
    }
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i) s_ += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s_) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if (i < nq) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
    float4* yr = reinterpret_cast<float4*>(y_out + row * d) + lane;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
        if (i < nq) {
            const float4 ww = reinterpret_cast<const float4*>(w)[lane + 64 * i], bb = reinterpret_cast<const float4*>(b)[lane + 64 * i];
            const float4 y4 = make_float4((v[i].x - mean) * rstd * ww.x + bb.x, (v[i].y - mean) * rstd * ww.y + bb.y,
                                          (v[i].z - mean) * rstd * ww.z + bb.z, (v[i].w - mean) * rstd * ww.w + bb.w);
            if constexpr (LINES) store_lines4(y_out, row, d, 4 * lane + 256 * i, y4.x, y4.y, y4.z, y4.w);
            else yr[64 * i] = y4;
        }
}

int launch_embed_layernorm_groups(const RowGroups& G, const float* wte, const float* wpe, int vocab, int d,
                                  const float* w, const float* b, float eps, float* x_out, float* y_out, hipStream_t s, bool y_lines) {
    R4D_REQUIRE(d % 64 == 0 && d <= 64 * MAXV, "embed: d=%d must be a multiple of 64 and <= %d", d, 64 * MAXV);
    R4D_REQUIRE(G.n >= 1 && G.n <= ATT_MAXG, "embed: %d batches per launch (max %d)", G.n, ATT_MAXG);
    const RowTable t = make_table(G);
    const long long rows = t.row_prefix[G.n];
    if (rows <= 0) return R4D_OK;
    RowInputs in;
    for (int g = 0; g < ATT_MAXG; ++g) { in.ids[g] = g < G.n ? G.ids[g] : nullptr; in.emb[g] = g < G.n ? G.emb[g] : nullptr; }
    ProfScope prof(PK_EMBED_LN, 12.0 * rows * d + 8.0 * rows, s);   // bytes: gather row + write x, y (+ ids)
    uintptr_t align = (uintptr_t)wte | (uintptr_t)wpe | (uintptr_t)w | (uintptr_t)b | (uintptr_t)x_out | (uintptr_t)y_out;
    for (int g = 0; g < G.n; ++g) align |= (uintptr_t)in.emb[g];
    const bool vec = d % 256 == 0 && (align & 15) == 0;
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (y_lines) {                                               // y as f16x2 lines (layernorm_lines_supported(d) checked by the caller)
        R4D_REQUIRE(vec, "embed: line output needs d %% 256 == 0 and 16-byte aligned pointers");
        R4D_BRANCH(EMBED_LN4);
        if (d <= 512) hipLaunchKernelGGL((embed_ln4_groups_kernel<2, true>), grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out);
        else if (d <= 1024) hipLaunchKernelGGL((embed_ln4_groups_kernel<4, true>), grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out);
        else hipLaunchKernelGGL((embed_ln4_groups_kernel<8, true>), grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out);
        R4D_CHECK_LAUNCH("embed_layernorm");
        return R4D_OK;
    }
    if (vec && d <= 512) { R4D_BRANCH(EMBED_LN4); hipLaunchKernelGGL(embed_ln4_groups_kernel<2>, grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out); }
    else if (vec && d <= 1024) { R4D_BRANCH(EMBED_LN4); hipLaunchKernelGGL(embed_ln4_groups_kernel<4>, grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out); }
    else if (vec) { R4D_BRANCH(EMBED_LN4); hipLaunchKernelGGL(embed_ln4_groups_kernel<8>, grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out); }
    else {
        R4D_BRANCH(EMBED_GENERIC);
        hipLaunchKernelGGL(embed_ln_groups_kernel, grid, dim3(256), 0, s, t, in, wte, wpe, vocab, d, w, b, eps, x_out, y_out);
    }
    R4D_CHECK_LAUNCH("embed_layernorm");
    return R4D_OK;
}

// Decode step: one NEW position per sequence.  x[b,:] = (ids ? wte[ids[b]] : emb[b]) + wpe[pos[b]], y = LayerNorm(x).
// A position outside the cache / the position table poisons the row (NaN) instead of faulting.
__global__ __launch_bounds__(256) void embed_pos_ln_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                           const int32_t* __restrict__ pos, const float* __restrict__ wte,
                                                           const float* __restrict__ wpe, int vocab, int n_positions,
                                                           int t_cap, int B, int d, const float* __restrict__ w,
                                                           const float* __restrict__ b, float eps,
                                                           float* __restrict__ x_out, float* __restrict__ y_out,
                                                           unsigned* __restrict__ zero_words, int n_zero) {
    // first kernel of a decode step: also clears the split-K ticket counters of the step's projections (no memset node)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0u;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int nv = d >> 6;
    const int p = pos[row];
    bool bad = p < 0 || p >= n_positions || p >= t_cap;
    const float* src;
    if (ids) {
        const long long id = ids[row];
        bad = bad || id < 0 || id >= vocab;
        src = wte + (bad ? 0 : id) * (long long)d;
    } else {
        src = emb + (long long)row * d;
    }
    const float* pe = wpe + (long long)(bad ? 0 : p) * d;
    float v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
            const float e = src[lane + 64 * i] + pe[lane + 64 * i];
            v[i] = bad ? __builtin_nanf("") : e;
            x_out[(long long)row * d + lane + 64 * i] = v[i];
        }
    ln_row(v, nv, d, eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) y_out[(long long)row * d + lane + 64 * i] = v[i] * w[lane + 64 * i] + b[lane + 64 * i];
}

int launch_embed_pos_layernorm(const int64_t* ids, const float* emb, const int32_t* pos, const float* wte,
                               const float* wpe, int vocab, int n_positions, int t_cap, int B, int d, const float* w,
                               const float* b, float eps, float* x_out, float* y_out, hipStream_t s, void* zero_words,
                               size_t zero_bytes) {
    R4D_REQUIRE(d % 64 == 0 && d <= 64 * MAXV, "embed: d=%d must be a multiple of 64 and <= %d", d, 64 * MAXV);
    if (B <= 0) return R4D_OK;
    ProfScope prof(PK_EMBED_LN, 12.0 * B * d + 12.0 * B, s);
    hipLaunchKernelGGL(embed_pos_ln_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, ids, emb, pos, wte, wpe, vocab,
                       n_positions, t_cap, B, d, w, b, eps, x_out, y_out, (unsigned*)zero_words, (int)(zero_bytes / 4));
    R4D_CHECK_LAUNCH("embed_pos_layernorm");
    return R4D_OK;
}

// Decode attention: ONE query (the new position) per (sequence, head) against that sequence's cached keys / values
// -- Attention._attn with a one-row query and layer_past (modeling_gpt2.py:140-160,177-197).  HBM-bound in principle
// (every cached K and V row of the head is read once) but LATENCY-bound in practice: the first version (4 waves, one
// key per wave and load, 4 keys in flight, a score pass and a value pass) took 71 us per layer at ~100 cached
// positions -- 2 x n/16 dependent round trips to memory.  Now: a key is owned by a GROUP of GL lanes (16 bytes per lane:
// a head row of <= 128 floats is one load of 32 lanes), 512 threads = 16 groups, each with U = 8 keys AND their values
// in flight (128 positions per trip), one pass with an online softmax per group (logits DIVIDED by sqrt(hd) as the
// reference does, :143), the 16 group states merged through LDS in group order.  The new position's K / V row is
// appended to the cache by group 0 and taken from registers, not read back.  No masking: only positions <= pos exist.
template <int GL>
__global__ __launch_bounds__(512) void decode_attn_kernel(const float* __restrict__ qkv_new, float* __restrict__ kv,
                                                          const int32_t* __restrict__ pos, int t_cap, int H, int d,
                                                          float* __restrict__ out) {
    constexpr int NGRP = 512 / GL, U = 8;
    __shared__ float4 pacc[NGRP][GL];
    __shared__ float pm[NGRP], pl[NGRP];
    const int h = blockIdx.x, b = blockIdx.y;
    const int hd = d / H;
    const int tid = threadIdx.x, grp = tid / GL, gl = tid % GL;
    const bool on = 4 * gl < hd;                       // head_dim % 4 == 0: a lane owns 4 whole columns or none
    const int p = pos[b];
    if (p < 0 || p >= t_cap) {                         // embed_pos_ln_kernel poisoned this row already
        for (int c = tid; c < hd; c += 512) out[(long long)b * d + h * hd + c] = __builtin_nanf("");
        return;
    }
    typedef unsigned int u32x4d __attribute__((ext_vector_type(4)));
    typedef float f32x4d __attribute__((ext_vector_type(4)));
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float* kvb = kv + (long long)b * t_cap * 2 * d + h * hd;          // this sequence and head: rows of [K (d) | V (d)]
    const float* nq = qkv_new + (long long)b * 3 * d + h * hd;
    // lanes past the head (hd = 96: 24 of 32) repeat its last 16 bytes with a ZERO query fragment: plain loads without
    // selects (a conditional float4 load compiles to four flat dword loads through a select of addresses)
    const int glc = min(gl, hd / 4 - 1);
    float4 q4 = reinterpret_cast<const float4*>(nq)[glc];
    if (!on) q4 = z4;
    const float scale = sqrtf((float)hd);
    float m = -INFINITY, l = 0.f;
    float4 acc = z4;
    auto group_sum = [](float v) {
#pragma unroll
        for (int o = GL / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    if (grp == 0) {                                    // the new position: append to the cache, start from registers
        const float4 k4 = reinterpret_cast<const float4*>(nq + d)[glc];
        const float4 v4 = reinterpret_cast<const float4*>(nq + 2 * d)[glc];
        reinterpret_cast<float4*>(kvb + (long long)p * 2 * d)[glc] = k4;          // idle lanes rewrite the same 16 bytes
        reinterpret_cast<float4*>(kvb + (long long)p * 2 * d + d)[glc] = v4;
        m = group_sum((k4.x * q4.x + k4.y * q4.y) + (k4.z * q4.z + k4.w * q4.w)) / scale;
        l = 1.f;
        acc = v4;
    }
    // cached rows through a buffer resource: 32-bit offsets (a sequence's cache is < 2 GB), one b128 load per K / V piece
    const unsigned row_bytes = 2u * d * 4u;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(kvb, 0, (unsigned)t_cap * row_bytes - h * hd * 4u, 0x00020000);
    const unsigned v_off = (unsigned)d * 4u + 16u * glc;
    for (int t0 = 0; t0 < p; t0 += NGRP * U) {         // uniform trip count; a group's keys: t0 + u * NGRP + grp
        u32x4d kr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned ro = (unsigned)min(t0 + u * NGRP + grp, p - 1) * row_bytes;
            kr[u] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ro + 16u * glc, 0, 0);
            vr[u] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ro + v_off, 0, 0);
        }
        // the U logits of the trip first (U independent reductions over the group's lanes), ONE new running maximum, then the
        // weights: the per-key form (maximum, rescale, accumulate, U times in a row) was a chain of U dependent
        // shuffle-reduce + exp + fma steps behind the loads
        float sv[U];
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4d k4 = __builtin_bit_cast(f32x4d, kr[u]);
            const float dot = group_sum((k4.x * q4.x + k4.y * q4.y) + (k4.z * q4.z + k4.w * q4.w)) / scale;
            sv[u] = (t0 + u * NGRP + grp < p) ? dot : -INFINITY;
            mn = fmaxf(mn, sv[u]);
        }
        if (mn != -INFINITY) {                         // group-uniform: at least one key so far
            const float corr = __expf(m - mn);         // m == -inf: 0
            l *= corr; acc.x *= corr; acc.y *= corr; acc.z *= corr; acc.w *= corr;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4d v4 = __builtin_bit_cast(f32x4d, vr[u]);
                const float e = __expf(sv[u] - mn);    // keys past p: exp(-inf) == 0
                l += e;
                acc.x += e * v4.x; acc.y += e * v4.y; acc.z += e * v4.z; acc.w += e * v4.w;
            }
            m = mn;
        }
    }
    pacc[grp][gl] = acc;
    if (gl == 0) { pm[grp] = m; pl[grp] = l; }
    __syncthreads();
    if (grp == 0 && on) {                              // merge the group states in group order
        float M = pm[0];
#pragma unroll
        for (int g = 1; g < NGRP; ++g) M = fmaxf(M, pm[g]);
        float L = 0.f;
        float4 o = z4;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const float w = __expf(pm[g] - M);         // groups that saw no key: exp(-inf) == 0
            const float4 a = pacc[g][gl];
            L += pl[g] * w;
            o.x += a.x * w; o.y += a.y * w; o.z += a.z * w; o.w += a.w * w;
        }
        o.x /= L; o.y /= L; o.z /= L; o.w /= L;
        reinterpret_cast<float4*>(out + (long long)b * d + h * hd)[gl] = o;
    }
}

int launch_decode_attention(const float* qkv_new, float* kv_layer, const int32_t* pos, int B, int t_cap, int H, int d,
                            float* out, hipStream_t s) {
    const int hd = d / H;
    R4D_REQUIRE(hd * H == d && hd % 4 == 0 && hd <= 256, "decode attention: head_dim=%d (multiple of 4, max 256)", hd);
    R4D_REQUIRE(d % 4 == 0 && t_cap >= 1 && (size_t)t_cap * 2 * d * 4 < ((size_t)1 << 31),
                "decode attention: d=%d t_cap=%d (a sequence's cache must stay below 2 GB)", d, t_cap);
    if (B <= 0) return R4D_OK;
    // algorithmic bytes: the cached K and V rows of every head (upper bound t_cap/2 on average is unknown here: count
    // the new row only; callers that profile use the HIP-event time)
    ProfScope prof(PK_DECODE_ATTN, 16.0 * B * d, s);
    if (hd <= 128) R4D_BRANCH(DEC_ATT_32); else R4D_BRANCH(DEC_ATT_64);
    if (hd <= 128) hipLaunchKernelGGL(decode_attn_kernel<32>, dim3(H, B), dim3(512), 0, s, qkv_new, kv_layer, pos, t_cap, H, d, out);
    else hipLaunchKernelGGL(decode_attn_kernel<64>, dim3(H, B), dim3(512), 0, s, qkv_new, kv_layer, pos, t_cap, H, d, out);
    R4D_CHECK_LAUNCH("decode_attention");
    return R4D_OK;
}

// Greedy bookkeeping of one decode step, on the device (one workgroup per sequence): argmax of the sequence's logits
// (lowest index among equal maxima) and the stop rules of the reference's greedy loops
// (Evaluation_SimpleDyG.py:126-145, Evaluation_generator.py:153-175) -- so that a step needs no host round trip and
// the whole step can be replayed as a captured graph.
//
// With `emb.x_out` the kernel is ALSO the first kernel of the decode step that follows: it writes the un-normalised input row
// x[b,:] = wte[next] + wpe[pos] of the token it has just chosen (the id and the position are in its registers: the separate
// embedding kernel spent most of its 11 us on the two dependent loads id -> row), and clears the step's split-K ticket counters.
__global__ __launch_bounds__(256) void greedy_advance_kernel(const float* __restrict__ logits, int V, GreedyState st,
                                                             GreedyEmbed emb) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_next, s_pos;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (b == 0)
        for (int i = tid; i < emb.n_zero; i += 256) emb.zero_words[i] = 0u;
    const float* row = logits + (long long)b * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < V; j += 256) {
        const float v = row[j];
        if (v > best || bi == 0x7fffffff) { best = v; bi = j; }      // ascending j per thread: first maximum kept
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        int a = st.active[b];
        const int len = st.lens[b];
        if (a) {
            int g = st.gen_len[b];
            if (g < st.out_cap) st.out_tokens[(long long)b * st.out_cap + g] = bi;
            ++g;
            st.gen_len[b] = g;
            const int max_gen = st.params[0], len_limit = min(st.params[1], st.t_cap), n_eos = min(st.params[2], 4);
            bool stop = g >= max_gen || g >= st.out_cap || len + 1 >= len_limit;      // the cache row written next is `len`
            for (int e = 0; e < n_eos; ++e) stop |= bi == st.params[3 + e];
            if (stop) a = 0;
            st.active[b] = a;
        }
        st.next[b] = bi;
        st.pos[b] = a ? len : 0;                 // finished sequences rewrite their row 0: harmless, they are never read again
        st.lens[b] = len + a;
        s_next = bi; s_pos = a ? len : 0;
    }
    if (!emb.x_out) return;                      // (uniform per launch)
    __syncthreads();
    const int id = s_next, p = s_pos;
    const bool bad = p < 0 || p >= emb.n_positions || p >= st.t_cap || id < 0 || id >= emb.vocab;   // as embed_pos_ln_kernel: poison, never fault
    const float* src = emb.wte + (long long)(bad ? 0 : id) * emb.d;
    const float* pe = emb.wpe + (long long)(bad ? 0 : p) * emb.d;
    for (int c = tid; c < emb.d; c += 256) emb.x_out[(long long)b * emb.d + c] = bad ? __builtin_nanf("") : src[c] + pe[c];
}

int launch_greedy_advance(const float* logits, int B, int V, const GreedyState& st, hipStream_t s, const GreedyEmbed* embed) {
    R4D_REQUIRE(logits && st.next && st.lens && st.pos && st.active && st.gen_len && st.out_tokens && st.params,
                "greedy advance: null pointer");
    R4D_REQUIRE(V >= 1 && st.out_cap >= 1 && st.t_cap >= 1, "greedy advance: V=%d out_cap=%d t_cap=%d", V, st.out_cap, st.t_cap);
    if (B <= 0) return R4D_OK;
    ProfScope prof(PK_GREEDY_ADVANCE, 4.0 * B * (double)V, s);
    GreedyEmbed e;
    memset(&e, 0, sizeof(e));
    if (embed) {
        e = *embed;
        R4D_REQUIRE(e.wte && e.wpe && e.x_out && e.d >= 1, "greedy advance: bad embedding arguments");
    }
    hipLaunchKernelGGL(greedy_advance_kernel, dim3(B), dim3(256), 0, s, logits, V, st, e);
    R4D_CHECK_LAUNCH("greedy_advance");
    return R4D_OK;
}

// In-place causal softmax of the scaled scores S[z][i][0..i] (z = batch*head), one wavefront per query
// row.  The reference masks with w*b - 1e4*(1-b) (modeling_gpt2.py:146) and soft-maxes all T keys (:152):
// masked terms are exp(-1e4 - max) == 0 in fp32 whenever the row max exceeds -9896, so only keys j <= i
// are read.  Columns i+1 .. roundup(i+1, row_tile)-1 are zero-filled: the P.V GEMM's causal K-loop
// covers whole row tiles.
__global__ __launch_bounds__(256) void causal_softmax_kernel(float* __restrict__ S, int nrows_total, int T, int ld,
                                                             int row_tile) {
    const int lane = threadIdx.x & 63;
    const long long gr = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= nrows_total) return;
    const int i = (int)(gr % T);
    float* row = S + (gr / T) * (long long)T * ld + (long long)i * ld;
    const int n = i + 1;
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = lane + 64 * c;
        v[c] = (j < n) ? row[j] : -INFINITY;
        m = fmaxf(m, v[c]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        v[c] = (lane + 64 * c < n) ? expf(v[c] - m) : 0.f;
        s += v[c];
    }
    s = wave_sum(s);
    const int nfill = min(ld, (n + row_tile - 1) / row_tile * row_tile);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = lane + 64 * c;
        if (j < nfill) row[j] = v[c] / s;
    }
}

int launch_causal_softmax(float* S, int nbh, int T, int ld, int row_tile, hipStream_t s) {
    R4D_REQUIRE(T <= 1024, "attention: T=%d exceeds n_positions cap 1024", T);
    const long long rows = (long long)nbh * T;
    ProfScope prof(PK_SOFTMAX, 4.0 * nbh * T * (double)(T + 1), s);   // bytes: causal half read + written
    hipLaunchKernelGGL(causal_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, (int)rows, T, ld,
                       row_tile);
    R4D_CHECK_LAUNCH("causal_softmax");
    return R4D_OK;
}

// hidden = ln_f(x) (modeling_gpt2.py:493) and/or pool[b,:] = mean over ALL T padded positions of hidden
// (train_retriever.py:420).  Two deterministic stages: (1) grid (S chunks of T, B): each workgroup layer-norms
// its rows (one wavefront per row), sums them per wave in registers, combines its 4 waves through LDS in a
// fixed order and writes one partial row; (2) one workgroup per sequence adds the S partials in order and
// divides by T.  No atomics -> bitwise reproducible.
template <int NV>                                   // NV = ceil over the instantiation: values per lane (d <= 64*NV)
__global__ __launch_bounds__(256) void lnf_partial_kernel(const RowTable G, const float* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ b, int d,
                                                          float eps, float* __restrict__ hidden_out,
                                                          float* __restrict__ partial, unsigned* __restrict__ range_flag) {
    extern __shared__ float red[];                     // [4][d]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int chunk = blockIdx.x, seq = blockIdx.y;
    int g = 0;
    while (g + 1 < G.n && seq >= G.seq_prefix[g + 1]) ++g;
    const int T = G.T[g], S = (T + LNF_ROWS_PER_CHUNK - 1) / LNF_ROWS_PER_CHUNK;
    if (chunk >= S) return;                            // grid.x covers the longest batch
    const int bidx = seq - G.seq_prefix[g];            // sequence within its batch
    const long long row0 = G.row0_of(g, bidx);
    const int nv = d >> 6;
    const int t0 = chunk * LNF_ROWS_PER_CHUNK, t1 = min(T, t0 + LNF_ROWS_PER_CHUNK);
    constexpr int RPW = LNF_ROWS_PER_CHUNK / 4;        // rows per wave: ALL their loads are issued before the first
    float v[RPW][NV];                                  // reduction (one row at a time left the kernel latency-bound)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int t = min(t0 + wid + 4 * r, T - 1);    // clamped: always valid; rows past t1 are dropped below
#pragma unroll
        for (int i = 0; i < NV; ++i) v[r][i] = (i < nv) ? x[(row0 + t) * d + lane + 64 * i] : 0.f;
    }
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int t = t0 + wid + 4 * r;
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s_ += v[r][i];
        const float mean = wave_sum(s_) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i < nv) { const float c = v[r][i] - mean; q += c * c; }
        const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
        // range guard (r4d_set_range_flag): a non-finite residual row -- what an activation beyond the fp16 range of the f16x2
        // GEMMs / h2 words turns into -- or a variance that overflowed leaves rstd NaN or 0 (wave-uniform)
        if (range_flag && t < t1 && !(rstd > 0.f) && lane == 0) atomicOr(range_flag, R4D_RANGE_NONFINITE_HIDDEN);
        if (t < t1) {
            const long long base = (row0 + t) * d;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (i < nv) {
                    const float y = (v[r][i] - mean) * rstd * w[lane + 64 * i] + b[lane + 64 * i];
                    if (hidden_out) hidden_out[base + lane + 64 * i] = y;
                    acc[i] += y;
                }
        }
    }
    if (!partial) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (i < nv) red[wid * d + lane + 64 * i] = acc[i];
    __syncthreads();
    const long long prow = G.part_prefix[g] + (long long)bidx * S + chunk;
    for (int c = threadIdx.x; c < d; c += 256)
        partial[prow * d + c] = (red[c] + red[d + c]) + (red[2 * d + c] + red[3 * d + c]);
}

__global__ __launch_bounds__(256) void meanpool_reduce_kernel(const RowTable G, const float* __restrict__ partial, int d,
                                                              float* __restrict__ pool_out) {
    const int seq = blockIdx.x;
    int g = 0;
    while (g + 1 < G.n && seq >= G.seq_prefix[g + 1]) ++g;
    const int T = G.T[g], S = (T + LNF_ROWS_PER_CHUNK - 1) / LNF_ROWS_PER_CHUNK;
    const long long prow = G.part_prefix[g] + (long long)(seq - G.seq_prefix[g]) * S;
    for (int c = threadIdx.x; c < d; c += 256) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partial[(prow + k) * d + c];
        pool_out[(long long)seq * d + c] = s / (float)T;
    }
}

size_t lnf_meanpool_scratch_floats(int B, int T, int d) { return (size_t)B * cdiv(T, LNF_ROWS_PER_CHUNK) * d; }

unsigned* g_range_flag = nullptr;       // r4d_set_range_flag: caller-owned device word, OR-ed by lnf_partial_kernel / normalize_rows_kernel

int launch_lnf_meanpool_groups(const RowGroups& G, const float* x, const float* w, const float* b, int d, float eps,
                               float* hidden_out, float* pool_out, float* scratch, hipStream_t s) {
    R4D_REQUIRE(d % 64 == 0 && d <= 64 * MAXV, "ln_f: d=%d must be a multiple of 64 and <= %d", d, 64 * MAXV);
    R4D_REQUIRE(!pool_out || scratch, "ln_f: mean-pool needs scratch");
    R4D_REQUIRE(G.n >= 1 && G.n <= ATT_MAXG, "ln_f: %d batches per launch (max %d)", G.n, ATT_MAXG);
    const RowTable t = make_table(G);
    const int nseq = t.seq_prefix[G.n];
    const long long rows = t.row_prefix[G.n], parts = t.part_prefix[G.n];
    if (nseq <= 0 || rows <= 0) return R4D_OK;
    R4D_REQUIRE(nseq <= 65535, "ln_f: %d sequences per launch exceed the grid limit", nseq);
    int Tmax = 0;
    for (int g = 0; g < G.n; ++g) Tmax = G.T[g] > Tmax ? G.T[g] : Tmax;
    {
        ProfScope prof(PK_LNF_MEANPOOL, 4.0 * rows * d * (hidden_out ? 2 : 1) + 4.0 * parts * d, s);
        const dim3 grid(cdiv(Tmax, LNF_ROWS_PER_CHUNK), nseq);
        float* part = pool_out ? scratch : nullptr;
        if (d <= 512) R4D_BRANCH(LNF_8); else if (d <= 1024) R4D_BRANCH(LNF_16); else R4D_BRANCH(LNF_32);
        if (d <= 512) hipLaunchKernelGGL(lnf_partial_kernel<8>, grid, dim3(256), 4 * d * sizeof(float), s, t, x, w, b, d, eps, hidden_out, part, g_range_flag);
        else if (d <= 1024) hipLaunchKernelGGL(lnf_partial_kernel<16>, grid, dim3(256), 4 * d * sizeof(float), s, t, x, w, b, d, eps, hidden_out, part, g_range_flag);
        else hipLaunchKernelGGL(lnf_partial_kernel<32>, grid, dim3(256), 4 * d * sizeof(float), s, t, x, w, b, d, eps, hidden_out, part, g_range_flag);
        R4D_CHECK_LAUNCH("lnf_partial");
    }
    if (pool_out) {
        ProfScope prof(PK_MEANPOOL_REDUCE, 4.0 * parts * d + 4.0 * nseq * d, s);
        hipLaunchKernelGGL(meanpool_reduce_kernel, dim3(nseq), dim3(256), 0, s, t, scratch, d, pool_out);
        R4D_CHECK_LAUNCH("meanpool_reduce");
    }
    return R4D_OK;
}

}  // namespace r4d
