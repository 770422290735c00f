// Fused causal multi-head attention for gfx950, exact fp32 (Attention._attn + split/merge_heads,
// models/modeling_gpt2.py:140-175): softmax(Q.K^T / sqrt(hd), causal) . V without materialising the T x T scores.
//
// One workgroup = one (batch, head, 32-query tile); its 4 wavefronts SPLIT THE KEYS (wave w takes the 32-key tiles
// w, w+4, ... up to the diagonal), each keeping a private online-softmax state (running max m, sum l) and a private
// O^T accumulator, merged once at the end.  Everything per tile stays in registers:
//   * S^T = K.Q^T ("swapped" product, exact-f32 MFMA 32x32x2): the MFMA result has the QUERY on the lane and 16 keys in
//     registers, so the row max / row sum of the softmax are 15 in-register ops plus ONE cross-lane exchange with the
//     other half-wave (__shfl_xor 32) -- no LDS, no serial lane loop.  A operand: each lane streams its own K row with
//     16-byte global loads (k permuted identically on both operands); B operand: the 32 queries staged k-major in LDS
//     (stride 33: conflict-free fill and reads).
//   * logits are DIVIDED by sqrt(hd) like the reference (:143); masked keys (key > query) are skipped, which equals
//     the reference's `w*b - 1e4*(1-b)` + softmax in fp32 whenever the row max exceeds -9896.
//   * O^T += V^T.P: the probability registers ARE the B operand of the next MFMA (it sums over the accumulator's row
//     index = key, no lane movement); A operand = V rows read with coalesced 4..16-byte loads, the head columns
//     interleaved over the MFMA tiles (tile j owns columns VW*i + j).
//   * the 4 partial states are merged pairwise through LDS (re-using the query buffer), O is transposed through LDS
//     and stored as whole 128..1024-byte rows of the merged-head layout [B, T, d].
#include <math.h>
#include "common.h"

#ifndef ATT_DBG
#define ATT_DBG 0   // tuning aid (tools/kc_ablate.sh attention_fused.hip ATT_DBG n), column-split kernel: bit 0 skips the Q.K^T MFMA loop, bit 1 the P.V loop, bit 2 the two barriers of the softmax, bit 3 the K loads after the first eight, bit 4 the V loads after the first three groups
#endif

namespace r4d {

typedef float f32x16a __attribute__((ext_vector_type(16)));

constexpr int ATT_LDQ = 33;

template <int VW>
struct VLoad;
template <>
struct VLoad<1> { static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) { v[0] = p[0]; } };
template <>
struct VLoad<2> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct VLoad<3> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
};
template <>
struct VLoad<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// HD = head_dim in {32, 64, 96, 128, 256}.  VW = floats per lane per V load, NU = V loads per key pair.
template <int HD>
__global__ __launch_bounds__(256, HD >= 256 ? 1 : 2) void attn_fused_kernel(const float* __restrict__ qkv, const AttnGroups G, int d, float scale,
                                                         float* __restrict__ out) {
    constexpr int VW = HD >= 128 ? 4 : HD / 32;
    constexpr int NU = HD >= 128 ? HD / 128 : 1;
    constexpr int NCB = HD / 32;                       // 32-column blocks of O^T  (= NU * VW)
    constexpr int NSTEP = HD / 8;                      // 16-byte K loads per key row
    constexpr int GRP = (HD >= 256) ? 8 : 4, NG = NSTEP / GRP;           // K loads per ping-pong group (NSTEP in {4,8,12,16,32})
    extern __shared__ float lds[];                     // Q tile [HD][33]; later 2 merge slots; later O tile [32][HD+4]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qt = blockIdx.x, h = blockIdx.y;
    int gi = 0;
    while (gi + 1 < G.n && (int)blockIdx.z >= G.seq_prefix[gi + 1]) ++gi;
    const int T = G.T[gi];
    const int q0 = qt * 32;
    if (q0 >= T) return;                               // grid.x covers the longest batch
    const long long rowb = G.row0[gi] + (long long)((int)blockIdx.z - G.seq_prefix[gi]) * T;     // first row of the sequence
    const long long ld3 = 3LL * d;
    const float* __restrict__ base = qkv + rowb * ld3 + (long long)h * HD;
    const float* __restrict__ Kb = base + d;
    const float* __restrict__ Vb = base + 2 * d;

    // ---- stage the 32 queries k-major: Qs[k][q].  Each wave owns 8 queries; all of its global loads are issued
    // back to back (one round trip), then written with conflict-free ds_write_b32 (bank = (k + q) % 32).
    {
        constexpr int NQL = (HD + 63) / 64;
        float qv[8][NQL];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = wid * 8 + j;
            const float* src = base + (long long)min(q0 + q, T - 1) * ld3;
#pragma unroll
            for (int i = 0; i < NQL; ++i) qv[j][i] = src[min(lane + 64 * i, HD - 1)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = wid * 8 + j;
            const bool ok = q0 + q < T;
#pragma unroll
            for (int i = 0; i < NQL; ++i) {
                const int k = lane + 64 * i;
                if (k < HD) lds[k * ATT_LDQ + q] = ok ? qv[j][i] : 0.f;
            }
        }
    }
    __syncthreads();

    f32x16a O[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int qidx = q0 + li;
    const float inv_scale = 1.0f / scale;

    for (int kt = wid; kt <= qt; kt += 4) {
        const int key0 = kt * 32;
        // ---- S^T[key][q] = sum_k K[key][k] Q[q][k]
        const float4* __restrict__ krow =
            reinterpret_cast<const float4*>(Kb + (long long)min(key0 + li, T - 1) * ld3) + lh;
        f32x16a S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        // static ping-pong over groups of GRP k-steps: the 16-byte K loads AND the LDS query operands of group g+1
        // are in flight while group g feeds the MFMAs (one wave per SIMD at hd = 256: nothing else hides latency)
        float4 kb[2][GRP];
        float qb[2][GRP][4];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            kb[0][u] = krow[2 * u];
            const float* qa = lds + (8 * u + 4 * lh) * ATT_LDQ + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) qb[0][u][c] = qa[c * ATT_LDQ];
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    kb[(g + 1) & 1][u] = krow[2 * ((g + 1) * GRP + u)];
                    const float* qa = lds + (8 * ((g + 1) * GRP + u) + 4 * lh) * ATT_LDQ + li;
#pragma unroll
                    for (int c = 0; c < 4; ++c) qb[(g + 1) & 1][u][c] = qa[c * ATT_LDQ];
                }
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const float4 kv = kb[g & 1][u];
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qb[g & 1][u][0], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qb[g & 1][u][1], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qb[g & 1][u][2], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qb[g & 1][u][3], S, 0, 0, 0);
            }
        }
        // ---- online softmax: this lane = query li; its 16 registers = keys key0 + kappa(r) + 4*lh
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            // the reference DIVIDES by sqrt(hd) (:143); for a power-of-two scale (hd = 64, 256) the reciprocal
            // multiply is the same value bit for bit and saves a ~12-instruction f32 division per logit
            // multiply by the rounded reciprocal: the IEEE division is ~10 instructions per logit (head_dim 32 / 96, where
            // sqrt(hd) is not a power of two, differ from the reference's division by <= 1 ulp of the logit)
            const float s = S[r] * inv_scale;
            S[r] = (key <= qidx) ? s : -INFINITY;
            mt = fmaxf(mt, S[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = (m_new == -INFINITY) ? 0.f : __expf(S[r] - m_new);    // exp(-inf) = 0 for masked keys
            S[r] = p;
            ps += p;
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        // wave-uniform: skip the 16*NCB multiplies when no query of this wave moved its running max
        if (__any(alpha != 1.0f && m_run != -INFINITY)) {
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T[c][q] += sum_key V[key][c] P[key][q]; register r of P covers the key pair kappa(r) + 4*{0,1}
        // ring of VD key-pair loads in flight (8 MFMAs = 512 cycles per pair do not cover an L2 round trip alone)
        constexpr int VD = (HD >= 256) ? 6 : 4;
        float vr[VD][NU][4];
#pragma unroll
        for (int r = 0; r < VD - 1; ++r) {
            const int kn = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float* vrow = Vb + (long long)min(kn, T - 1) * ld3 + VW * li;
#pragma unroll
            for (int u = 0; u < NU; ++u) VLoad<VW>::ld(vrow + 128 * u, vr[r][u]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r + VD - 1 < 16) {
                const int rn = r + VD - 1;
                const int kn = key0 + (rn & 3) + 8 * (rn >> 2) + 4 * lh;
                const float* vrow = Vb + (long long)min(kn, T - 1) * ld3 + VW * li;
#pragma unroll
                for (int u = 0; u < NU; ++u) VLoad<VW>::ld(vrow + 128 * u, vr[rn % VD][u]);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int j = 0; j < VW; ++j)
                    O[u * VW + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[r % VD][u][j], S[r], O[u * VW + j], 0, 0, 0);
        }
    }

    // ---- merge the 4 partial states: (2,3) -> (0,1), then 1 -> 0.  Slot = NCB*16*64 O values + 64 m + 64 l floats.
    constexpr int SLOT = NCB * 16 * 64 + 128;
    __syncthreads();                                    // every wave is done with the query buffer
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
        const int writers_lo = step == 0 ? 2 : 1;       // waves [writers_lo, 2*writers_lo) write, [0, writers_lo) merge
        if (wid >= writers_lo && wid < 2 * writers_lo) {
            float* sl = lds + (wid - writers_lo) * SLOT;
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sl[(c * 16 + r) * 64 + lane] = O[c][r];
            sl[NCB * 16 * 64 + lane] = m_run;
            sl[NCB * 16 * 64 + 64 + lane] = l_run;
        }
        __syncthreads();
        if (wid < writers_lo) {
            const float* sl = lds + wid * SLOT;
            const float m_b = sl[NCB * 16 * 64 + lane], l_b = sl[NCB * 16 * 64 + 64 + lane];
            const float m_new = fmaxf(m_run, m_b);
            const float fa = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
            const float fb = (m_b == -INFINITY) ? 0.f : expf(m_b - m_new);
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[c][r] = O[c][r] * fa + sl[(c * 16 + r) * 64 + lane] * fb;
            l_run = l_run * fa + l_b * fb;
            m_run = m_new;
        }
        __syncthreads();
    }
    // ---- wave 0: normalise, transpose O^T -> Ot[q][c] in LDS; all waves: coalesced row stores
    constexpr int LDO = HD + 1;                        // odd stride: conflict-free transposed writes
    if (wid == 0) {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = 128 * u + VW * ((r & 3) + 8 * (r >> 2) + 4 * lh) + j;
                    lds[li * LDO + c] = O[u * VW + j][r] * inv;
                }
    }
    __syncthreads();
    for (int q = wid; q < 32; q += 4) {
        if (q0 + q >= T) break;
        float* dst = out + (rowb + q0 + q) * d + (long long)h * HD;
        for (int c = lane * 4; c < HD; c += 256) {
            const float* sp = &lds[q * LDO + c];
            *reinterpret_cast<float4*>(dst + c) = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Column-split variant for wide heads (head_dim 128 / 256).  The key-split kernel above needs HD/2 accumulator
// registers per lane for O^T, which at head_dim 256 leaves one wave per SIMD.  Here the 4 wavefronts of a
// workgroup walk 128-key super-tiles TOGETHER: wave w computes S^T for its own 32-key sub-tile (full head_dim),
// the row max / row sum are combined across the 4 waves through 1 KB of LDS, every wave publishes its
// probabilities to LDS, and wave w accumulates only ITS quarter of the head columns of O^T over all 128 keys.
// O^T is HD/8 registers per lane (32 at hd 256), so three workgroups fit per CU, the waves are always balanced and
// there is no merge phase.
//
// Instruction diet (plain VALU / LDS / VMEM instructions do not hide under v_mfma on gfx950, tools/mfma_peak.hip):
//   * Q is staged ROW-major [query][HD+4], unscaled: the logits come out of the MFMA as raw dot products and the softmax is
//     exp2((s - max) * log2(e)/sqrt(hd)) (subtract first: exact for nearby logits), and a lane's Q operand for FOUR MFMAs is one
//     conflict-free ds_read_b128 -- matching the four k of the lane's 16-byte K load;
//   * K and V rows come in through buffer loads: lane offset loop-invariant, the key row in the SCALAR offset, rows
//     past the sequence answered with zeros by the range check -- no clamps, no 64-bit address arithmetic;
//   * the causal mask is applied only on super-tiles that touch the diagonal (wave-uniform test);
//   * P is published [query][key] with four ds_write_b128 per wave and read back as one ds_read_b128 per EIGHT
//     P.V MFMAs (keys permuted identically on the V loads).
typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2a __attribute__((ext_vector_type(2)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef float f32x2a __attribute__((ext_vector_type(2)));

template <int VW>
struct VBuf;                                            // one V fragment: VW consecutive head columns of one key row
template <>
struct VBuf<1> {
    float v[1];
    __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
};
template <>
struct VBuf<2> {
    float v[2];
    __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        // (bit_cast the WHOLE vector: __builtin_bit_cast(float, t.y) on an ext-vector element reads element 0)
        const f32x2a t = __builtin_bit_cast(f32x2a, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        v[0] = t.x; v[1] = t.y;
    }
};

template <int HD>
__global__ __launch_bounds__(256, 3) void attn_colsplit_kernel(const float* __restrict__ qkv, const AttnGroups G, int d, int H, int ntq,
                                                               float qscale, float* __restrict__ out) {
    constexpr int CW = HD >= 128 ? HD / 4 : 32;        // head columns owned by one wave (hd 96: three waves own 32 each,
    constexpr int NPV = HD / CW;                       //   the fourth only takes part in Q.K^T and the softmax)
    constexpr int VW = CW / 32;                        // floats per lane per V load = O^T tiles per wave (1 or 2)
    constexpr int NSTEP = HD / 8;                      // 16-byte K loads (and Q reads) per key row
    constexpr int LDQ = HD + 4, LDP = 132;             // row strides (floats): 16-lane b128 groups hit 16 distinct slots
    constexpr int KD = NSTEP < 8 ? NSTEP : 8;          // K loads in flight per lane
    constexpr int VD = 3;                              // V key-groups (8 keys) in flight
    extern __shared__ float lds[];
    float* Qs = lds;                                   // [32][LDQ]  Q * log2(e)/sqrt(hd)
    float* Ps = lds + 32 * LDQ;                        // [32][LDP]  probabilities of the current super-tile, [query][key]
    float* red = Ps + 32 * LDP;                        // [2][4][32] per-wave row max / row sum
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // Work mapping (1-D grid): all query tiles of one (sequence, head) run on the SAME XCD (workgroup ids are dealt
    // round-robin to the 8 XCDs, each with its own L2), back to back, so K and V of that head are fetched from HBM once
    // instead of once per query tile (measured before: 970 MB per launch against 290 MB of qkv + out, L2 hit rate 22 %,
    // i.e. the kernel ran at the HBM limit).  Long (late) query tiles go first.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / ntq) * 8 + xcd;           // (sequence, head) index
    if (pair >= G.seq_prefix[G.n] * H) return;
    const int qt = ntq - 1 - slot % ntq, h = pair % H, seq = pair / H;
    int gi = 0;
    while (gi + 1 < G.n && seq >= G.seq_prefix[gi + 1]) ++gi;
    const int T = G.T[gi];
    const int q0 = qt * 32;
    if (q0 >= T) return;                               // ntq covers the longest batch
    const long long rowb = G.row0[gi] + (long long)(seq - G.seq_prefix[gi]) * T;     // first row of the sequence
    const int ld3 = 3 * d;
    const float* __restrict__ base = qkv + rowb * ld3 + (long long)h * HD;
    const int seq_bytes = ((T - 1) * ld3 + HD) * 4;    // one head's K (or V) rows of this sequence, as a byte range
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + d), 0, seq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + 2 * d), 0, seq_bytes, 0x00020000);
    const int qidx = q0 + li;
    const int key_limit = min(T, q0 + 32);             // keys >= key_limit are masked for every query of the tile
    const int k_voff = ((32 * wid + li) * ld3 + 4 * lh) * 4;                 // bytes; + scalar (st0 * ld3 + 8u) * 4
    const int v_voff = (4 * lh * ld3 + wid * CW + VW * li) * 4;             // bytes; + scalar (st0 + 8s + c) * ld3 * 4
    // K fragments of the NEXT super-tile are requested before the P.V phase of the current one (and the first ones
    // here, before the Q tile is staged), V fragments before the softmax: no phase starts on an exposed memory latency
    u32x4a kb[KD];
    if (wid * 32 < key_limit) {
#pragma unroll
        for (int u = 0; u < KD; ++u) kb[u] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_voff, 32 * u, 0);
    }
    {   // Q tile: thread t stages row t/8, 16-byte pieces 4*(t%8) + 32j; rows past T repeat the last row (never stored)
        const int row = tid >> 3, seg = 4 * (tid & 7);
        const float* src = base + (long long)min(q0 + row, T - 1) * ld3 + seg;
        float4 qv[HD / 32];
#pragma unroll
        for (int j = 0; j < HD / 32; ++j) qv[j] = *reinterpret_cast<const float4*>(src + 32 * j);
#pragma unroll
        for (int j = 0; j < HD / 32; ++j)
            *reinterpret_cast<float4*>(Qs + row * LDQ + seg + 32 * j) =
                qv[j];        // UNSCALED (round 5): the logits leave the MFMA as raw dot products, see the softmax below
    }
    __syncthreads();

    f32x16a O[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[c][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float* q_frag = Qs + li * LDQ + 4 * lh;
    const float* p_frag = Ps + li * LDP + 4 * lh;

    for (int st0 = 0; st0 < key_limit; st0 += 128) {
        const int key0 = st0 + wid * 32;
        const bool active = key0 < key_limit;           // wave-uniform
        f32x16a S;
        float mt = -INFINITY;
        if (active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const int k_soff = st0 * ld3 * 4;
#pragma unroll
            for (int u = 0; u < ((ATT_DBG & 1) ? 1 : NSTEP); ++u) {
                const f32x4a kv = __builtin_bit_cast(f32x4a, kb[u % KD]);
                const float4 qf = *reinterpret_cast<const float4*>(q_frag + 8 * u);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qf.x, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qf.y, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qf.z, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qf.w, S, 0, 0, 0);
                if (!(ATT_DBG & 8) && u + KD < NSTEP)  // refill AFTER the slot's MFMAs in program order: no register copies
                    kb[u % KD] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_voff, k_soff + 32 * (u + KD), 0);
            }
            if (key0 + 31 > q0) {                      // the sub-tile touches the diagonal (or runs past T): mask
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    S[r] = (key <= qidx) ? S[r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, S[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        }
        const int ngroups = (min(128, key_limit - st0) + 7) >> 3;      // 8-key groups of the P.V phase, wave-uniform
        const int v_soff = st0 * ld3 * 4;
        VBuf<VW> vb[VD][4];
        if (wid < NPV) {
#pragma unroll
            for (int u = 0; u < VD; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) vb[u][c].ld(v_rsrc, v_voff, v_soff + (8 * u + c) * ld3 * 4);
        }
        if (lh == 0) red[wid * 32 + li] = mt;
        if (!(ATT_DBG & 4)) __syncthreads();
        const float m_tile = fmaxf(fmaxf(red[li], red[32 + li]), fmaxf(red[64 + li], red[96 + li]));
        const float m_new = fmaxf(m_run, m_tile);      // finite from the first super-tile on (key 0 is never masked)
        // softmax on the RAW dot products: p = exp2((s - max) * log2(e)/sqrt(hd)).  The difference of two nearby logits is exact
        // (Sterbenz) and the scale then rounds a SMALL number; scaling first (q pre-multiplied, rounds 1-4) rounded every logit
        // at its own magnitude -- 2^-24 x 3,900 = 2e-4 in the exponent at logits of several hundred (G2 attn_hd256_T40_s30:
        // 2.8e-4 from the reference against 5.5e-5 of the reference from float64).  Same instruction count per score.
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * qscale);
        float ps = 0.f;
        if (active) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 p;
                p.x = __builtin_amdgcn_exp2f((S[4 * g + 0] - m_new) * qscale);
                p.y = __builtin_amdgcn_exp2f((S[4 * g + 1] - m_new) * qscale);
                p.z = __builtin_amdgcn_exp2f((S[4 * g + 2] - m_new) * qscale);
                p.w = __builtin_amdgcn_exp2f((S[4 * g + 3] - m_new) * qscale);
                ps += (p.x + p.y) + (p.z + p.w);
                *reinterpret_cast<float4*>(Ps + li * LDP + wid * 32 + 8 * g + 4 * lh) = p;   // keys 8g + 4lh + 0..3
            }
            ps += __shfl_xor(ps, 32, 64);
        }
        if (lh == 0) red[128 + wid * 32 + li] = ps;
        if (!(ATT_DBG & 4)) __syncthreads();
        l_run = l_run * alpha + ((red[128 + li] + red[160 + li]) + (red[192 + li] + red[224 + li]));
        if (__any(alpha != 1.0f && m_run != -INFINITY)) {
#pragma unroll
            for (int c = 0; c < VW; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[c][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T[c][q] += sum over the valid keys of this super-tile of V[key][c] * P[q][key], 8 keys per group:
        // lane half lh handles keys 8s + 4lh + c (c = 0..3) -- the four components of its ds_read_b128 of P
        if (!(ATT_DBG & 8) && st0 + 128 + wid * 32 < key_limit) {         // this wave's K rows of the next super-tile
            const int kn_soff = (st0 + 128) * ld3 * 4;
#pragma unroll
            for (int u = 0; u < KD; ++u) kb[u] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_voff, kn_soff + 32 * u, 0);
        }
        for (int s0 = 0; s0 < ((ATT_DBG & 2) ? min(ngroups, 1) : (wid < NPV ? ngroups : 0)); s0 += VD) {
#pragma unroll
            for (int u = 0; u < VD; ++u) {
                const int sg = s0 + u;
                if (sg < ngroups) {
                    const float4 pf = *reinterpret_cast<const float4*>(p_frag + 8 * sg);
                    // the MFMAs read the prefetch registers DIRECTLY and the refill of the slot follows them in program
                    // order: copying the slot first ("vc = vb[u]; vb[u].ld()") cost 69 v_mov per 24 MFMAs -- every one
                    // of them taken from the matrix pipe's issue slots (rocprofv3: 1.74 VALU per MFMA in this kernel)
#pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        O[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[u][0].v[j], pf.x, O[j], 0, 0, 0);
                        O[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[u][1].v[j], pf.y, O[j], 0, 0, 0);
                        O[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[u][2].v[j], pf.z, O[j], 0, 0, 0);
                        O[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[u][3].v[j], pf.w, O[j], 0, 0, 0);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (!(ATT_DBG & 16)) vb[u][c].ld(v_rsrc, v_voff, v_soff + (8 * (sg + VD) + c) * ld3 * 4);   // past the range: zeros
                }
            }
        }
    }
    // ---- normalise, transpose through LDS (re-using the Q / P buffers), coalesced row stores
    __syncthreads();
    constexpr int LDO = HD + 1;
    {
        const float inv = 1.0f / l_run;
        if (wid < NPV) {
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = wid * CW + VW * ((r & 3) + 8 * (r >> 2) + 4 * lh) + j;
                    lds[li * LDO + c] = O[j][r] * inv;
                }
        }
    }
    __syncthreads();
    for (int q = wid; q < 32; q += 4) {
        if (q0 + q >= T) break;
        float* dst = out + (rowb + q0 + q) * d + (long long)h * HD;
        for (int c = lane * 4; c < HD; c += 256) {
            const float* sp = &lds[q * LDO + c];
            *reinterpret_cast<float4*>(dst + c) = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
    }
}

template <int HD>
static int launch_colsplit(const float* qkv, const AttnGroups& G, int Tmax, double flop, int H, int d, float* out,
                           hipStream_t s) {
    const size_t lds = ((size_t)32 * (HD + 4) + 32 * 132 + 256) * 4;       // >= the [32][HD+1] output tile
    ProfScope prof(PK_ATTN_FUSED, flop, s);
    const int ntq = cdiv(Tmax, 32);
    const long long pairs8 = ((long long)G.seq_prefix[G.n] * H + 7) / 8;
    R4D_REQUIRE(pairs8 * 8 * ntq < (1ll << 31), "attention: grid too large");
    hipLaunchKernelGGL((attn_colsplit_kernel<HD>), dim3((unsigned)(pairs8 * 8 * ntq)), dim3(256), lds, s, qkv, G, d, H, ntq,
                       (float)(1.4426950408889634 / sqrt((double)HD)), out);
    R4D_CHECK_LAUNCH("attn_colsplit");
    return R4D_OK;
}

template <int HD>
static int launch_hd(const float* qkv, const AttnGroups& G, int Tmax, double flop, int H, int d, float* out,
                     hipStream_t s) {
    constexpr int NCB = HD / 32;
    const size_t q_bytes = (size_t)HD * ATT_LDQ * 4, slot_bytes = 2 * ((size_t)NCB * 16 * 64 + 128) * 4,
                 o_bytes = (size_t)32 * (HD + 1) * 4;
    size_t lds = q_bytes > slot_bytes ? q_bytes : slot_bytes;
    if (o_bytes > lds) lds = o_bytes;
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute((const void*)attn_fused_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                set_error("attention: cannot raise dynamic LDS limit");
                return R4D_ERR_HIP;
            }
            raised = true;
        }
    }
    // algorithmic flop: causal half of Q.K^T and P.V = 2 * T^2 * hd per head
    ProfScope prof(PK_ATTN_FUSED, flop, s);
    hipLaunchKernelGGL((attn_fused_kernel<HD>), dim3(cdiv(Tmax, 32), H, G.seq_prefix[G.n]), dim3(256), lds, s, qkv, G, d,
                       (float)sqrt((double)HD), out);
    R4D_CHECK_LAUNCH("attn_fused");
    return R4D_OK;
}

int g_attention_variant = 0;          // tuning aid: 1 forces the key-split kernel at head_dim 128 / 256

// Attention of n <= ATT_MAXG batches in one launch; batch g holds Bs[g] sequences of Ts[g] tokens starting at token
// row row0s[g] of qkv [rows, 3d] / out [rows, d].  Returns R4D_OK, an error, or +1 when head_dim has no fused
// instantiation (caller falls back to the three-launch form).
int launch_attention_fused_groups(const float* qkv, int n, const int* Bs, const int* Ts, const long long* row0s, int H,
                                  int d, float* out, hipStream_t s) {
    const int hd = d / H;
    R4D_REQUIRE(n >= 1 && n <= ATT_MAXG, "attention: %d batches per launch (max %d)", n, ATT_MAXG);
    R4D_REQUIRE(H <= 65535, "attention: H=%d exceeds the grid limit", H);
    AttnGroups G;
    G.n = n;
    G.seq_prefix[0] = 0;
    int Tmax = 0;
    double flop = 0.0;                                  // causal half of Q.K^T and P.V: 2 * T^2 * hd per head
    for (int g = 0; g < n; ++g) {
        G.seq_prefix[g + 1] = G.seq_prefix[g] + Bs[g];
        G.T[g] = Ts[g];
        G.row0[g] = row0s[g];
        if (Ts[g] > Tmax) Tmax = Ts[g];
        flop += 2.0 * Bs[g] * H * (double)Ts[g] * Ts[g] * hd;
    }
    for (int g = n; g < ATT_MAXG; ++g) { G.seq_prefix[g + 1] = G.seq_prefix[n]; G.T[g] = 0; G.row0[g] = 0; }
    R4D_REQUIRE(G.seq_prefix[n] <= 65535, "attention: %d sequences per launch exceed the grid limit", G.seq_prefix[n]);
    switch (hd) {
        case 32: R4D_BRANCH(ATT_KS32); break;
        case 64: R4D_BRANCH(ATT_KS64); break;
        case 96: if (g_attention_variant == 1) R4D_BRANCH(ATT_KS_FORCED); else R4D_BRANCH(ATT_CS96); break;
        case 128: if (g_attention_variant == 1) R4D_BRANCH(ATT_KS_FORCED); else R4D_BRANCH(ATT_CS128); break;
        case 256: if (g_attention_variant == 1) R4D_BRANCH(ATT_KS_FORCED); else R4D_BRANCH(ATT_CS256); break;
        default: break;
    }
    switch (hd) {
        case 32: return launch_hd<32>(qkv, G, Tmax, flop, H, d, out, s);
        case 64: return launch_hd<64>(qkv, G, Tmax, flop, H, d, out, s);
        case 96: return g_attention_variant == 1 ? launch_hd<96>(qkv, G, Tmax, flop, H, d, out, s)
                                                 : launch_colsplit<96>(qkv, G, Tmax, flop, H, d, out, s);
        case 128: return g_attention_variant == 1 ? launch_hd<128>(qkv, G, Tmax, flop, H, d, out, s)
                                                  : launch_colsplit<128>(qkv, G, Tmax, flop, H, d, out, s);
        case 256: return g_attention_variant == 1 ? launch_hd<256>(qkv, G, Tmax, flop, H, d, out, s)
                                                  : launch_colsplit<256>(qkv, G, Tmax, flop, H, d, out, s);
        default: return 1;
    }
}

int launch_attention_fused(const float* qkv, int B, int T, int H, int d, float* out, hipStream_t s) {
    const long long row0 = 0;
    return launch_attention_fused_groups(qkv, 1, &B, &T, &row0, H, d, out, s);
}

}  // namespace r4d

namespace r4d { int dbgflag_att() { return ATT_DBG != 0; } }
