// Device ranking on gfx950 with the canonical order of the build (value descending, index ascending ==
// np.argsort(-x, kind='stable')): per-row top-k in ONE launch, shard merge, and the full stable permutation.
// Reference: train/train_retriever.py:357-358,461-467 and retrieval_data_annotation.py:88-103 (argsort / top-10).
//
// Selection is a latency problem, not a bandwidth one (32 rows x 100k scores are 12.8 MB in the Infinity Cache; the
// first version spent 14 us per launch in ds_bpermute butterflies).  Hence:
//   * values become order-preserving unsigned keys once (NaN -> -inf, -0.0 -> +0.0), so a wave-wide maximum is four
//     DPP v_max steps inside the 16-lane rows + four v_readlane;
//   * a lane owns CONSECUTIVE columns, so among equal keys the winner is simply the lowest lane of a ballot (s_ff1) and,
//     inside the lane, the lowest slot: no second reduction over positions;
//   * a workgroup of up to 16 wavefronts reduces 16 x 1024 columns to k through LDS in the same launch, and when a row
//     spans several workgroups the LAST one to arrive (device-scope counter) merges their candidates: one launch for
//     rows of up to 16384 * 1024 / k columns.
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "common.h"

namespace r4d {

// ------------------------------------------------------------------------------------ order-preserving keys
template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    typedef uint32_t K;
    static __device__ __forceinline__ K make(float x) {           // larger value <=> larger key; NaN sorts with -inf
        x = (x == x) ? x + 0.0f : -INFINITY;                       // -0.0 + 0.0 == +0.0
        const uint32_t b = __builtin_bit_cast(uint32_t, x);
        return b ^ (uint32_t)(((int32_t)b >> 31) | (int32_t)0x80000000);
    }
    static __device__ __forceinline__ float value(K k) {
        const uint32_t b = k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu);
        return __builtin_bit_cast(float, b);
    }
};
template <> struct KeyOf<double> {
    typedef uint64_t K;
    static __device__ __forceinline__ K make(double x) {
        x = (x == x) ? x + 0.0 : -(double)INFINITY;
        const uint64_t b = __builtin_bit_cast(uint64_t, x);
        return b ^ (uint64_t)(((int64_t)b >> 63) | (int64_t)0x8000000000000000ull);
    }
    static __device__ __forceinline__ double value(K k) {
        const uint64_t b = k ^ ((k >> 63) ? 0x8000000000000000ull : 0xffffffffffffffffull);
        return __builtin_bit_cast(double, b);
    }
};
// Key 0 is below every real key (the smallest real key is that of -inf): it marks padding and consumed slots.

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t readlane_u32(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }

// maximum over the 64 lanes, returned to every lane (wave-uniform)
__device__ __forceinline__ uint32_t wave_max_key(uint32_t x) {
    x = max(x, dpp_u32<0xB1>(x));      // quad_perm [1,0,3,2]
    x = max(x, dpp_u32<0x4E>(x));      // quad_perm [2,3,0,1]
    x = max(x, dpp_u32<0x141>(x));     // row_half_mirror
    x = max(x, dpp_u32<0x140>(x));     // row_mirror: every 16-lane row is uniform now
    return max(max(readlane_u32(x, 0), readlane_u32(x, 16)), max(readlane_u32(x, 32), readlane_u32(x, 48)));
}
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_u64(uint64_t x) {
    return ((uint64_t)dpp_u32<CTRL>((uint32_t)(x >> 32)) << 32) | dpp_u32<CTRL>((uint32_t)x);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, int l) {
    return ((uint64_t)readlane_u32((uint32_t)(x >> 32), l) << 32) | readlane_u32((uint32_t)x, l);
}
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t wave_max_key(uint64_t x) {
    x = umax64(x, dpp_u64<0xB1>(x));
    x = umax64(x, dpp_u64<0x4E>(x));
    x = umax64(x, dpp_u64<0x141>(x));
    x = umax64(x, dpp_u64<0x140>(x));
    return umax64(umax64(readlane_u64(x, 0), readlane_u64(x, 16)), umax64(readlane_u64(x, 32), readlane_u64(x, 48)));
}

// ------------------------------------------------------------------------------------ wavefront selection
// Iterative form: k rounds of (wave maximum, lowest lane of the ballot, that lane consumes its slot and rescans).  Lane
// L holds keys c[0..E) at list positions base + e with base = L * E + const: positions ascend with (lane, slot), so ties
// resolve to the lowest lane, then the lowest slot.  Lane r < k returns the r-th winner (key 0 = none).  ~115 vector
// instructions per round at E = 16: kept for k > 16 and as the fallback of the filter form below.
template <typename K, int E>
__device__ __forceinline__ void wave_select_iter(K (&c)[E], unsigned base, int k, int lane, K& won_k, unsigned& won_p) {
    K bk = 0;
    int bs = 0;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {                 // lowest slot among equal keys
        const bool ge = c[e] >= bk;
        bk = ge ? c[e] : bk; bs = ge ? e : bs;
    }
    won_k = 0; won_p = 0xffffffffu;
    for (int r = 0; r < k; ++r) {
        const K M = wave_max_key(bk);
        if (M == 0) break;                             // wave-uniform: nothing left but padding
        const unsigned long long mask = __ballot(bk == M);
        const int wl = __builtin_ctzll(mask);
        const unsigned wp = readlane_u32(base + (unsigned)bs, wl);
        if (lane == r) { won_k = M; won_p = wp; }
        if (lane == wl) {                              // the winner consumes its slot and rescans its E registers
            K nb = 0;
            int ns = 0;
#pragma unroll
            for (int e = E - 1; e >= 0; --e) {
                c[e] = (e == bs) ? (K)0 : c[e];
                const bool ge = c[e] >= nb;
                nb = ge ? c[e] : nb; ns = ge ? e : ns;
            }
            bk = nb; bs = ns;
        }
    }
}

__device__ __forceinline__ uint32_t readlane_key(uint32_t x, int l) { return readlane_u32(x, l); }
__device__ __forceinline__ uint64_t readlane_key(uint64_t x, int l) { return readlane_u64(x, l); }

// Rank of every lane's item among the first m <= 64 lanes' items under (key desc, position asc): m rounds of two
// v_readlane + three compares, no memory, no cross-lane shuffles.
template <typename K>
__device__ __forceinline__ int wave_rank(K key, unsigned pos, int m) {
    int rank = 0;
    for (int j = 0; j < m; ++j) {                      // m is wave-uniform
        const K kj = readlane_key(key, j);
        const unsigned pj = readlane_u32(pos, j);
        rank += (int)((kj > key) | ((kj == key) & (pj < pos)));
    }
    return rank;
}

// Filter form (k <= 16): the k-th largest of the 64 LANE maxima is a lower bound T of the k-th best key (k lanes hold a
// key >= T), so only keys >= T can win -- typically k plus a handful.  T comes out of a bitwise descent with one
// v_cmp + s_bcnt1 per bit (everything else scalar); the survivors are compacted through a 64-entry LDS list (ballot +
// mbcnt prefix) and ranked exactly by wave_rank.  ~330 instructions per 1024 candidates, independent of k, against
// ~115 per ROUND for the iterative form.  More than 64 survivors (ties en masse, fewer than k valid lanes): iterative
// form on the untouched registers.
// Result, "scatter" form: lane i reports at most one winner -- (rank, key, position); `has` false otherwise.
template <typename K, int E>
__device__ __forceinline__ void wave_topk(K (&c)[E], unsigned base, int k, int lane, K* __restrict__ l_key,
                                          unsigned* __restrict__ l_pos, bool& has, int& rank, K& key, unsigned& pos) {
    constexpr int BITS = sizeof(K) * 8;
    has = false; rank = 0; key = 0; pos = 0xffffffffu;
    bool done = false;
    if (k <= 16) {
        K lm = c[0];
#pragma unroll
        for (int e = 1; e < E; ++e) lm = c[e] > lm ? c[e] : lm;
        K t = 0;
        for (int b = BITS - 1; b >= 0; --b) {          // largest t with #{lanes: lm >= t} >= k  (all scalar but the compare)
            const K cand = t | ((K)1 << b);
            if (__builtin_popcountll(__ballot(lm >= cand)) >= k) t = cand;
        }
        int m = 0;
        if (t != 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool pass = c[e] >= t;
                const unsigned long long mask = __ballot(pass);
                const int at = m + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (pass && at < 64) { l_key[at] = c[e]; l_pos[at] = base + (unsigned)e; }
                m += __builtin_popcountll(mask);
            }
        }
        if (t != 0 && m <= 64) {                       // wave-uniform
            if (lane < m) { key = l_key[lane]; pos = l_pos[lane]; }
            rank = wave_rank<K>(key, pos, m);
            has = lane < m && rank < k;
            done = true;
        }
    }
    if (!done) {
        K wk;
        unsigned wp;
        wave_select_iter<K, E>(c, base, k, lane, wk, wp);
        has = lane < k && wk != 0; rank = lane; key = wk; pos = wp;
    }
}

constexpr int SEG_E = 16;                  // columns per lane
constexpr int SEG = 64 * SEG_E;            // columns per wavefront segment
constexpr int CHUNK_WAVES = 16;            // most wavefronts (segments) per workgroup
constexpr int MAXK = 64;

template <typename T> struct Vec4;         // 16-byte-or-wider pieces of a row, alignment of ONE element only: rows start
template <> struct Vec4<float> {           // at row * ld with any ld (global loads tolerate dword alignment)
    typedef float V __attribute__((ext_vector_type(4), aligned(4)));
};
template <> struct Vec4<double> {
    typedef double V __attribute__((ext_vector_type(4), aligned(8)));
};

// keys of the SEG_E consecutive columns [col0, col0 + SEG_E) of one row; columns >= n are padding (key 0)
template <typename T>
__device__ __forceinline__ void load_keys(const T* __restrict__ v, int n, int col0, typename KeyOf<T>::K (&c)[SEG_E]) {
    typedef typename Vec4<T>::V V4;
#pragma unroll
    for (int j = 0; j < SEG_E / 4; ++j) {
        const int c0 = col0 + 4 * j;
        if (c0 + 3 < n) {
            const V4 x = *reinterpret_cast<const V4*>(v + c0);
            c[4 * j] = KeyOf<T>::make(x.x); c[4 * j + 1] = KeyOf<T>::make(x.y);
            c[4 * j + 2] = KeyOf<T>::make(x.z); c[4 * j + 3] = KeyOf<T>::make(x.w);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // clamped address, value masked: no pointer select
                const typename KeyOf<T>::K kk = KeyOf<T>::make(v[min(max(c0 + e, 0), n - 1)]);
                c[4 * j + e] = (c0 + e < n) ? kk : 0;
            }
        }
    }
}

// the k best of a list of m <= 1024 keys in LDS (list position = payload handle), by ONE wavefront
template <typename K>
__device__ __forceinline__ void topk_of_list(const K* __restrict__ keys, int m, int k, int lane, K* l_key, unsigned* l_pos,
                                             bool& has, int& rank, K& key, unsigned& pos) {
    if (m <= 64) {                                     // one item per lane: rank directly
        key = lane < m ? keys[lane] : (K)0;
        pos = (unsigned)lane;
        rank = wave_rank<K>(key, pos, m);
        has = lane < m && rank < k && key != 0;
    } else if (m <= 256) {
        K c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = (lane * 4 + e < m) ? keys[lane * 4 + e] : (K)0;
        wave_topk<K, 4>(c, (unsigned)lane * 4, k, lane, l_key, l_pos, has, rank, key, pos);
    } else {
        K c[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = (lane * 16 + e < m) ? keys[lane * 16 + e] : (K)0;
        wave_topk<K, 16>(c, (unsigned)lane * 16, k, lane, l_key, l_pos, has, rank, key, pos);
    }
}

// write-through (sc1) publication of a candidate to the other workgroups of the launch: relaxed agent-scope atomics
// compile to global_store / global_load ... sc1, which need no release fence (a release would write back the whole L2)
__device__ __forceinline__ void publish(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish(long long* p, long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float consume(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double consume(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ long long consume(const long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid (nchunks, rows); block = 64 * nw wavefronts, nw in {1..4, 8, 16}.  Workgroup (c, row) reduces columns
// [c * nw * 1024, (c+1) * nw * 1024) of the row to k candidates: every wavefront its 1024 columns, wave 0 the nw * k
// survivors from LDS.  With one chunk they ARE the result; otherwise they are published write-through to
// cand_* [rows, nchunks * k] and -- when `finish` -- the last workgroup of the row to draw a ticket selects among them.
//   vals [rows, ld]; idx_in (nullable) parallel global indices (deeper levels / merges), else index = column + offset.
template <typename T>
__global__ __launch_bounds__(64 * CHUNK_WAVES) void topk_chunk_kernel(
    const T* __restrict__ vals, const long long* __restrict__ idx_in, int n, long long ld, int k, long long index_offset,
    int nchunks, T* cand_v, long long* cand_i, T* __restrict__ out_v,
    long long* __restrict__ out_i, unsigned* counters, int finish) {
    typedef typename KeyOf<T>::K K;
    __shared__ K s_key[CHUNK_WAVES * MAXK];            // level-1 survivors (wave, rank); later the row's candidates
    __shared__ unsigned s_pos[CHUNK_WAVES * MAXK];
    __shared__ K l_key[CHUNK_WAVES][64];               // per-wave compaction lists of the filter form
    __shared__ unsigned l_pos[CHUNK_WAVES][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int chunk = blockIdx.x, row = blockIdx.y;
    const T* v = vals + (long long)row * ld;
    bool has;
    int rank;
    K key;
    unsigned pos;
    {
        K c[SEG_E];
        const int col0 = (chunk * nw + w) * SEG + lane * SEG_E;
        load_keys<T>(v, n, col0, c);
        wave_topk<K, SEG_E>(c, (unsigned)col0, k, lane, l_key[w], l_pos[w], has, rank, key, pos);
        const int nvalid = __builtin_popcountll(__ballot(has));
        if (has) { s_key[w * k + rank] = key; s_pos[w * k + rank] = pos; }                                  // ranks [0, nvalid)
        if (lane >= nvalid && lane < k) { s_key[w * k + lane] = 0; s_pos[w * k + lane] = 0xffffffffu; }     // fewer than k columns
    }
    if (nw > 1) {
        __syncthreads();
        if (w != 0) return;
        unsigned lp;
        topk_of_list<K>(s_key, nw * k, k, lane, l_key[0], l_pos[0], has, rank, key, lp);
        pos = has ? s_pos[lp] : 0xffffffffu;
    }
    // wave 0: lane holds winner `rank` of the chunk (has) -- (key, column)
    const int nvalid = __builtin_popcountll(__ballot(has));
    // winners fill slots [0, nvalid), lanes nvalid..k-1 the padding slots (fewer than k columns): disjoint writers
    const bool padder = lane >= nvalid && lane < k;
    const T pad_v = -(T)INFINITY;
    const long long pad_i = 0x7fffffffffffffffLL;
    T val = pad_v;
    long long gi = pad_i;
    if (has) {
        val = KeyOf<T>::value(key);
        gi = idx_in ? idx_in[(long long)row * ld + pos] : (long long)pos + index_offset;
    }
    if (nchunks == 1) {
        if (has) { out_v[(long long)row * k + rank] = val; out_i[(long long)row * k + rank] = gi; }
        if (padder) { out_v[(long long)row * k + lane] = pad_v; out_i[(long long)row * k + lane] = pad_i; }
        return;
    }
    const long long cbase = ((long long)row * nchunks + chunk) * k;
    if (!finish) {
        if (has) { cand_v[cbase + rank] = val; cand_i[cbase + rank] = gi; }
        if (padder) { cand_v[cbase + lane] = pad_v; cand_i[cbase + lane] = pad_i; }
        return;
    }
    if (has) { publish(cand_v + cbase + rank, val); publish(cand_i + cbase + rank, gi); }
    if (padder) { publish(cand_v + cbase + lane, pad_v); publish(cand_i + cbase + lane, pad_i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the write-through stores have landed before the ticket
    int last = 0;
    if (lane == 0)
        last = __hip_atomic_fetch_add(&counters[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nchunks - 1u;
    if (!__builtin_amdgcn_readfirstlane(last)) return;          // wave-uniform: only wave 0 of the LAST workgroup goes on
    // Ordering of this hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1 table): every candidate is
    // stored write-through (sc1) by the publishing wave, which drains its stores (vmcnt(0)) BEFORE its ticket add; the last
    // arriver's add has returned (its value decided the branch above) before any of its candidate loads issue, and every such load
    // is an sc1 load (L1 bypassed).  That is an ISA argument, not a C++ one: relaxed atomics on different addresses carry no
    // order in the language, so the compiler barrier below pins the loads behind the branch (zero instructions; an agent-scope
    // acquire fence here would cost one buffer_inv + wait, ~1.7 us on a 10 us kernel), and rag4dyg_amd/build.py refuses a build
    // whose publish stores / consume loads are not lowered to sc1.
    asm volatile("" ::: "memory");
    const int m = nchunks * k;                                  // <= 1024 (host guarantees it when finish is set)
    const T* cv = cand_v + (long long)row * m;
    for (int i = lane; i < m; i += 64) s_key[i] = KeyOf<T>::make(consume(cv + i));
    unsigned lp;
    topk_of_list<K>(s_key, m, k, lane, l_key[0], l_pos[0], has, rank, key, lp);
    const int nv = __builtin_popcountll(__ballot(has));
    if (has) {
        out_v[(long long)row * k + rank] = KeyOf<T>::value(key);
        out_i[(long long)row * k + rank] = consume(cand_i + (long long)row * m + lp);
    }
    if (lane >= nv && lane < k) {
        out_v[(long long)row * k + lane] = pad_v;
        out_i[(long long)row * k + lane] = pad_i;
    }
}

// wavefronts per workgroup for a row of nseg segments: as few as keep the launch single (nchunks * k <= 1024), so that
// a row spreads over many CUs with about one wave per SIMD
static inline int pick_waves(long long nseg, int k) {
    // filter form (k <= 16): a segment costs ~330 instructions whatever k is, so up to 16 segments sit best in ONE workgroup
    // (no publication, no ticket: two memory round trips less); the iterative form (k rounds x 115) wants its waves spread
    if (nseg <= 4 || (k <= 16 && nseg <= CHUNK_WAVES)) return (int)nseg;
    if (k <= 16) return CHUNK_WAVES;
    for (int nw = 4; nw <= CHUNK_WAVES; nw *= 2)
        if (((nseg + nw - 1) / nw) * k <= 1024) return nw;
    return CHUNK_WAVES;
}

template <typename T>
size_t topk_ws_bytes(int rows, int n, int k) {
    size_t total = align_up((size_t)rows * sizeof(unsigned), 256);          // ticket counters
    long long cur = n;
    while (true) {
        const long long nseg = (cur + SEG - 1) / SEG;
        const int nw = pick_waves(nseg, k);
        const long long nchunks = (nseg + nw - 1) / nw;
        if (nchunks == 1) break;
        total += align_up((size_t)rows * nchunks * k * sizeof(T), 256) + align_up((size_t)rows * nchunks * k * 8, 256);
        if (nchunks * k <= 1024) break;                                      // finished inside the launch
        cur = nchunks * k;
    }
    return total + 256;
}
template size_t topk_ws_bytes<float>(int, int, int);
template size_t topk_ws_bytes<double>(int, int, int);

// rows x n values -> rows x k best (value, index).  `counters_zeroed`: the caller (the scan kernel) has already cleared
// the ticket counters at the start of ws on this stream.
template <typename T>
int topk_rows(const T* vals, const long long* idx_in, int rows, int n, long long ld, int k, long long index_offset,
              T* out_v, long long* out_i, void* ws, size_t ws_bytes, bool counters_zeroed, hipStream_t s) {
    R4D_REQUIRE(k >= 1 && k <= MAXK && k <= n, "topk: k=%d must be in [1, min(64, n=%d)]", k, n);
    R4D_REQUIRE(rows >= 1 && rows <= 65535, "topk: rows=%d outside [1, 65535]", rows);
    R4D_REQUIRE(n <= (1 << 30), "topk: n=%d exceeds 2^30 columns", n);          // 32-bit column arithmetic incl. the padded tail
    if (!ws || ws_bytes < topk_ws_bytes<T>(rows, n, k)) {
        set_error("topk: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    unsigned* counters = (unsigned*)ws;
    char* wp = (char*)ws + align_up((size_t)rows * sizeof(unsigned), 256);
    const T* cv = vals;
    const long long* ci = idx_in;
    long long cur = n, cld = ld;
    while (true) {
        const long long nseg = (cur + SEG - 1) / SEG;
        const int nw = pick_waves(nseg, k);
        const int nchunks = (int)((nseg + nw - 1) / nw);
        R4D_REQUIRE(nchunks <= 0x7fffffff / 2, "topk: row too long");
        const bool finish = nchunks > 1 && (long long)nchunks * k <= 1024;
        T* cand_v = nullptr;
        long long* cand_i = nullptr;
        if (nchunks > 1) {
            cand_v = (T*)wp; wp += align_up((size_t)rows * nchunks * k * sizeof(T), 256);
            cand_i = (long long*)wp; wp += align_up((size_t)rows * nchunks * k * 8, 256);
            // ticket counters: a 16-byte-multiple block at the start of ws, zeroed before EVERY launch that polls them
            if (finish && !counters_zeroed)
                R4D_HIP(hipMemsetAsync(counters, 0, align_up((size_t)rows * sizeof(unsigned), 16), s));
            counters_zeroed = false;
        }
        if (nchunks == 1) R4D_BRANCH(TOPK_ONE_WG); else if (finish) R4D_BRANCH(TOPK_TICKET); else R4D_BRANCH(TOPK_MULTI);
        if (sizeof(T) == 8) R4D_BRANCH(TOPK_F64);
        {
            ProfScope prof(PK_TOPK, (double)rows * cur * (sizeof(T) + (ci ? 8 : 0)), s);
            hipLaunchKernelGGL((topk_chunk_kernel<T>), dim3(nchunks, rows), dim3(64 * nw), 0, s, cv, ci, (int)cur, cld, k,
                               index_offset, nchunks, cand_v, cand_i, out_v, out_i, counters, finish ? 1 : 0);
            R4D_CHECK_LAUNCH("topk_chunk");
        }
        if (nchunks == 1 || finish) break;
        cv = cand_v; ci = cand_i; cur = (long long)nchunks * k; cld = cur; index_offset = 0;
    }
    return R4D_OK;
}
template int topk_rows<float>(const float*, const long long*, int, int, long long, int, long long, float*, long long*,
                              void*, size_t, bool, hipStream_t);
template int topk_rows<double>(const double*, const long long*, int, int, long long, int, long long, double*, long long*,
                               void*, size_t, bool, hipStream_t);

// int64 -> int32 index narrowing for the f64 (Jaccard) API
__global__ void narrow_idx_kernel(const long long* __restrict__ in, int32_t* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// ------------------------------------------------------------------------------------ shard merge
// vals / idx [G,Q,k] (shard-major: shard g holds the global indices of pool rows [offset_g, offset_{g+1}), offsets
// ascending with g, every shard list sorted by (value desc, index asc)) -> [Q,k].  List position g * k + j therefore
// orders equal values like the global index does; one wavefront per query, no gather pass.
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ vals, const long long* __restrict__ idx,
                                                         int G, int Q, int k, float* __restrict__ ov,
                                                         long long* __restrict__ oi) {
    __shared__ uint32_t l_key[4][64];
    __shared__ unsigned l_pos[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= Q) return;
    const int m = G * k;
    auto at = [&](int lp) { return ((long long)(lp / k) * Q + q) * k + lp % k; };
    bool has;
    int rank;
    uint32_t key;
    unsigned lp;
    if (m <= 64) {
        key = lane < m ? KeyOf<float>::make(vals[at(lane)]) : 0u;
        lp = (unsigned)lane;
        rank = wave_rank<uint32_t>(key, lp, m);
        has = lane < m && rank < k && key != 0;
    } else if (m <= 256) {
        uint32_t c[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = (lane * 4 + e < m) ? KeyOf<float>::make(vals[at(min(lane * 4 + e, m - 1))]) : 0u;
        wave_topk<uint32_t, 4>(c, (unsigned)lane * 4, k, lane, l_key[w], l_pos[w], has, rank, key, lp);
    } else {
        uint32_t c[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = (lane * 16 + e < m) ? KeyOf<float>::make(vals[at(min(lane * 16 + e, m - 1))]) : 0u;
        wave_topk<uint32_t, 16>(c, (unsigned)lane * 16, k, lane, l_key[w], l_pos[w], has, rank, key, lp);
    }
    const int nvalid = __builtin_popcountll(__ballot(has));
    if (has) {
        ov[(long long)q * k + rank] = KeyOf<float>::value(key);
        oi[(long long)q * k + rank] = idx[at((int)lp)];
    }
    if (lane >= nvalid && lane < k) {
        ov[(long long)q * k + lane] = -INFINITY;
        oi[(long long)q * k + lane] = 0x7fffffffffffffffLL;
    }
}

// general form (G * k > 1024): row-major candidate lists [Q, G*k], then the chunked top-k over them
__global__ __launch_bounds__(256) void gather_candidates_kernel(const float* __restrict__ vals,
                                                                const long long* __restrict__ idx, int G, int Q, int k,
                                                                float* __restrict__ ov, long long* __restrict__ oi) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)G * Q * k) return;
    const int j = (int)(t % k), g = (int)((t / k) % G), q = (int)(t / ((long long)k * G));
    const long long src = ((long long)g * Q + q) * k + j;
    ov[t] = vals[src];
    oi[t] = idx[src];
}

// ------------------------------------------------------------------------------------ full-row ranking
// perm = np.argsort(-x, kind='stable') for rows of any length: (1) chunks of 2048 (key, index) pairs are sorted in LDS
// (bitonic network, ascending in the DESCENDING-score key, index as the tie-break: all pairs distinct, so the order is
// unique and needs no stable network); (2) an element's rank is the sum over the row's sorted chunks of the number of
// pairs below it (12-step binary search per chunk), and perm[rank] = index.  A single chunk is its own permutation.
template <typename T> struct DescKey;
template <> struct DescKey<float> {
    typedef uint32_t K;
    static __device__ __forceinline__ K make(float x) {            // ascending key == descending score; NaN last
        if (x != x) return 0xffffffffu;
        x += 0.0f;
        const uint32_t b = __builtin_bit_cast(uint32_t, x);
        return ~(b ^ (uint32_t)(((int32_t)b >> 31) | (int32_t)0x80000000));
    }
};
template <> struct DescKey<double> {
    typedef uint64_t K;
    static __device__ __forceinline__ K make(double x) {
        if (x != x) return 0xffffffffffffffffull;
        x += 0.0;
        const uint64_t b = __builtin_bit_cast(uint64_t, x);
        return ~(b ^ (uint64_t)(((int64_t)b >> 63) | (int64_t)0x8000000000000000ull));
    }
};
constexpr int SORT_CHUNK = 2048;

template <typename K>
__device__ __forceinline__ bool pair_less(K ka, uint32_t ia, K kb, uint32_t ib) { return (ka < kb) | ((ka == kb) & (ia < ib)); }

// grid (nchunks, rows), block 1024.  Sorted pairs -> ws_k / ws_i [rows, nchunks, 2048]; a single chunk writes perm.
template <typename T>
__global__ __launch_bounds__(1024) void sort_chunks_kernel(const T* __restrict__ scores, int n, int nchunks,
                                                           typename DescKey<T>::K* __restrict__ ws_k,
                                                           uint32_t* __restrict__ ws_i, int32_t* __restrict__ perm) {
    typedef typename DescKey<T>::K K;
    __shared__ K sk[SORT_CHUNK];
    __shared__ uint32_t si[SORT_CHUNK];
    const int t = threadIdx.x, chunk = blockIdx.x, row = blockIdx.y;
    const T* v = scores + (long long)row * n;
    for (int j = t; j < SORT_CHUNK; j += 1024) {
        const long long col = (long long)chunk * SORT_CHUNK + j;
        const bool ok = col < n;
        sk[j] = ok ? DescKey<T>::make(v[col]) : (K)~(K)0;           // padding: above every real pair
        si[j] = ok ? (uint32_t)col : 0xffffffffu;
    }
    __syncthreads();
    for (int size = 2; size <= SORT_CHUNK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int l = ((t / stride) * 2 * stride) + (t % stride), r = l + stride;
            const bool asc = (l & size) == 0;
            const K kl = sk[l], kr = sk[r];
            const uint32_t il = si[l], ir = si[r];
            if (pair_less<K>(kr, ir, kl, il) == asc) { sk[l] = kr; si[l] = ir; sk[r] = kl; si[r] = il; }
            __syncthreads();
        }
    }
    if (nchunks == 1) {
        for (int j = t; j < n; j += 1024) perm[(long long)row * n + j] = (int32_t)si[j];
        return;
    }
    const long long base = ((long long)row * nchunks + chunk) * SORT_CHUNK;
    for (int j = t; j < SORT_CHUNK; j += 1024) { ws_k[base + j] = sk[j]; ws_i[base + j] = si[j]; }
}

// grid (cdiv(n, 2048), rows), block 1024: a workgroup ranks 2048 elements of a row.  Every sorted chunk of the row passes
// through LDS once per workgroup (coalesced 16-24 KB), and each thread binary-searches its two elements there: 12 LDS
// steps per chunk instead of 12 dependent L2 round trips (first version: 4.2 ms for 32 x 100k rows).
template <typename T>
__global__ __launch_bounds__(1024) void rank_scatter_kernel(const T* __restrict__ scores, int n, int nchunks,
                                                            const typename DescKey<T>::K* __restrict__ ws_k,
                                                            const uint32_t* __restrict__ ws_i, int32_t* __restrict__ perm) {
    typedef typename DescKey<T>::K K;
    __shared__ K ck[SORT_CHUNK];
    __shared__ uint32_t ci[SORT_CHUNK];
    const int row = blockIdx.y, t = threadIdx.x;
    const long long e0 = (long long)blockIdx.x * SORT_CHUNK + t, e1 = e0 + 1024;
    const bool ok0 = e0 < n, ok1 = e1 < n;
    const K k0 = DescKey<T>::make(scores[(long long)row * n + (ok0 ? e0 : n - 1)]);
    const K k1 = DescKey<T>::make(scores[(long long)row * n + (ok1 ? e1 : n - 1)]);
    const K* gk = ws_k + (long long)row * nchunks * SORT_CHUNK;
    const uint32_t* gi = ws_i + (long long)row * nchunks * SORT_CHUNK;
    long long r0 = 0, r1 = 0;
    for (int c = 0; c < nchunks; ++c, gk += SORT_CHUNK, gi += SORT_CHUNK) {
        __syncthreads();
        ck[t] = gk[t]; ck[t + 1024] = gk[t + 1024];
        ci[t] = gi[t]; ci[t + 1024] = gi[t + 1024];
        __syncthreads();
        int b0 = 0, b1 = 0;
#pragma unroll
        for (int s_ = SORT_CHUNK / 2; s_ >= 1; s_ >>= 1) {
            b0 += pair_less<K>(ck[b0 + s_ - 1], ci[b0 + s_ - 1], k0, (uint32_t)e0) ? s_ : 0;
            b1 += pair_less<K>(ck[b1 + s_ - 1], ci[b1 + s_ - 1], k1, (uint32_t)e1) ? s_ : 0;
        }
        b0 += pair_less<K>(ck[b0], ci[b0], k0, (uint32_t)e0) ? 1 : 0;
        b1 += pair_less<K>(ck[b1], ci[b1], k1, (uint32_t)e1) ? 1 : 0;
        r0 += b0; r1 += b1;
    }
    if (ok0) perm[(long long)row * n + r0] = (int32_t)e0;
    if (ok1) perm[(long long)row * n + r1] = (int32_t)e1;
}

template <typename T>
static size_t argsort_ws_bytes(int rows, int n) {
    if (rows <= 0 || n <= SORT_CHUNK) return 256;
    const size_t nchunks = ((size_t)n + SORT_CHUNK - 1) / SORT_CHUNK;
    return align_up((size_t)rows * nchunks * SORT_CHUNK * sizeof(typename DescKey<T>::K), 256) +
           align_up((size_t)rows * nchunks * SORT_CHUNK * 4, 256) + 256;
}

template <typename T>
static int argsort_desc(const T* scores, int rows, int n, int32_t* perm, void* ws, size_t ws_bytes, hipStream_t s) {
    typedef typename DescKey<T>::K K;
    R4D_REQUIRE(scores && perm, "argsort: null pointer");
    R4D_REQUIRE(rows >= 0 && rows <= 65535 && n >= 1 && n <= (1 << 27), "argsort: rows=%d (<= 65535), n=%d (<= 2^27) out of range", rows, n);
    if (rows == 0) return R4D_OK;
    const int nchunks = cdiv(n, SORT_CHUNK);
    R4D_REQUIRE(nchunks <= 65535, "argsort: n=%d too long", n);
    K* ws_k = nullptr;
    uint32_t* ws_i = nullptr;
    if (nchunks > 1) {
        if (!ws || ws_bytes < argsort_ws_bytes<T>(rows, n)) {
            set_error("argsort: workspace too small (%zu < %zu)", ws_bytes, argsort_ws_bytes<T>(rows, n));
            return R4D_ERR_WORKSPACE;
        }
        ws_k = (K*)ws;
        ws_i = (uint32_t*)((char*)ws + align_up((size_t)rows * nchunks * SORT_CHUNK * sizeof(K), 256));
    }
    if (nchunks > 1) R4D_BRANCH(ARGSORT_MULTI); else R4D_BRANCH(ARGSORT_ONE);
    ProfScope prof(PK_RANK_COUNT, (double)rows * n * (sizeof(T) + 4), s);
    hipLaunchKernelGGL((sort_chunks_kernel<T>), dim3(nchunks, rows), dim3(1024), 0, s, scores, n, nchunks, ws_k, ws_i, perm);
    R4D_CHECK_LAUNCH("sort_chunks");
    if (nchunks > 1) {
        hipLaunchKernelGGL((rank_scatter_kernel<T>), dim3(nchunks, rows), dim3(1024), 0, s, scores, n, nchunks, ws_k, ws_i, perm);
        R4D_CHECK_LAUNCH("rank_scatter");
    }
    return R4D_OK;
}

}  // namespace r4d

using namespace r4d;

extern "C" {

size_t r4d_topk_f32_workspace_bytes(int32_t rows, int32_t n, int32_t k) {
    if (rows <= 0 || n <= 0 || k <= 0) return 0;
    return topk_ws_bytes<float>(rows, n, k);
}

int r4d_topk_f32(const float* m_d, int32_t rows, int32_t n, int32_t k, float* out_val_d, int64_t* out_idx_d,
                 void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(m_d && out_val_d && out_idx_d && workspace_d, "topk_f32: null pointer");
    R4D_REQUIRE(rows >= 1 && n >= 1, "topk_f32: empty input");
    return topk_rows<float>(m_d, nullptr, rows, n, n, k, 0, out_val_d, (long long*)out_idx_d, workspace_d, workspace_bytes,
                            false, (hipStream_t)stream);
}

size_t r4d_merge_topk_workspace_bytes(int32_t G, int32_t Q, int32_t k) {
    if (G <= 0 || Q <= 0 || k <= 0) return 0;
    if ((long long)G * k <= 1024) return 256;
    return align_up((size_t)G * Q * k * 4, 256) + align_up((size_t)G * Q * k * 8, 256) + topk_ws_bytes<float>(Q, G * k, k);
}

int r4d_merge_topk_f32(const float* vals_d, const int64_t* idx_d, int32_t G, int32_t Q, int32_t k, float* out_val_d,
                       int64_t* out_idx_d, void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(vals_d && idx_d && out_val_d && out_idx_d, "merge_topk: null pointer");
    R4D_REQUIRE(G >= 1 && Q >= 1 && Q <= 65535 && k >= 1 && k <= MAXK, "merge_topk: G=%d Q=%d k=%d out of range", G, Q, k);
    hipStream_t s = (hipStream_t)stream;
    const long long tot = (long long)G * Q * k;
    if ((long long)G * k <= 1024) {
        ProfScope prof(PK_MERGE_TOPK, 24.0 * tot, s);
        hipLaunchKernelGGL(merge_topk_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, s, vals_d, (const long long*)idx_d, G, Q, k,
                           out_val_d, (long long*)out_idx_d);
        R4D_CHECK_LAUNCH("merge_topk");
        return R4D_OK;
    }
    if (!workspace_d || workspace_bytes < r4d_merge_topk_workspace_bytes(G, Q, k)) {
        set_error("merge_topk: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    float* cv = (float*)workspace_d;
    long long* ci = (long long*)((char*)workspace_d + align_up((size_t)tot * 4, 256));
    char* ws = (char*)ci + align_up((size_t)tot * 8, 256);
    {
        ProfScope prof(PK_MERGE_TOPK, 24.0 * tot, s);
        hipLaunchKernelGGL(gather_candidates_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, vals_d,
                           (const long long*)idx_d, G, Q, k, cv, ci);
        R4D_CHECK_LAUNCH("gather_candidates");
    }
    return topk_rows<float>(cv, ci, Q, G * k, (long long)G * k, k, 0, out_val_d, (long long*)out_idx_d, ws,
                            workspace_bytes - (size_t)(ws - (char*)workspace_d), false, s);
}

size_t r4d_argsort_workspace_bytes(int32_t rows, int32_t n, int32_t elem_bytes) {
    return elem_bytes == 8 ? argsort_ws_bytes<double>(rows, n) : argsort_ws_bytes<float>(rows, n);
}
int r4d_argsort_desc_f32(const float* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* workspace_d,
                         size_t workspace_bytes, void* stream) {
    return argsort_desc<float>(scores_d, rows, n, perm_d, workspace_d, workspace_bytes, (hipStream_t)stream);
}
int r4d_argsort_desc_f64(const double* scores_d, int32_t rows, int32_t n, int32_t* perm_d, void* workspace_d,
                         size_t workspace_bytes, void* stream) {
    return argsort_desc<double>(scores_d, rows, n, perm_d, workspace_d, workspace_bytes, (hipStream_t)stream);
}

size_t r4d_topk_f64_workspace_bytes(int32_t rows, int32_t n, int32_t k) {
    if (rows <= 0 || n <= 0 || k <= 0) return 0;
    return topk_ws_bytes<double>(rows, n, k) + align_up((size_t)rows * k * 8, 256);
}

int r4d_topk_f64(const double* m_d, int32_t rows, int32_t n, int32_t k, double* out_val_d, int32_t* out_idx_d,
                 void* workspace_d, size_t workspace_bytes, void* stream) {
    R4D_REQUIRE(m_d && out_val_d && out_idx_d && workspace_d, "topk_f64: null pointer");
    R4D_REQUIRE(rows >= 1 && n >= 1, "topk_f64: empty input");
    if (workspace_bytes < r4d_topk_f64_workspace_bytes(rows, n, k)) {
        set_error("topk_f64: workspace too small");
        return R4D_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    long long* idx64 = (long long*)workspace_d;
    const size_t skip = align_up((size_t)rows * k * 8, 256);
    // rows > 65535 (grid.y): slices of 65535 rows
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
        int rc = topk_rows<double>(m_d + (long long)r0 * n, nullptr, nr, n, n, k, 0, out_val_d + (long long)r0 * k,
                                   idx64 + (long long)r0 * k, (char*)workspace_d + skip, workspace_bytes - skip, false, s);
        if (rc) return rc;
    }
    const long long tot = (long long)rows * k;
    hipLaunchKernelGGL(narrow_idx_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, idx64, out_idx_d, tot);
    R4D_CHECK_LAUNCH("narrow_idx");
    return R4D_OK;
}

}  // extern "C"
