// The two-term fp16 form of an fp32 number (gemm_h2.hip, attention_h2.hip):  x = hi + 2^-11 lo' + e,  hi = RN16(x),
// lo' = RN16((x - hi) 2^11),  |e| <= 2^-22 |x|  (the header of gemm_h2.hip has the derivation and the range).
#pragma once

namespace r4d {

typedef float h2_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));

constexpr float H2_A_PRESCALE = 0.25f;        // 2^-2 on an activation operand (exact), undone by the consumer
constexpr float H2_A_UNSCALE = 4.0f;
constexpr float H2_LO_SCALE = 2048.0f;        // 2^11 on the second term of BOTH operands
constexpr float H2_LO_UNSCALE = 1.0f / 2048.0f;

// two fp32 -> packed (hi, hi), (lo', lo') of x * PRE
template <bool PRESCALE>
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& l) {
    h2_f32x2 v = {x0, x1};
    if (PRESCALE) v = v * H2_A_PRESCALE;
    const h2_f16x2 hh = __builtin_convertvector(v, h2_f16x2);         // v_cvt_pk_f16_f32: RNE
    const h2_f32x2 hf = __builtin_convertvector(hh, h2_f32x2);
    const h2_f32x2 r = (v - hf) * H2_LO_SCALE;                        // exact
    const h2_f16x2 ll = __builtin_convertvector(r, h2_f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// "h2 word" of an element: hi in the low half, lo' in the high half -- as an MFMA operand a register of such words is
// two k-slots (hi, lo') of ONE element.  Against it the other operand takes the forms
//     F1 = (hi, 0)    ->  sum hi.hi                    (first accumulator set)
//     F2 = (lo', hi)  ->  sum hi.lo' + lo'.hi          (second set, factor 2^11)
// so the three partial products of the f16x2 form cost two instructions per 8 elements (attention_h2.hip).
template <bool PRESCALE>
__device__ __forceinline__ void h2_words(float x0, float x1, unsigned& w0, unsigned& w1) {
    unsigned h, l;
    split2_pair<PRESCALE>(x0, x1, h, l);
    w0 = __builtin_amdgcn_perm(l, h, 0x05040100u);                    // bytes: h.0 h.1 l.0 l.1
    w1 = __builtin_amdgcn_perm(l, h, 0x07060302u);                    //        h.2 h.3 l.2 l.3
}
__device__ __forceinline__ void h2_forms(float x0, float x1, unsigned& f1_0, unsigned& f1_1, unsigned& f2_0, unsigned& f2_1) {
    unsigned h, l;
    split2_pair<false>(x0, x1, h, l);
    f1_0 = h & 0xffffu;
    f1_1 = h >> 16;
    f2_0 = __builtin_amdgcn_perm(h, l, 0x05040100u);                  // bytes: l.0 l.1 h.0 h.1
    f2_1 = __builtin_amdgcn_perm(h, l, 0x07060302u);
}

}  // namespace r4d
