// Probe for the round-4 finding (DESIGN_LOG 11.2 / ADVICE r4 medium): a packed-fp32 join  acc0 + c * acc1  of two MFMA accumulator
// sets, formed by the compiler as v_pk_fma_f32 / v_pk_mul_f32 with an SGPR multiplier, against the same arithmetic pinned scalar.
// Four forms of the multiplier operand x two accumulator layouts (A.B and the operand-swapped B.A the removed epilogue used),
// straight after the MFMAs (compiler-inserted wait states only).  Prints the lanes / registers that differ, if any.
//   hipcc -w -O3 --offload-arch=gfx950 tools/pk_fma_probe.hip -o tools/_bin/pk_fma_probe && tools/_bin/pk_fma_probe
// Result on MI355X / ROCm 7.2 (round 5): every form agrees with the scalar arithmetic and is bit-reproducible -- the instruction forms
// are exonerated; tests/test_gpu_dispatch.py runs this probe, rag4dyg_amd/build.py lists the forms the shipped kernels may contain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM, bool SWAP>
__global__ __launch_bounds__(64) void probe(const _Float16* __restrict__ a, const _Float16* __restrict__ b, float c,
                                            float bias, float* __restrict__ out_pk, float* __restrict__ out_sc, int reps) {
    const int lane = threadIdx.x;
    f16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = a[lane * 8 + i]; fb[i] = b[lane * 8 + i]; }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int k = 0; k < reps; ++k) {
        if (SWAP) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc1, 0, 0, 0);
        }
    }
    // packed form: left to the vectoriser / written on 2-vectors
    float pk[16], sc[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 x0 = {acc0[r], acc0[r + 1]}, x1 = {acc1[r], acc1[r + 1]};
        f32x2 v;
        if (FORM == 0) v = x1 * c + x0;                                    // SGPR scalar broadcast, fma contraction
        else if (FORM == 1) v = (x1 * c + x0) * 4.0f + bias;               // + the epilogue's unscale and bias
        else if (FORM == 2) { const f32x2 cc = {c, c}; v = __builtin_elementwise_fma(x1, cc, x0); }
        else if (FORM == 5) {     // the two forms gemm_h2 / attention_h2 contain most, spelled out: SGPR pair, low word broadcast to both halves
            unsigned long long cp = (unsigned long long)__builtin_bit_cast(unsigned, c) | ((unsigned long long)__builtin_bit_cast(unsigned, bias) << 32);
            f32x2 t, u;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x1), "s"(cp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(u) : "v"(t), "s"(cp), "v"(x0));
            v = u;
        }
        else if (FORM == 6) {     // a GENUINE SGPR pair (two different constants, default op_sel_hi): low half x c, high half x bias
            unsigned long long cp = (unsigned long long)__builtin_bit_cast(unsigned, c) | ((unsigned long long)__builtin_bit_cast(unsigned, bias) << 32);
            f32x2 u;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(u) : "v"(x1), "s"(cp), "v"(x0));
            v = u;
        }
        else if (FORM == 4) { v = (x0 - x1) * c; v = v * bias; v = x0 - v * c; }      // v_pk_mul with a broadcast SGPR, fma with negated operands (the A-split of gemm_h2's staging)
        else {                                                             // GELU on top (the variant that was not reproducible)
            v = (x1 * c + x0) * 4.0f + bias;
            const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
            const f32x2 aa = v * v * k1 + k0;
            const f32x2 w = v * aa;
            f32x2 e;
            e.x = __builtin_amdgcn_exp2f(w.x); e.y = __builtin_amdgcn_exp2f(w.y);
            e = e + 1.0f;
            f32x2 rr;
            rr.x = __builtin_amdgcn_rcpf(e.x); rr.y = __builtin_amdgcn_rcpf(e.y);
            v = v * rr;
        }
        pk[r] = v.x; pk[r + 1] = v.y;
    }
    // scalar form: every element on its own, opaque to the vectoriser
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x0 = acc0[r], x1 = acc1[r], cs = c;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(cs));
        float v = (FORM == 4 || FORM == 5 || FORM == 6) ? 0.f : __builtin_fmaf(x1, cs, x0);
        if (FORM == 6) { float b2 = bias; asm volatile("" : "+v"(b2)); v = __builtin_fmaf(x1, (r & 1) ? b2 : cs, x0); }
        if (FORM == 5) { float t = x1 * cs; asm volatile("" : "+v"(t)); v = __builtin_fmaf(t, cs, -x0); }
        if (FORM == 4) { float b2 = bias; asm volatile("" : "+v"(b2)); v = (x0 - x1) * cs; v = v * b2; v = __builtin_fmaf(-v, cs, x0); }
        if (FORM == 1 || FORM == 3) v = __builtin_fmaf(v, 4.0f, bias);
        if (FORM == 3) {
            const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, k1 = 0.044715f * k0;
            const float aa = __builtin_fmaf(v * v, k1, k0);
            v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * aa));
        }
        asm volatile("" : "+v"(v));
        sc[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { out_pk[(blockIdx.x * 64 + lane) * 16 + r] = pk[r]; out_sc[(blockIdx.x * 64 + lane) * 16 + r] = sc[r]; }
}

template <int FORM, bool SWAP>
static int run(const char* name, const _Float16* a, const _Float16* b, float* opk, float* osc, int blocks) {
    int bad_runs = 0;
    std::vector<float> hp(blocks * 64 * 16), hs(blocks * 64 * 16), first;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((probe<FORM, SWAP>), dim3(blocks), dim3(64), 0, 0, a, b, 1.0f / 2048.0f, 0.37f, opk, osc, 8);
        hipDeviceSynchronize();
        hipMemcpy(hp.data(), opk, hp.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hs.data(), osc, hs.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, lanes[64] = {0};
        float worst = 0.f;
        for (size_t i = 0; i < hp.size(); ++i) {
            // contraction differences allowed: fma(x1, c, x0) then *4 + bias may round differently by 1 ulp when the compiler does not contract; report > 2 ulp
            const float d = fabsf(hp[i] - hs[i]), tol = 4e-7f * fabsf(hs[i]) + 1e-30f;
            if (!(d <= tol)) { ++bad; lanes[(i / 16) % 64]++; if (d > worst) worst = d; }
        }
        bool rep = true;
        if (it == 0) first = hp; else rep = memcmp(first.data(), hp.data(), hp.size() * 4) == 0;
        if (bad || !rep) {
            ++bad_runs;
            printf("%s run %d: %d elements differ (worst %.3e)%s; lanes:", name, it, bad, worst, rep ? "" : "; NOT REPRODUCIBLE run to run");
            for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l);
            printf("\n");
        }
    }
    if (!bad_runs) printf("%s: packed == scalar (to 2 ulp), 5 runs bit-identical\n", name);
    return bad_runs;
}

int main() {
    const int blocks = 2048;
    std::vector<_Float16> ha(64 * 8), hb(64 * 8);
    srand(5);
    for (auto& x : ha) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    for (auto& x : hb) x = (_Float16)((rand() % 2001 - 1000) / 700.0f);
    _Float16 *a, *b;
    float *opk, *osc;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2);
    hipMalloc(&opk, (size_t)blocks * 64 * 16 * 4); hipMalloc(&osc, (size_t)blocks * 64 * 16 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<0, false>("join, A.B", a, b, opk, osc, blocks);
    bad += run<0, true>("join, B.A (swapped)", a, b, opk, osc, blocks);
    bad += run<1, false>("join+unscale+bias, A.B", a, b, opk, osc, blocks);
    bad += run<1, true>("join+unscale+bias, B.A", a, b, opk, osc, blocks);
    bad += run<2, false>("elementwise_fma {c,c}, A.B", a, b, opk, osc, blocks);
    bad += run<2, true>("elementwise_fma {c,c}, B.A", a, b, opk, osc, blocks);
    bad += run<3, false>("join+GELU, A.B", a, b, opk, osc, blocks);
    bad += run<3, true>("join+GELU, B.A", a, b, opk, osc, blocks);
    bad += run<4, false>("sub, mul, negated fma, A.B", a, b, opk, osc, blocks);
    bad += run<4, true>("sub, mul, negated fma, B.A", a, b, opk, osc, blocks);
    bad += run<5, false>("v_pk_mul V,V,S op_sel_hi:[1,0] + v_pk_fma V,V,S,V neg addend, A.B", a, b, opk, osc, blocks);
    bad += run<5, true>("v_pk_mul V,V,S op_sel_hi:[1,0] + v_pk_fma V,V,S,V neg addend, B.A", a, b, opk, osc, blocks);
    bad += run<6, false>("v_pk_fma V,V,S,V with a genuine SGPR pair, A.B", a, b, opk, osc, blocks);
    bad += run<6, true>("v_pk_fma V,V,S,V with a genuine SGPR pair, B.A", a, b, opk, osc, blocks);
    printf("%s\n", bad ? "MISMATCHES FOUND" : "all forms agree");
    return bad ? 1 : 0;
}
