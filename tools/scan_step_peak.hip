// Micro-benchmark (tuning aid, not product): cycles per v_mfma_f32_32x32x16_bf16 of the pool scan's inner step with NO memory
// traffic -- six dependent MFMAs on ONE accumulator per 8 fp32 inputs that are split into three bf16 terms first -- for the
// instruction orders the kernel could use.  Build: hipcc -O3 --offload-arch=gfx950 tools/scan_step_peak.hip -o tools/_bin/scan_step_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) { const f32x2v v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v)); }
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pk(s0, s1);
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, u32x4v& h, u32x4v& m, u32x4v& l) {
    unsigned hh[4], mm[4], ll[4];
    split_pair(a.x, a.y, hh[0], mm[0], ll[0]); split_pair(a.z, a.w, hh[1], mm[1], ll[1]);
    split_pair(b.x, b.y, hh[2], mm[2], ll[2]); split_pair(b.z, b.w, hh[3], mm[3], ll[3]);
    h = u32x4v{hh[0], hh[1], hh[2], hh[3]}; m = u32x4v{mm[0], mm[1], mm[2], mm[3]}; l = u32x4v{ll[0], ll[1], ll[2], ll[3]};
}
#define MF(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, A_), __builtin_bit_cast(bf16x8v, B_), ACC_, 0, 0, 0)
#define STEP6(ACC_, Q_, BH_, BM_, BL_) do { MF(Q_[2], BH_, ACC_); MF(Q_[0], BL_, ACC_); MF(Q_[1], BM_, ACC_); MF(Q_[1], BH_, ACC_); MF(Q_[0], BM_, ACC_); MF(Q_[0], BH_, ACC_); } while (0)

// MODE 0: MFMAs only (operands fixed); 1: split then MFMAs, compiler order; 2: split of the next step pinned into the MFMA gaps;
// 3: like 2 with TWO accumulators (two tiles interleaved)
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, const float4* in, int iters, unsigned long long* cyc) {
    f32x16 acc, acc1;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    u32x4v q[3];
    float4 x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + 64 * i];
    split8(x[0], x[1], q[0], q[1], q[2]);
    u32x4v ch, cm, cl;
    split8(x[2], x[3], ch, cm, cl);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE == 0) {
                STEP6(acc, q, ch, cm, cl);
            } else if (MODE == 1) {
                x[2 * s].x += 1.0f;                                  // new data every step
                split8(x[2 * s], x[2 * s + 1], ch, cm, cl);
                STEP6(acc, q, ch, cm, cl);
            } else if (MODE == 2) {
                u32x4v nh, nm, nl;
                x[2 * s].x += 1.0f;
                split8(x[2 * s], x[2 * s + 1], nh, nm, nl);
                STEP6(acc, q, ch, cm, cl);
#pragma unroll
                for (int j = 0; j < 6; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
                ch = nh; cm = nm; cl = nl;
            } else {
                u32x4v nh, nm, nl;
                x[2 * s].x += 1.0f;
                split8(x[2 * s], x[2 * s + 1], nh, nm, nl);
                MF(q[2], ch, acc); MF(q[2], cm, acc1); MF(q[0], cl, acc); MF(q[0], ch, acc1); MF(q[1], cm, acc); MF(q[1], cl, acc1);
#pragma unroll
                for (int j = 0; j < 6; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
                ch = nh; cm = nm; cl = nl;
            }
        }
    }
    asm volatile("" :: "v"(acc[0]), "v"(acc1[0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + x[0].x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int WAVES>
static void run(const char* name, float* out, float4* in, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, in, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, in, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = 24.0 * iters;                                  // MFMAs per wave
    printf("%-44s waves/SIMD=%d  %7.3f ms  %6.1f s_memtime-cycles per MFMA per wave  (%.1f per SIMD)\n", name, WAVES / 4, ms, c / mf, c / mf / (WAVES / 4));
}
// straight-line code executed ONCE per wavefront (iters = 1: 24 MFMAs + 4 splits, like one tile of the scan): cold and warm instruction cache
template <int MODE, int WAVES>
static void once(const char* name, float* out, float4* in, unsigned long long* cyc) {
    unsigned long long c[4];
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, in, 1, cyc);
        hipDeviceSynchronize();
        hipMemcpy(&c[i], cyc, 8, hipMemcpyDeviceToHost);
    }
    hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(512), 0, 0, out, in, 100, cyc);       // other kernels in between
    hipLaunchKernelGGL((k<3, 8>), dim3(256), dim3(512), 0, 0, out, in, 100, cyc);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, in, 1, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&c[3], cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD=%d once: first launch %llu cycles, second %llu, third %llu, after two other kernels %llu (24 MFMAs: %d at 32)\n", name, WAVES / 4, c[0], c[1], c[2], c[3], 24 * 32);
}
int main() {
    float* out; float4* in; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 64 * 8 * 16 * 8); hipMalloc(&cyc, 8);
    hipMemset(in, 0x3c, 64 * 8 * 16 * 8);
    once<2, 4>("pinned, straight-line", out, in, cyc);
    once<2, 8>("pinned, straight-line", out, in, cyc);
    once<1, 8>("compiler order, straight-line", out, in, cyc);
    run<0, 4>("MFMA only, one chain", out, in, cyc);
    run<0, 8>("MFMA only, one chain", out, in, cyc);
    run<1, 4>("split + MFMA, compiler order", out, in, cyc);
    run<1, 8>("split + MFMA, compiler order", out, in, cyc);
    run<2, 4>("split pinned into the MFMA gaps", out, in, cyc);
    run<2, 8>("split pinned into the MFMA gaps", out, in, cyc);
    run<3, 4>("pinned, two accumulators", out, in, cyc);
    run<3, 8>("pinned, two accumulators", out, in, cyc);
    return 0;
}
