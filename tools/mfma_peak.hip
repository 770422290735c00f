// Micro-benchmark (tuning aid, not product): the f32 MFMA issue rate a wavefront can sustain with NO memory traffic,
// for the accumulator patterns the GEMM kernels use.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// NV independent VALU ops (v_cndmask-like selects / adds) between consecutive MFMAs: does plain VALU work overlap with
// the matrix pipe, or does it steal MFMA issue slots?
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k32v(float* out, int iters, float a, float b) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float& r = x[(u * 2 + i + v) & 7];
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(bv));
                    else if (KIND == 1) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(r));
                    else asm volatile("ds_write_b32 %0, %1" :: "v"(0), "v"(r) : "memory");
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_wave_iter, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    launch(blocks, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = flop_per_wave_iter * iters * blocks * 4.0;
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flop / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
    for (int occ = 1; occ <= 4; occ *= 2) {
        const int blocks = 256 * occ;    // 256-thread blocks: 1 wave per SIMD per block
        run("32x32x2 1 acc (chain)", [&](int b, int it) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<2>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("16x16x4 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 64 * 2048.0, blocks);
        run("32x32x2 2acc + 1 v_add/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<1, 0>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2acc + 2 v_add/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<2, 0>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2acc + 4 v_add/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<4, 0>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2acc + 1 cndmask/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<1, 1>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2acc + 4 cndmask/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<4, 1>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("32x32x2 2acc + 1 ds_write/mfma", [&](int b, int it) { hipLaunchKernelGGL((k32v<1, 2>), dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 32 * 4096.0, blocks);
        run("16x16x4 16 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<16>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 64 * 2048.0, blocks);
    }
    return 0;
}
