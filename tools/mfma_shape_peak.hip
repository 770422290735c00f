// Micro-benchmark (tuning aid, not product): sustained bf16 MFMA rate of the two shapes gemm_s3 could use, every CU busy for
// milliseconds (the clock the chip holds under load depends on the shape): v_mfma_f32_32x32x16_bf16 (32 cycles, 32 Kflop)
// against v_mfma_f32_16x16x32_bf16 (16 cycles, 16 Kflop), operands in registers, 4 independent accumulators, 1 or 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -w tools/mfma_shape_peak.hip -o tools/_bin/mfma_shape_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, int iters, unsigned seed, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4v a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 4; ++c) { a[i][c] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + seed * (i + 1) + c) & 0x007f007fu); b[i][c] = 0x3f003f00u ^ ((threadIdx.x * 40503u + seed + c * 7 + i) & 0x007f007fu); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a[(u + i) & 1]), __builtin_bit_cast(bf16x8v, b[u & 1]), acc[i], 0, 0, 0);
    }
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(float* out, int iters, unsigned seed, unsigned long long* clk) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    u32x4v a[2], b[2];
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 4; ++c) { a[i][c] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + seed * (i + 1) + c) & 0x007f007fu); b[i][c] = 0x3f003f00u ^ ((threadIdx.x * 40503u + seed + c * 7 + i) & 0x007f007fu); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, a[(u + i) & 1]), __builtin_bit_cast(bf16x8v, b[(u + (i >> 2)) & 1]), acc[i], 0, 0, 0);
    }
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[5][0]), "v"(acc[10][0]), "v"(acc[15][0]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <typename F>
static void run(const char* name, F launch, double flop_per_wave_iter, int waves) {
    const int iters = 60000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 64 * waves * 4); hipMalloc(&clk, 16);
    launch(out, 200, clk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    printf("%-40s waves/SIMD=%d  %8.2f ms  %7.1f TFLOP/s   shader clock %.2f GHz\n", name, waves / 4, ms, flop_per_wave_iter * iters * 256.0 * waves / ms / 1e9,
           (double)c[0] / ((double)c[1] * 10.0));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run("v_mfma_f32_32x32x16_bf16", [](float* o, int it, unsigned long long* c) { hipLaunchKernelGGL((k32<4>), dim3(256), dim3(256), 0, 0, o, it, 1u, c); }, 32.0 * 32768, 4);
        run("v_mfma_f32_16x16x32_bf16", [](float* o, int it, unsigned long long* c) { hipLaunchKernelGGL((k16<4>), dim3(256), dim3(256), 0, 0, o, it, 1u, c); }, 64.0 * 16384, 4);
        run("v_mfma_f32_32x32x16_bf16", [](float* o, int it, unsigned long long* c) { hipLaunchKernelGGL((k32<8>), dim3(256), dim3(512), 0, 0, o, it, 1u, c); }, 32.0 * 32768, 8);
        run("v_mfma_f32_16x16x32_bf16", [](float* o, int it, unsigned long long* c) { hipLaunchKernelGGL((k16<8>), dim3(256), dim3(512), 0, 0, o, it, 1u, c); }, 64.0 * 16384, 8);
    }
    return 0;
}
