// Probe (tuning/verification aid): does the raw-buffer range check of gfx950 include the SCALAR offset?
// Kernels that park a row index in soffset and rely on "past the range reads as zero" need to know.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const float* p, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 64, 0x00020000);
    const int lane = threadIdx.x;
    int soff = 0;
    asm volatile("s_mov_b32 %0, 128" : "=s"(soff));
    out[0 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0));       // in range
    out[1 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 128, 0, 0));     // voffset past
    out[2 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, soff, 0));    // soffset past
    out[3 * 64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 32, soff / 4, 0)); // 32 + 32 = 64: sum past
}
int main() {
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("in range            -> %.0f (expect 1000)\n", r[0]);
    printf("voffset 128 (past)  -> %.0f (0 = range-checked)\n", r[64]);
    printf("soffset 128 (past)  -> %.0f (0 = soffset is range-checked, 1032 = it is NOT)\n", r[128]);
    printf("voffset 32 + soffset 32 = 64 (past by sum) -> %.0f (0 = sum checked, 1016 = only voffset checked)\n", r[192]);
    return 0;
}
