// Per-launch floor of dependent tiny kernels on one stream: launched one by one vs replayed as a captured graph.
// hipcc --offload-arch=gfx950 -O2 tools/launch_floor.hip -o tools/_bin/launch_floor && tools/_bin/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main() {
    float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NK = 70, REP = 50;
    for (int blocks : {1, 32, 256}) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, blocks * 256);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r)
            for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, blocks * 256);
        CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("blocks %3d  eager: %.2f us per launch\n", blocks, us / (REP * NK));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, s, p, blocks * 256);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("blocks %3d  graph: %.2f us per kernel node\n", blocks, us / (REP * NK));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
