// Diagnostic micro-benchmark (not part of the product): does preloading the leading kernel arguments into SGPRs
// (-mllvm -amdgpu-kernarg-preload-count=N: the command processor hands them over at wave launch instead of the wave's first s_load from the
// kernarg segment) shorten a latency-bound kernel on gfx950?  A one-workgroup kernel follows a pointer chain of `depth` dependent global
// loads starting at an argument; back-to-back launches, time per launch.  Build twice:
//   hipcc --offload-arch=gfx950 -O3 kernarg_preload.hip -o kernarg_preload_off
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 kernarg_preload.hip -o kernarg_preload_on
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_chain(const long long *p, long long *out, int depth) {
    long long i = threadIdx.x == 0 ? p[0] : 0;
    for (int d = 1; d < depth; ++d) i = p[i];
    if (threadIdx.x == 0) out[0] = i;
}
int main() {
    long long *p, *out, h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i + 1) % 64;
    (void)hipMalloc(&p, sizeof(h)); (void)hipMalloc(&out, 8); (void)hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
    for (int depth : {1, 2, 3}) {
        const int N = 4000;
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, p, out, depth);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, p, out, depth);
        (void)hipDeviceSynchronize();
        printf("depth %d: %.3f us per launch\n", depth, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    return 0;
}
