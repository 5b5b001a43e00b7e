// Diagnostic micro-benchmark (not part of the product): what a kernel launch costs in a stream on gfx950 as a function
// of grid size, block size and dynamic LDS, for kernels that do nothing.  Back-to-back launches, wall time / count.
//   hipcc --offload-arch=gfx950 -O3 launch_overhead.hip -o launch_overhead
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

extern __shared__ double dyn[];
__global__ void k_empty(double *out) { if (out && threadIdx.x == 0 && blockIdx.x == 1u << 30) out[0] = dyn[0]; }
__global__ void k_touch(double *out, int n) {      // every workgroup writes n doubles of LDS (to see LDS allocation alone is not it)
    for (int i = threadIdx.x; i < n; i += blockDim.x) dyn[i] = i;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 1u << 30) out[0] = dyn[n - 1];
}

int main() {
    double *out; hipMalloc(&out, 8);
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipFuncSetAttribute((const void *)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    struct Cfg { int grid, block, lds_kb; } cfgs[] = {{1, 64, 0}, {1, 1024, 0}, {1, 1024, 150}, {256, 64, 0}, {427, 64, 0}, {427, 256, 0}, {427, 1024, 0},
                                                      {427, 1024, 96}, {250, 1024, 150}, {427, 256, 96}, {177, 896, 8}, {91, 1024, 8}, {2000, 64, 0}};
    for (auto c : cfgs) {
        const int N = 2000;
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), (size_t)c.lds_kb * 1024, 0, out);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), (size_t)c.lds_kb * 1024, 0, out);
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("empty kernel %5d x %4d threads, %3d KB LDS: %6.2f us per launch (back to back in one stream)\n", c.grid, c.block, c.lds_kb, us);
    }
    return 0;
}
