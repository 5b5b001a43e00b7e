// Diagnostic micro-benchmark: which waves of a 1024-thread workgroup share a matrix core on gfx950?
// A chosen set of waves runs a chain of 64 dependent v_mfma_f64_16x16x4_f64; the others wait at the barrier.
//   hipcc --offload-arch=gfx950 -O3 mfma_sharing.hip -o mfma_sharing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k(unsigned long long *out, double *sink, unsigned mask) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if ((mask >> wave) & 1) {
        v4d c = {1.0, 2.0, 3.0, 4.0};
        double a = 1.0 + lane * 1e-3, b = 0.5;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 64; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        asm volatile("" : "+v"(c));
        const double r = c[0] + c[1] + c[2] + c[3];
        const int rl = __builtin_amdgcn_readfirstlane(__double2loint(r));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[wave] = t1 - t0 + (rl == 12345 ? 1 : 0);
        sink[threadIdx.x] = r;
    }
    __syncthreads();
}

int main() {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 1024 * 8);
    const unsigned masks[] = {0x1, 0x3, 0x11, 0xF, 0x1111, 0x5555, 0x00FF, 0xFFFF};
    for (unsigned m : masks) {
        hipMemset(out, 0, 16 * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, out, sink, m);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
        printf("waves mask 0x%04x: ticks per 64 dependent MFMAs per wave:", m);
        for (int w = 0; w < 16; ++w) if ((m >> w) & 1) printf(" w%d=%llu", w, h[w]);
        printf("\n");
    }
    return 0;
}
