// Diagnostic micro-benchmark (not part of the product): a producer grid (122 workgroups, each writes 1 KB) followed by a one-workgroup
// consumer that reads all of it — the shape of k_reduce_c -> k_pose_solve_c — (a) as two kernels in a stream, (b) as ONE kernel whose
// workgroup 0 waits on a device-scope counter the producers bump after a release fence (and takes the data after an acquire).
// Question: does the in-kernel hand-over cost less than the kernel boundary on this 8-XCD part?  (Round 2 measured a last-workgroup
// ticket in a 427-workgroup kernel at +14 us.)  The consumer's own work is emulated by `spin` dependent FMAs.
//   hipcc --offload-arch=gfx950 -O3 fused_handover.hip -o fused_handover
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

extern __shared__ double dyn[];
constexpr int NP = 122, PER = 128;          // producers, doubles each

__device__ __forceinline__ void produce(double *buf, int b, int it) {
    for (int i = threadIdx.x; i < PER; i += blockDim.x) buf[b * PER + i] = (double)(b + i + it);
}
__device__ __forceinline__ void consume(const double *buf, double *out, int spin) {
    double s = 0;
    for (int i = threadIdx.x; i < NP * PER; i += blockDim.x) s += buf[i];
    for (int k = 0; k < spin; ++k) s = fma(s, 1.0000001, 1e-9);
    dyn[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < (int)blockDim.x; ++i) t += dyn[i]; out[0] = t; }
}
__global__ __launch_bounds__(1024) void k_prod(double *buf, int it) { produce(buf, blockIdx.x, it); }
__global__ __launch_bounds__(1024) void k_cons(const double *buf, double *out, int spin) { consume(buf, out, spin); }
__global__ __launch_bounds__(1024) void k_fused(double *buf, double *out, unsigned *cnt, int it, int spin) {
    if (blockIdx.x > 0) {
        produce(buf, blockIdx.x - 1, it);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) {
        const unsigned want = (unsigned)NP * (unsigned)(it + 1);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    consume(buf, out, spin);
}

int main() {
    double *buf, *out; unsigned *cnt;
    hipMalloc(&buf, NP * PER * 8); hipMalloc(&out, 8); hipMalloc(&cnt, 4); hipMemset(cnt, 0, 4);
    hipFuncSetAttribute((const void *)k_cons, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipFuncSetAttribute((const void *)k_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    for (int spin : {0, 2000}) for (int lds_kb : {8, 150}) {
        const int N = 2000;
        auto run2 = [&](int n) { for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_prod, dim3(NP), dim3(1024), 0, 0, buf, i); hipLaunchKernelGGL(k_cons, dim3(1), dim3(1024), (size_t)lds_kb * 1024, 0, buf, out, spin); } };
        run2(50); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now(); run2(N); hipDeviceSynchronize();
        const double two = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        hipMemset(cnt, 0, 4);
        int it = 0;
        auto run1 = [&](int n) { for (int i = 0; i < n; ++i, ++it) hipLaunchKernelGGL(k_fused, dim3(NP + 1), dim3(1024), (size_t)lds_kb * 1024, 0, buf, out, cnt, it, spin); };
        run1(50); hipDeviceSynchronize();
        t0 = std::chrono::steady_clock::now(); run1(N); hipDeviceSynchronize();
        const double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("consumer work %4d FMAs, %3d KB LDS per workgroup: two kernels %6.2f us, one kernel with the hand-over %6.2f us   (out %.3g)\n", spin, lds_kb, two, one, h);
    }
    return 0;
}
