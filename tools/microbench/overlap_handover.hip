// Diagnostic micro-benchmark (not part of the product): k_reduce_c and k_pose_solve_c as ONE launch in which the solver's workgroup does NOT wait
// for all the summing workgroups at once (tools/microbench/fused_handover.hip measured that: +2.7 us against the kernel boundary) but in two
// steps, with work of its own between them — the shape the real pair has:
//   group A: 20 fast producers (one round trip: the speed-bias rows of the image: IMU + prior terms, nothing of the item sums), 85 KB
//   group B: 101 slow producers (three dependent round trips: list bounds -> list -> slabs), 35 KB (camera block, right-hand side, chi2)
//   consumer: wait A -> read A's 85 KB -> `work1` ticks of dependent work (level 0 of the chain) -> wait B -> read B's 35 KB -> `work2` ticks.
// (a) two kernels in a stream (producers, then the consumer reading everything at its start); (b) one kernel with the two hand-overs.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/overlap_handover.hip -o tools/microbench/overlap_handover
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

extern __shared__ double dyn[];
constexpr int NA = 20, NB = 101, PA = 544, PB = 44;         // producers and doubles each: 20 x 544 x 8 = 85 KB, 101 x 44 x 8 = 35 KB

__device__ __forceinline__ double chase(const int *idx, int hops, int start) {      // `hops` dependent round trips
    int p = start;
    for (int h = 0; h < hops; ++h) p = idx[p & 1023] + h;
    return (double)p;
}
__device__ __forceinline__ void produce(double *buf, const int *idx, int b, int it) {
    if (b < NA) { const double v = chase(idx, 1, b + it); for (int i = threadIdx.x; i < PA; i += blockDim.x) buf[b * PA + i] = v + i; }
    else { const double v = chase(idx, 3, b + it); for (int i = threadIdx.x; i < PB; i += blockDim.x) buf[NA * PA + (b - NA) * PB + i] = v + i; }
}
__device__ __forceinline__ double spin_ticks(long ticks, double s) {
    const long t0 = __builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < ticks) s = fma(s, 1.0000001, 1e-9);
    return s;
}
__device__ __forceinline__ double read_range(const double *buf, int lo, int hi) {
    double s = 0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) s += buf[i];
    return s;
}
__device__ __forceinline__ bool wait_count(unsigned *cnt, unsigned want) {
    // (bounded: a hand-over that never comes must not hang the device)
    for (int spin = 0; spin < 2000000; ++spin) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return true; }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__global__ __launch_bounds__(1024) void k_prod(double *buf, const int *idx, int it) { produce(buf, idx, blockIdx.x, it); }
__global__ __launch_bounds__(1024) void k_cons(const double *buf, double *out, long w1, long w2) {
    double s = read_range(buf, 0, NA * PA) + read_range(buf, NA * PA, NA * PA + NB * PB);
    dyn[threadIdx.x] = s;
    __syncthreads();
    s = spin_ticks(w1, s); s = spin_ticks(w2, s);
    if (threadIdx.x == 0) out[0] = s + dyn[5];
}
__global__ __launch_bounds__(1024) void k_fused(double *buf, const int *idx, double *out, unsigned *cnt, int it, long w1, long w2) {
    if (blockIdx.x > 0) {
        const int b = blockIdx.x - 1;
        produce(buf, idx, b, it);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (b < NA ? 0 : 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned epoch = (unsigned)(it + 1);
    bool ok = wait_count(cnt, NA * epoch);
    double s = read_range(buf, 0, NA * PA);
    dyn[threadIdx.x] = s;
    __syncthreads();
    s = spin_ticks(w1, s);
    ok = wait_count(cnt + 1, NB * epoch) && ok;
    s += read_range(buf, NA * PA, NA * PA + NB * PB);
    dyn[threadIdx.x] += s;
    __syncthreads();
    s = spin_ticks(w2, s);
    if (threadIdx.x == 0) out[0] = ok ? s + dyn[5] : -1.0;
}

int main() {
    double *buf, *out; unsigned *cnt; int *idx;
    hipMalloc(&buf, (NA * PA + NB * PB) * 8); hipMalloc(&out, 8); hipMalloc(&cnt, 8); hipMemset(cnt, 0, 8);
    hipMalloc(&idx, 1024 * 4);
    { int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023; hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice); }
    hipFuncSetAttribute((const void *)k_cons, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    hipFuncSetAttribute((const void *)k_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    const size_t lds = 140 * 1024;
    // ticks of s_memtime (about 2.25 per ns in these kernels: DESIGN.md section 4): level 0 of the chain ~ 3 k ticks = 1.3 us, the rest of the solve ~ 50 k = 22 us
    for (long w1 : {0L, 3000L}) for (long w2 : {0L, 50000L}) {
        const int N = 2000;
        auto run2 = [&](int n) { for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_prod, dim3(NA + NB), dim3(1024), 0, 0, buf, idx, i); hipLaunchKernelGGL(k_cons, dim3(1), dim3(1024), lds, 0, buf, out, w1, w2); } };
        run2(50); hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now(); run2(N); hipDeviceSynchronize();
        const double two = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        hipMemset(cnt, 0, 8);
        int it = 0;
        auto run1 = [&](int n) { for (int i = 0; i < n; ++i, ++it) hipLaunchKernelGGL(k_fused, dim3(NA + NB + 1), dim3(1024), lds, 0, buf, idx, out, cnt, it, w1, w2); };
        run1(50); hipDeviceSynchronize();
        t0 = std::chrono::steady_clock::now(); run1(N); hipDeviceSynchronize();
        const double one = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        // the consumer alone (what the work costs without any producer)
        auto runc = [&](int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_cons, dim3(1), dim3(1024), lds, 0, buf, out, w1, w2); };
        runc(50); hipDeviceSynchronize();
        t0 = std::chrono::steady_clock::now(); runc(N); hipDeviceSynchronize();
        const double alone = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("work %ld + %ld ticks: consumer kernel alone %6.2f us | producers + consumer as two kernels %6.2f us | ONE kernel, two hand-overs %6.2f us  (%s)\n",
               w1, w2, alone, two, one, h < 0 ? "TIMED OUT" : "ok");
    }
    return 0;
}
