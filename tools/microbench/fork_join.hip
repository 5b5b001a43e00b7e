// Diagnostic micro-benchmark (not part of the product): what a fork / join between two HIP streams costs per iteration on gfx950,
// for the shape the GN loop would take with the lambda-known speed-bias elimination on a second stream:
//   stream A:  K1 (13 us, 256 workgroups) -> K2 (5 us, 100 workgroups) ---.
//   stream B:  Kc (16 us, 1 workgroup)  ----------------------------------+-> K3 (22 us, 1 workgroup, stream A) -> next iteration
// against the same kernels in ONE stream (K1 -> K2 -> Kc -> K3) and against K1 -> K2 -> K3' (30 us) as today.
// Kernels spin on s_memrealtime (100 MHz).   hipcc --offload-arch=gfx950 -O3 fork_join.hip -o fork_join
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_spin(int ticks, double *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) { }
    if (out && threadIdx.x == 0 && blockIdx.x == 1u << 30) out[0] = 1.0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    double *out; CK(hipMalloc(&out, 8));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const int N = 1000;
    hipEvent_t evA[2], evB[2];
    for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming)); }
    auto run = [&](int mode) -> double {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            if (mode == 0) {            // today: three kernels in one stream
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, A, 1300, out);
                hipLaunchKernelGGL(k_spin, dim3(100), dim3(1024), 0, A, 500, out);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, A, 3000, out);
            } else if (mode == 1) {     // four kernels in one stream (the split solve, no overlap)
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, A, 1300, out);
                hipLaunchKernelGGL(k_spin, dim3(100), dim3(1024), 0, A, 500, out);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, A, 1600, out);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, A, 2200, out);
            } else {                    // fork / join
                // B's chain kernel needs the states K3 of the previous iteration left: wait for A's event
                if (i > 0) hipStreamWaitEvent(B, evA[(i - 1) & 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, B, 1600, out);
                hipEventRecord(evB[i & 1], B);
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, A, 1300, out);
                hipLaunchKernelGGL(k_spin, dim3(100), dim3(1024), 0, A, 500, out);
                hipStreamWaitEvent(A, evB[i & 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(1024), 0, A, 2200, out);
                hipEventRecord(evA[i & 1], A);
            }
        }
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    };
    for (int rep = 0; rep < 3; ++rep) {
        const double a = run(0), b = run(1), c = run(2);
        printf("one stream, 3 kernels (13+5+30 us of spinning): %6.2f us / iteration;  one stream, 4 kernels (13+5+16+22): %6.2f;  fork/join (max(13+5, 16) + 22 = 40): %6.2f\n", a, b, c);
    }
    return 0;
}
