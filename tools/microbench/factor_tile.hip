// Diagnostic micro-benchmark (not part of the product): one wave factoring one diagonal block, alone on its CU — what the pivot
// chain of k_pose_solve / k_pose_solve_c costs per variant.  Includes the kernels file for the routines under test.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/microbench/factor_tile.hip -o tools/microbench/factor_tile
#include "../../visual-inertial-odometry_amd/csrc/vio_kernels.hip"
#include <cstdio>
#include <vector>

template <int V, int NP, int TS>
__global__ __launch_bounds__(64) void bench(const double *in, unsigned long long *out, double *res) {
    __shared__ double tile[16 * 17], sI[16 * 17], sM[16 * 17], rt[16 * 17], scr[64];
    const int lane = threadIdx.x;
    for (int rep = 0; rep < 4; ++rep) {
        for (int i = lane; i < 16 * 17; i += 64) { tile[i] = 0.0; sM[i] = 0.0; sI[i] = 0.0; rt[i] = 0.25 + 0.001 * i; }
        __syncthreads();
        for (int i = lane; i < NP * NP; i += 64) { tile[(i / NP) * TS + i % NP] = in[i]; }
        if (lane < NP) sI[lane * TS + lane] = 1.0;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (V == 0) ps_factor_diag((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane);
        if (V == 1) ch_factor<NP, TS, TS>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane);
        if (V == 3) ch_factor<NP, TS, TS, true>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane, (lds_double *)rt, (lds_double *)scr);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (lane == 0) out[rep] = t1 - t0;
    }
    for (int i = lane; i < 16 * 17; i += 64) { res[i] = tile[i]; res[272 + i] = sM[i]; }
}

int main() {
    const int n = 16;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 40.0 + i : 1.0 / (1.0 + i + j)) ;
    double *d_in, *d_res; unsigned long long *d_out;
    hipMalloc(&d_in, n * n * 8); hipMalloc(&d_out, 64); hipMalloc(&d_res, 544 * 8);
    std::vector<double> r0(544), r1(544);
    auto run = [&](auto kern, const char *name, int np, std::vector<double> *keep) {
        std::vector<double> B(np * np);
        for (int i = 0; i < np; ++i) for (int j = 0; j < np; ++j) B[i * np + j] = A[i * n + j];
        hipMemcpy(d_in, B.data(), np * np * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_in, d_out, d_res);
        unsigned long long t[4];
        hipMemcpy(t, d_out, 32, hipMemcpyDeviceToHost);
        if (keep) hipMemcpy(keep->data(), d_res, 544 * 8, hipMemcpyDeviceToHost);
        printf("%-44s %5llu %5llu %5llu %5llu ticks  (%.0f per pivot)\n", name, t[0], t[1], t[2], t[3], (double)t[3] / np);
    };
    run(bench<0, 16, 17>, "ps_factor_diag (round 3, reciprocal-multiply)", 16, nullptr);
    run(bench<1, 16, 17>, "ch_factor<16>  (true division, LDS)", 16, &r0);
    run(bench<3, 16, 17>, "ch_factor<16> with a riding tile", 16, &r1);
    double md = 0; for (int i = 0; i < 544; ++i) md = fmax(md, fabs(r0[i] - r1[i]));
    printf("   max |plain - riding| over tile and M: %g\n", md);
    run(bench<1, 9, 11>, "ch_factor<9>", 9, &r0);
    run(bench<3, 9, 11>, "ch_factor<9> with a riding tile", 9, &r1);
    md = 0; for (int i = 0; i < 544; ++i) md = fmax(md, fabs(r0[i] - r1[i]));
    printf("   max |plain - riding| over tile and M: %g\n", md);
    run(bench<1, 8, 17>, "ch_factor<8>", 8, nullptr);
    run(bench<1, 2, 17>, "ch_factor<2>", 2, nullptr);
    return 0;
}
