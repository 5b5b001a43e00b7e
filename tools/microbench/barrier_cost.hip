// Diagnostic micro-benchmark (not part of the product): what one __syncthreads() costs a 1024-thread workgroup on
// gfx950, alone and with the LDS write -> barrier -> broadcast read -> 16 FMAs round of k_pose_solve's back-substitution.
//   hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void bench(unsigned long long *out, double *sink, int nthreads_active) {
    __shared__ double sx[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) sx[tid] = 1.0 + tid * 1e-3;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[0] = (t1 - t0) / 64;
    // barrier with an s_waitcnt only (no fence semantics beyond LDS)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[1] = (t1 - t0) / 64;
    // one round: wave (i mod 16) writes 16 values, barrier, everybody reads them (broadcast) and runs 16 dependent-pair FMAs
    double acc = lane, a2 = 0.0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        if (wave == (i & 15) && lane < 16) sx[(i & 15) * 16 + lane] = acc * 1e-3;
        __syncthreads();
        const double *x = sx + (i & 15) * 16;
#pragma unroll
        for (int r = 0; r < 16; r += 2) { acc = fma(acc, 1e-9, x[r]); a2 = fma(a2, 1e-9, x[r + 1]); }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[2] = (t1 - t0) / 64;
    // (b) the same, lanes 0..15 of every wave only
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        if (wave == (i & 15) && lane < 16) sx[(i & 15) * 16 + lane] = acc * 1e-3;
        __syncthreads();
        if (lane < 16) {
            const double *x = sx + (i & 15) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { acc = fma(acc, 1e-9, x[r]); a2 = fma(a2, 1e-9, x[r + 1]); }
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[3] = (t1 - t0) / 64;
    // (c) one reader wave: the next writer
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        if (wave == (i & 15) && lane < 16) sx[(i & 15) * 16 + lane] = acc * 1e-3;
        __syncthreads();
        if (lane < 16 && wave == ((i + 1) & 15)) {
            const double *x = sx + (i & 15) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 2) { acc = fma(acc, 1e-9, x[r]); a2 = fma(a2, 1e-9, x[r + 1]); }
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[4] = (t1 - t0) / 64;
    // (d) as (c) with one read instead of eight and one FMA
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        if (wave == (i & 15) && lane < 16) sx[(i & 15) * 16 + lane] = acc * 1e-3;
        __syncthreads();
        if (lane < 16 && wave == ((i + 1) & 15)) acc = fma(acc, 1e-9, sx[(i & 15) * 16]);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[5] = (t1 - t0) / 64;
    // (e) 16 dependent FMAs in two chains, registers only; (f) 16 v_fmac_f64_dpp row_newbcast in two chains
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) { acc = fma(acc, 1e-9, a2); a2 = fma(a2, 1e-9, acc); }
        asm volatile("" : "+v"(acc), "+v"(a2));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[6] = (t1 - t0) / 64;
    double m = 1e-9;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        asm volatile("s_nop 1\n\t"
                     ".rept 8\n\t"
                     "v_fmac_f64_dpp %0, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                     ".endr" : "+v"(acc), "+v"(a2) : "v"(m), "v"(m));
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[7] = (t1 - t0) / 64;
    sink[tid] = acc + a2;
}

int main() {
    unsigned long long *d, h[8];
    double *sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 1024 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(bench, dim3(1), dim3(1024), 0, 0, d, sink, 1024);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("1024 threads: __syncthreads %llu cycles | s_waitcnt+s_barrier %llu | write + barrier + 8 broadcast reads + 16 FMAs: all lanes %llu, lanes<16 %llu, one wave %llu, one wave 1 read + 1 FMA %llu | 16 FMAs (2 chains) %llu | 16 DPP fmacs (2 chains) %llu\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
