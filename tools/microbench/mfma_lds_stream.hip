// Diagnostic micro-benchmark: the operand-streaming loop of k_linearize's phase 2 (a chain of v_mfma_f64_16x16x4_f64 whose A / B
// operands come from LDS, one element per lane per product, the next chunk's loads issued before this chunk's products) run by a
// chosen set of waves of a 1024-thread workgroup.  How many cycles per product, alone and in company?
//   hipcc --offload-arch=gfx950 -O3 mfma_lds_stream.hip -o mfma_lds_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define RROW 26
#define G 82
#define PLANE (G * RROW + 6)
#ifndef GRID
#define GRID 1
#endif

template <int MODE>   // 0: two streams (A and B), 1: one stream used for both operands, 2: no LDS at all (register operands),
                      // 3: two streams whose step stride is a compile-time constant (immediate offsets in the loads: no VALU in the loop but the two pointer bumps)
__global__ __launch_bounds__(1024) void k(unsigned long long *out, double *sink, unsigned mask, int reps, int rrow, int gg, int boff) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cl = lane & 15, rg = lane >> 4;
    for (int e = threadIdx.x; e < 4 * PLANE + 64; e += 1024) lds[e] = 1.0 + 1e-3 * (e % 97);
    __syncthreads();
    if ((mask >> wave) & 1) {
        const int kpl = wave & 3;
        const int ca = cl, cac = ca < 12 ? ca : 11;
        const int oa = (cac < 6 ? cac : 12 + cac - 6) + (rg & 1) * 6;
        const double *plane = lds + kpl * PLANE + (rg >> 1) * RROW;
        const double *zero = lds + 4 * PLANE;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; ++r) {
            const double *pa = ca < 12 ? plane + oa : zero, *pb = ca < 12 ? plane + oa + boff : zero;      // (run-time strides, offsets and trip count: the compiler must not fold the two streams into one or unroll the chunks)
            const int sa = MODE == 3 ? 2 : (ca < 12 ? 2 * rrow : 0), sb = MODE == 3 ? 2 : (ca < 12 ? 2 * rrow : 0);
            const int chunks_full = (gg >> 1) >> 2;
            double va[4], vb[4], xa[4], xb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = (MODE == 0 || MODE == 3) ? pb[u * sb] : va[u]; }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            int ch = 0;
            for (; ch + 2 <= chunks_full; ch += 2) {
                pa += 4 * sa; pb += 4 * sb;
                if (MODE != 2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xa[u] = pa[u * sa]; xb[u] = (MODE == 0 || MODE == 3) ? pb[u * sb] : xa[u]; }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xa[u] = va[u] + 1.0; xb[u] = xa[u]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], vb[u], acc, 0, 0, 0);
                const int nx = ch + 2 < chunks_full ? 4 : 0;
                pa += nx * sa; pb += nx * sb;
                __builtin_amdgcn_sched_barrier(0);
                if (MODE != 2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = (MODE == 0 || MODE == 3) ? pb[u * sb] : va[u]; }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { va[u] = xa[u] + 1.0; vb[u] = va[u]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], xb[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" : "+v"(acc));
        const double rs = acc[0] + acc[1] + acc[2] + acc[3];
        const int rl = __builtin_amdgcn_readfirstlane(__double2loint(rs));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0 + (rl == 12345 ? 1 : 0);
        sink[threadIdx.x] = rs;
    }
    __syncthreads();
}

template <int MODE>
void run(const char *name, unsigned long long *out, double *sink) {
    const unsigned masks[] = {0x1, 0xF, 0x7F, 0x11, 0x1111, 0xFFFF};
    const int reps = 4, products = reps * 40;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (unsigned m : masks) {
        hipMemset(out, 0, 16 * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(GRID), dim3(1024), (4 * PLANE + 64) * 8, 0, out, sink, m, reps, RROW, G, 0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
        printf("%-28s waves 0x%04x: cycles per product:", name, m);
        for (int w = 0; w < 16; ++w) if ((m >> w) & 1) printf(" %llu", h[w] / products);
        printf("\n");
    }
}

int main() {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 1024 * 8);
    run<0>("A and B streams from LDS", out, sink);
    run<1>("one stream from LDS", out, sink);
    run<2>("operands in registers", out, sink);
    run<3>("A and B, constant stride", out, sink);
    return 0;
}
