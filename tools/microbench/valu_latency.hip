// Diagnostic micro-benchmark (not part of the product): cycles per instruction of a gfx950 wave for the instruction
// kinds k_pose_solve's serial chain is made of.  Every test is one inline-asm block (s_memtime inside) so that the
// compiler cannot move anything across the stamps.
//   hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define T_BEGIN "s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n s_nop 7\n"
#define T_END(reg) "v_cmp_eq_f64 vcc, " reg ", " reg "\n s_mov_b32 s30, vcc_lo\n s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
#define T_END32(reg) "v_readfirstlane_b32 s30, " reg "\n s_nop 3\n s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
#define OUTS [t0] "=&s"(t0), [t1] "=&s"(t1)
#define RECORD(i) do { if (lane == 0) out[wave * 32 + (i)] = t1 - t0; } while (0)

__global__ __launch_bounds__(1024) void bench(unsigned long long *out, double *sink, int active_waves) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    double x = 1.0 + lane * 1e-3, y = 0.999, z = 0.5;
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    float f = 1.0f + lane, g0 = f, g1 = f + 1, g2 = f + 2, g3 = f + 3;
    int q = lane, xl = __double2loint(x), xh = __double2hiint(x);
    unsigned long long t0, t1;
    unsigned ldsaddr = 64;
    if (wave < active_waves) {
        asm volatile(T_BEGIN ".rept 64\n v_fma_f64 %[x], %[x], %[y], %[z]\n .endr\n" T_END("%[x]")
                     : OUTS, [x] "+v"(x) : [y] "v"(y), [z] "v"(z) : "s30", "vcc");
        RECORD(0);
        asm volatile(T_BEGIN ".rept 8\n v_fma_f64 %[a0], %[a0], %[y], %[z]\n v_fma_f64 %[a1], %[a1], %[y], %[z]\n v_fma_f64 %[a2], %[a2], %[y], %[z]\n v_fma_f64 %[a3], %[a3], %[y], %[z]\n"
                     "v_fma_f64 %[a4], %[a4], %[y], %[z]\n v_fma_f64 %[a5], %[a5], %[y], %[z]\n v_fma_f64 %[a6], %[a6], %[y], %[z]\n v_fma_f64 %[a7], %[a7], %[y], %[z]\n .endr\n" T_END("%[a7]")
                     : OUTS, [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7) : [y] "v"(y), [z] "v"(z) : "s30", "vcc");
        RECORD(1);
        // readlane pair + nop + fma with the SGPR pair as operand, independent accumulators
        asm volatile(T_BEGIN ".rept 16\n"
                     "v_readlane_b32 s32, %[xl], 1\n v_readlane_b32 s33, %[xh], 1\n s_nop 1\n v_fma_f64 %[a0], s[32:33], %[y], %[a0]\n"
                     "v_readlane_b32 s32, %[xl], 2\n v_readlane_b32 s33, %[xh], 2\n s_nop 1\n v_fma_f64 %[a1], s[32:33], %[y], %[a1]\n"
                     "v_readlane_b32 s32, %[xl], 3\n v_readlane_b32 s33, %[xh], 3\n s_nop 1\n v_fma_f64 %[a2], s[32:33], %[y], %[a2]\n"
                     "v_readlane_b32 s32, %[xl], 4\n v_readlane_b32 s33, %[xh], 4\n s_nop 1\n v_fma_f64 %[a3], s[32:33], %[y], %[a3]\n"
                     ".endr\n" T_END("%[a3]")
                     : OUTS, [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : [xl] "v"(xl), [xh] "v"(xh), [y] "v"(y) : "s30", "s32", "s33", "vcc");
        RECORD(2);   // 64 groups of (2 readlane + nop + fma)
        asm volatile(T_BEGIN ".rept 32\n v_readlane_b32 s32, %[xl], 1\n v_readlane_b32 s33, %[xh], 2\n .endr\n" T_END32("%[xl]")
                     : OUTS : [xl] "v"(xl), [xh] "v"(xh) : "s30", "s32", "s33");
        RECORD(3);   // 64 readlanes
        asm volatile(T_BEGIN ".rept 64\n v_mov_b32 %[q], %[q]\n .endr\n" T_END32("%[q]") : OUTS, [q] "+v"(q) : : "s30", "vcc");
        RECORD(4);
        asm volatile(T_BEGIN ".rept 64\n v_fma_f32 %[f], %[f], %[f], %[f]\n .endr\n" T_END32("%[f]") : OUTS, [f] "+v"(f) : : "s30", "vcc");
        RECORD(5);
        asm volatile(T_BEGIN ".rept 16\n v_fma_f32 %[g0], %[g0], %[g0], %[g0]\n v_fma_f32 %[g1], %[g1], %[g1], %[g1]\n v_fma_f32 %[g2], %[g2], %[g2], %[g2]\n v_fma_f32 %[g3], %[g3], %[g3], %[g3]\n .endr\n" T_END32("%[g3]")
                     : OUTS, [g0] "+v"(g0), [g1] "+v"(g1), [g2] "+v"(g2), [g3] "+v"(g3) : : "s30", "vcc");
        RECORD(6);
        asm volatile(T_BEGIN ".rept 64\n v_rcp_f64 %[x], %[x]\n .endr\n" T_END("%[x]") : OUTS, [x] "+v"(x) : : "s30", "vcc");
        RECORD(7);
        asm volatile(T_BEGIN ".rept 16\n v_rcp_f64 %[a0], %[a0]\n v_rcp_f64 %[a1], %[a1]\n v_rcp_f64 %[a2], %[a2]\n v_rcp_f64 %[a3], %[a3]\n .endr\n" T_END("%[a3]")
                     : OUTS, [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3) : : "s30", "vcc");
        RECORD(8);
        asm volatile(T_BEGIN ".rept 64\n v_mul_f64 %[x], %[x], %[y]\n .endr\n" T_END("%[x]") : OUTS, [x] "+v"(x) : [y] "v"(y) : "s30", "vcc");
        RECORD(9);
        // 64 uniform ds_read_b64 back to back, one wait at the end
        asm volatile(T_BEGIN ".rept 8\n ds_read_b64 %[a0], %[ad]\n ds_read_b64 %[a1], %[ad] offset:8\n ds_read_b64 %[a2], %[ad] offset:16\n ds_read_b64 %[a3], %[ad] offset:24\n"
                     "ds_read_b64 %[a4], %[ad] offset:32\n ds_read_b64 %[a5], %[ad] offset:40\n ds_read_b64 %[a6], %[ad] offset:48\n ds_read_b64 %[a7], %[ad] offset:56\n .endr\n s_waitcnt lgkmcnt(0)\n" T_END("%[a7]")
                     : OUTS, [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7) : [ad] "v"(ldsaddr) : "s30", "vcc", "memory");
        RECORD(10);
        // one ds_read_b64 + wait: round-trip latency, 16 times
        asm volatile(T_BEGIN ".rept 16\n ds_read_b64 %[a0], %[ad]\n s_waitcnt lgkmcnt(0)\n .endr\n" T_END("%[a0]")
                     : OUTS, [a0] "=&v"(a0) : [ad] "v"(ldsaddr) : "s30", "vcc", "memory");
        RECORD(11);
        // ds_write_b64 then ds_read_b64 of the same address + wait, 16 times
        asm volatile(T_BEGIN ".rept 16\n ds_write_b64 %[ad], %[a1]\n ds_read_b64 %[a0], %[ad]\n s_waitcnt lgkmcnt(0)\n .endr\n" T_END("%[a0]")
                     : OUTS, [a0] "=&v"(a0) : [ad] "v"(ldsaddr), [a1] "v"(a1) : "s30", "vcc", "memory");
        RECORD(12);
        asm volatile(T_BEGIN ".rept 32\n v_cndmask_b32 %[q], %[q], %[q], vcc\n v_cndmask_b32 %[g0], %[g0], %[g0], vcc\n .endr\n" T_END32("%[q]") : OUTS, [q] "+v"(q), [g0] "+v"(g0) : : "s30", "vcc");
        RECORD(13);
        asm volatile(T_BEGIN ".rept 64\n s_add_u32 s32, s32, 1\n .endr\n" T_END32("%[q]") : OUTS : [q] "v"(q) : "s30", "s32", "scc");
        RECORD(14);
        asm volatile(T_BEGIN ".rept 64\n s_nop 0\n .endr\n" T_END32("%[q]") : OUTS : [q] "v"(q) : "s30", "vcc");
        RECORD(15);
        // alternating dependent fma64 with independent v_mov (does other work fit in the fma's shadow?)
        asm volatile(T_BEGIN ".rept 64\n v_fma_f64 %[x], %[x], %[y], %[z]\n v_mov_b32 %[q], %[q]\n .endr\n" T_END("%[x]")
                     : OUTS, [x] "+v"(x), [q] "+v"(q) : [y] "v"(y), [z] "v"(z) : "s30", "vcc");
        RECORD(16);
        asm volatile(T_BEGIN ".rept 64\n v_fma_f64 %[x], %[x], %[y], %[z]\n v_mov_b32 %[q], %[q]\n v_mov_b32 %[g0], %[g0]\n v_mov_b32 %[g1], %[g1]\n .endr\n" T_END("%[x]")
                     : OUTS, [x] "+v"(x), [q] "+v"(q), [g0] "+v"(g0), [g1] "+v"(g1) : [y] "v"(y), [z] "v"(z) : "s30", "vcc");
        RECORD(17);
        sink[threadIdx.x] = x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q + f + g0 + g1 + g2 + g3;
    }
    __syncthreads();
}

int main() {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 16 * 32 * 8); hipMalloc(&sink, 1024 * 8);
    const int NT = 18;
    const char *names[NT] = {"64 dependent v_fma_f64", "64 v_fma_f64, 8 independent chains", "64 x (2 readlane + s_nop 1 + fma64 sgpr)", "64 v_readlane_b32",
                             "64 dependent v_mov_b32", "64 dependent v_fma_f32", "64 v_fma_f32, 4 chains", "64 dependent v_rcp_f64", "64 v_rcp_f64, 4 chains",
                             "64 dependent v_mul_f64", "64 uniform ds_read_b64 + 1 wait", "16 x (ds_read_b64 + wait)", "16 x (ds_write + ds_read + wait)",
                             "64 v_cndmask_b32 (2 chains)", "64 dependent s_add_u32", "64 s_nop 0", "64 x (dep fma64 + 1 v_mov)", "64 x (dep fma64 + 3 v_mov)"};
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int threads = cfg == 0 ? 64 : 1024, act = cfg == 0 ? 1 : (cfg == 1 ? 1 : (cfg == 2 ? 4 : 16));
        hipMemset(out, 0, 16 * 32 * 8);
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(bench, dim3(1), dim3(threads), 4096 * 8, 0, out, sink, act);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
        printf("workgroup of %d threads, %d active waves: ticks, wave 0 | last active wave\n", threads, act);
        for (int i = 0; i < NT; ++i) printf("  %-44s %6llu | %6llu\n", names[i], h[i], h[(act - 1) * 32 + i]);
    }
    return 0;
}
