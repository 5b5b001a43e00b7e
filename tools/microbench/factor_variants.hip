// Diagnostic micro-benchmark (not part of the product): variants of ch_factor's pivot step, one wave alone on its CU.
// A lone wave issues one instruction per 4 cycles, so a pivot costs its instruction count: what each of these is worth —
//   bit 0 (1)  the d == 0 guard as a scalar test + a branch to a rare block instead of v_cmp + two v_cndmask on every pivot
//              (the compiler answers with AGPR copies around every branch: 449 -> 588 instructions; kept for the record)
//   bit 4 (16) the guard as ONE select on the high word of the raw reciprocal (rcp(+-0) = +-inf: its low word is 0 already)
//   bit 5 (32) the explicit wait in front of the pivot's first update leaves the LAST refill group's loads in flight
//              (lgkmcnt(n) instead of lgkmcnt(0): those values are not needed before the end of the next trailing phase)
//   bit 6 (64) NO guard in the pivot loop: the wave publishes into a scratch tile, so the input tile stays as it was; behind the last pivot
//              the pivots are tested once and a zero one sends the wave through the guarded routine again (the same result as today in
//              every case: the two differ only when a pivot is exactly zero)
//   bit 1 (2)  one Newton step on v_rcp_f64 instead of the cubic step (the quotient's correction step absorbs the difference)
//   bit 2 (4)  rows publish in place (tile[row][j+1], stride 1 like the identity rows' M[row][j+1]: immediate offsets, no pointer add);
//              the pivot row is then read as a column (ds_read2_b64 takes two arbitrary offsets)
//   bit 3 (8)  no explicit lgkmcnt(0) (the compiler's own waits only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/microbench/factor_variants.hip -o tools/microbench/factor_variants
#include "../../visual-inertial-odometry_amd/csrc/vio_kernels.hip"
#include <cstdio>
#include <vector>
#include <cstring>

template <int VAR>
__device__ __forceinline__ double v_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    if (VAR & 16) { const int hi = (d == 0.0) ? 0 : __double2hiint(x); x = __hiloint2double(hi, __double2loint(x)); }
    const double e = fma(-d, x, 1.0);
    if (VAR & 2) return fma(x, e, x);
    const double t = fma(e, e, e);
    return fma(x, t, x);
}

template <int NP, int TS, int MS, bool RIDE, int VAR>
__device__ __noinline__ void ch_factor_v(lds_double *tile_in, lds_double *sI, lds_double *M, int lane, lds_double *rtile = nullptr, lds_double *scratch = nullptr,
                                        lds_double *pub = nullptr) {
    lds_double *tile = (VAR & (64 | 128)) ? pub : tile_in;
    asm volatile("" : "+v"(tile));
    const bool ident = (lane >> 4) == 1;
    const bool ride = RIDE && (lane >> 4) == 2;
    const int row = min(lane & 15, NP - 1);
    lds_double *p0 = (ident ? sI : (ride ? rtile : tile_in)) + row * TS;
    double a0[NP], u[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) a0[j] = p0[j];
    constexpr bool INPLACE = (VAR & 4) != 0;
    lds_double *wp = ident ? M + row * MS : (ride ? scratch + lane : (INPLACE ? tile + row * TS : tile + row));
    const int ws = ident ? 1 : (ride ? 0 : TS);
    __builtin_amdgcn_sched_barrier(0);
    wp[0] = a0[0];
    double d = d_readlane(a0[0], 0);
#pragma unroll
    for (int c = 1; c < NP; ++c) u[c] = INPLACE ? tile[c * TS] : tile[c];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        double l0;
        if (VAR & 1) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
            if (__builtin_expect((bits << 1) == 0ull, 0)) l0 = 0.0;
            else l0 = d_div(a0[j], d, v_rcp<VAR>(d));
        } else {
            const double x = (VAR & (16 | 64)) ? v_rcp<VAR>(d) : ((d == 0.0) ? 0.0 : v_rcp<VAR>(d));
            l0 = d_div(a0[j], d, x);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (VAR & 32) {
            // loads in flight: the refills of the previous trailing phase, n = NP - j - 1 values (columns j+1 .. NP-1) in groups of 4, two per
            // instruction; the last group's instructions stay in flight
            constexpr int dummy = 0; (void)dummy;
            const int n = NP - j - 1;                       // values refilled during pivot j - 1 (j >= 1); pivot 0: the NP - 1 initial loads
            const int lastg = n <= 0 ? 0 : ((n - 1) & 3) + 1;
            const int keep = (n > 4) ? (lastg + 1) / 2 : 0;
            if (keep == 0) __builtin_amdgcn_s_waitcnt(0xC07F);
            else if (keep == 1) __builtin_amdgcn_s_waitcnt(0xC17F);
            else __builtin_amdgcn_s_waitcnt(0xC27F);
        } else if (!(VAR & 8)) __builtin_amdgcn_s_waitcnt(0xC07F);
        if (j + 1 < NP) {
            a0[j + 1] = fma(-l0, u[j + 1], a0[j + 1]);
            if (INPLACE) wp[j + 1] = a0[j + 1];
            else { wp += ws; wp[0] = a0[j + 1]; }
            d = d_readlane(a0[j + 1], j + 1);
        }
        if (RIDE) a0[j] = l0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = j + 2; c < NP; ++c) {
            a0[c] = fma(-l0, u[c], a0[c]);
            if (((c - j - 2) & 3) == 3 || c + 1 == NP) {
#pragma unroll
                for (int e = c - ((c - j - 2) & 3); e <= c; ++e) u[e] = INPLACE ? tile[e * TS + (j + 1)] : tile[(j + 1) * TS + e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (RIDE && ride) {
        lds_double *lp = rtile + row * TS;
#pragma unroll
        for (int j = 0; j < NP; ++j) lp[j] = a0[j];
    }
}

template <int VAR, int NP, int TS, bool RIDE>
__global__ __launch_bounds__(64) void bench(const double *in, unsigned long long *out, double *res) {
    __shared__ double tile[16 * 17], sI[16 * 17], sM[16 * 17], rt[16 * 17], scr[96], pub[16 * 17];
    const int lane = threadIdx.x;
    for (int rep = 0; rep < 4; ++rep) {
        for (int i = lane; i < 16 * 17; i += 64) { tile[i] = 0.0; sM[i] = 0.0; sI[i] = 0.0; rt[i] = 0.25 + 0.001 * i; }
        __syncthreads();
        for (int i = lane; i < NP * NP; i += 64) { tile[(i / NP) * TS + i % NP] = in[i]; }
        if (lane < NP) sI[lane * TS + lane] = 1.0;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (VAR < 0) {
            if (RIDE) ch_factor<NP, TS, TS, true>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane, (lds_double *)rt, (lds_double *)scr);
            else ch_factor<NP, TS, TS>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane);
        } else {
            ch_factor_v<NP, TS, TS, RIDE, (VAR < 0 ? 0 : VAR)>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane, (lds_double *)rt, (lds_double *)scr, (lds_double *)pub);
            if (VAR >= 0 && (VAR & 64)) {
                const double pv = pub[min(lane, NP - 1) * (TS + 1)];
                if (__builtin_expect(__ballot(pv == 0.0) != 0ull, 0))
                    ch_factor_v<NP, TS, TS, RIDE, (VAR < 0 ? 0 : (VAR & ~64) | 128)>((lds_double *)tile, (lds_double *)sI, (lds_double *)sM, lane, (lds_double *)rt, (lds_double *)scr, (lds_double *)pub);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (lane == 0) out[rep] = t1 - t0;
    }
    // what the callers read: the pivots, M, the riding tile
    for (int i = lane; i < 16 * 17; i += 64) { res[i] = (i / TS == i % TS) ? ((VAR >= 0 && (VAR & 64)) ? pub[i] : tile[i]) : 0.0; res[272 + i] = sM[i]; res[544 + i] = rt[i]; }
}

static std::vector<double> g_ref;
template <int VAR, int NP, int TS, bool RIDE>
void run(const std::vector<double> &A, const double *d_in_, unsigned long long *d_out, double *d_res, const char *name, bool is_ref) {
    double *d_in = (double *)d_in_;
    std::vector<double> B(NP * NP);
    for (int i = 0; i < NP; ++i) for (int j = 0; j < NP; ++j) B[i * NP + j] = A[i * 16 + j];
    hipMemcpy(d_in, B.data(), NP * NP * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((bench<VAR, NP, TS, RIDE>), dim3(1), dim3(64), 0, 0, d_in, d_out, d_res);
    unsigned long long t[4];
    hipMemcpy(t, d_out, 32, hipMemcpyDeviceToHost);
    std::vector<double> r(816);
    hipMemcpy(r.data(), d_res, 816 * 8, hipMemcpyDeviceToHost);
    int ndiff = 0; double md = 0;
    if (is_ref) g_ref = r;
    else for (int i = 0; i < 816; ++i) { if (memcmp(&r[i], &g_ref[i], 8)) ++ndiff; md = fmax(md, fabs(r[i] - g_ref[i])); }
    printf("%-40s NP=%2d ride=%d  %5llu %5llu %5llu %5llu ticks (%.0f per pivot)  differing doubles %d (max %.3g)\n", name, NP, (int)RIDE, t[0], t[1], t[2], t[3],
           (double)t[3] / NP, ndiff, md);
}

#define RUNSET(NP, TS, RIDE) \
    run<-1, NP, TS, RIDE>(A, d_in, d_out, d_res, "product ch_factor", true); \
    run<0, NP, TS, RIDE>(A, d_in, d_out, d_res, "variant 0 (= product)", false); \
    run<1, NP, TS, RIDE>(A, d_in, d_out, d_res, "1 scalar guard", false); \
    run<2, NP, TS, RIDE>(A, d_in, d_out, d_res, "2 one Newton step", false); \
    run<4, NP, TS, RIDE>(A, d_in, d_out, d_res, "4 in-place publish", false); \
    run<8, NP, TS, RIDE>(A, d_in, d_out, d_res, "8 compiler's waits", false); \
    run<16, NP, TS, RIDE>(A, d_in, d_out, d_res, "16 high-word guard", false); \
    run<32, NP, TS, RIDE>(A, d_in, d_out, d_res, "32 last group in flight", false); \
    run<6, NP, TS, RIDE>(A, d_in, d_out, d_res, "2+4", false); \
    run<12, NP, TS, RIDE>(A, d_in, d_out, d_res, "4+8", false); \
    run<36, NP, TS, RIDE>(A, d_in, d_out, d_res, "4+32", false); \
    run<38, NP, TS, RIDE>(A, d_in, d_out, d_res, "2+4+32", false); \
    run<54, NP, TS, RIDE>(A, d_in, d_out, d_res, "2+4+16+32", false); \
    run<30, NP, TS, RIDE>(A, d_in, d_out, d_res, "2+4+8+16", false); \
    run<64, NP, TS, RIDE>(A, d_in, d_out, d_res, "64 no guard, tested behind F", false); \
    run<68, NP, TS, RIDE>(A, d_in, d_out, d_res, "4+64", false); \
    run<70, NP, TS, RIDE>(A, d_in, d_out, d_res, "2+4+64", false);

int main() {
    const int n = 16;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 40.0 + i : 1.0 / (1.0 + i + j));
    double *d_in, *d_res; unsigned long long *d_out;
    hipMalloc(&d_in, n * n * 8); hipMalloc(&d_out, 64); hipMalloc(&d_res, 816 * 8);
    RUNSET(16, 17, false)
    RUNSET(9, 11, true)
    RUNSET(9, 11, false)
    // a zero pivot (rows / columns 3 are zero): the guard's semantics
    for (int i = 0; i < n; ++i) { A[3 * n + i] = 0.0; A[i * n + 3] = 0.0; }
    printf("-- with a zero row/column 3 --\n");
    RUNSET(16, 17, false)
    RUNSET(9, 11, true)
    return 0;
}
