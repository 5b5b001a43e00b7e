// host_dense.cpp — see host_dense.h.  Plain C++17, no third-party linear algebra.
#include "host_dense.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef VIO_TAIL_TIMING
#include <chrono>
#include <cstdio>
#endif

namespace vio_host {

// LU with partial pivoting, then solve against the identity: the route Eigen takes for a fixed 15x15
// `covariance.inverse()` (LU/PartialPivLU.h, LU/InverseImpl.h).
static void inverse15_narrow(const double *cov, double *info) {
    constexpr int n = 15;
    double lu[n * n];
    int piv[n];
    std::memcpy(lu, cov, sizeof(lu));
    for (int k = 0; k < n; ++k) {
        int row = k;
        double big = std::fabs(lu[n * k + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(lu[n * i + k]) > big) { big = std::fabs(lu[n * i + k]); row = i; }
        piv[k] = row;
        if (big != 0) {
            if (row != k)
                for (int j = 0; j < n; ++j) std::swap(lu[n * k + j], lu[n * row + j]);
            for (int i = k + 1; i < n; ++i) lu[n * i + k] /= lu[n * k + k];
        }
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) lu[n * i + j] -= lu[n * i + k] * lu[n * k + j];
    }
    for (int c = 0; c < n; ++c) {
        double x[n];
        for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) std::swap(x[k], x[piv[k]]);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) x[i] -= lu[n * i + j] * x[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) x[i] -= lu[n * i + j] * x[j];
            x[i] /= lu[n * i + i];
        }
        for (int i = 0; i < n; ++i) info[n * i + c] = x[i];
    }
}

// sqrt(a^2 + b^2) of the QL iteration's plane rotations.  std::hypot guards against overflow of the squares and is correctly rounded to under an
// ulp at the price of ~75 cycles on the iteration's one dependent chain (6 000 rotations for a 75-row block: 125 us, round 6's measurement);
// the matrices here have entries up to 1e16, far from where the squares overflow or vanish, so the plain form — within 1.5 ulp — is taken
// whenever both arguments are in [1e-150, 1e150] (and zero), std::hypot otherwise (non-finite input included).  Round 6; it moves the last
// bits of the eigenpairs against earlier rounds — the tolerances to the reference's own solver (another algorithm altogether) are unchanged.
static inline double vio_hypot(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)      // (inlined into routines built for AVX-512 hosts too: a * a + b * b stays a product and a sum everywhere)
#endif
    const double aa = std::fabs(a), bb = std::fabs(b);
    const double mx = aa > bb ? aa : bb, mn = aa > bb ? bb : aa;
    if (mx < 1e150 && (mn > 1e-150 || mn == 0.0) && mx > 1e-150) return std::sqrt(a * a + b * b);
    return std::hypot(a, b);
}

// Householder tridiagonalisation followed by the implicit-shift QL iteration.  The work matrix is kept column-major
// (at(i, j) = V[j n + i]): every inner loop of the two phases walks down a column.
// (compiled twice, AVX2 and baseline, dispatched at load time: same operations in the same order, wider vectors;
//  -DVIO_NO_TARGET_CLONES: one plain copy — ifunc resolvers run before ThreadSanitizer's runtime is up)
#if defined(VIO_NO_TARGET_CLONES) || defined(__HIP_DEVICE_COMPILE__)     // (hipcc's device pass of this host-only file knows no multiversioning)
#define VIO_CLONES
#else
#define VIO_CLONES __attribute__((target_clones("avx2", "default")))
#endif

#if !defined(VIO_NO_TARGET_CLONES) && !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {
struct Rot { int i; double c, s; };
struct QlJob {
    double *V = nullptr;
    int n = 0;
    Rot *rots = nullptr;
    size_t cap = 0;
    double *d = nullptr, *e = nullptr;
    std::atomic<size_t> avail{0};           // rotations recorded so far (release-published by the producer)
    std::atomic<bool> done{false}, overflow{false};
    bool ok = true;
};
constexpr int QL_MAXB = 96;                 // rows of a block (the carried column lives on the stack)

#if !defined(VIO_NO_TARGET_CLONES) && !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define VIO_QL_AVX512 1
// The same loop on a host with AVX-512 (the MI355X boxes' EPYC 9575F): the carried column — up to 96 rows — stays in NV of the 32 vector
// registers instead of on the stack, so a rotation is one load and one store per eight rows where the 256-bit form has two of each per four.
// The same operations on every element in the same order (two products and a sum, two products and a difference: no contraction), so the
// eigenvectors are the bits the other clones give.  The last vector is masked: a column's tail must not touch the next column.
// (no contraction of a product with the sum that follows it: the other clones have no fused instruction to contract into.  hipcc compiles
//  this file with -ffp-contract=fast, and the intrinsics' operations carry the flags of the header they come from — a pragma in the caller does
//  not reach them, and __arithmetic_fence does not hold the backend's contraction back either —, so every product passes through an empty
//  asm that pins it in a register: no instruction, and nothing to fuse with)
#if defined(__clang__)
#define VIO_512_ATTR __attribute__((target("avx512f")))
#else
#define VIO_512_ATTR __attribute__((target("avx512f"), optimize("fp-contract=off")))
#endif
VIO_512_ATTR static inline __m512d vio_mul512(__m512d a, __m512d b) {
    __m512d p = _mm512_mul_pd(a, b);
    asm("" : "+v"(p));
    return p;
}
#define VIO_MUL512(a, b) vio_mul512((a), (b))
bool have_avx512() {
    static const bool have = __builtin_cpu_supports("avx512f") && std::getenv("VIO_NO_AVX512") == nullptr;
    return have;
}
template <int NV>
VIO_512_ATTR void ql_apply_512(double *V, int n, int r0, int nb, const Rot *rots, size_t from, size_t to, double *carry_io, int &carry_col) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const __mmask8 tail = (nb & 7) ? (__mmask8)((1u << (nb & 7)) - 1u) : (__mmask8)0xff;
    __m512d cr[NV];
#pragma GCC unroll 16
    for (int v = 0; v < NV; ++v) cr[v] = _mm512_maskz_loadu_pd(v == NV - 1 ? tail : (__mmask8)0xff, carry_io + 8 * v);
    int cc = carry_col;
    for (size_t q = from; q < to; ++q) {
        const int i = rots[q].i;
        const __m512d vc = _mm512_set1_pd(rots[q].c), vs = _mm512_set1_pd(rots[q].s);
        const double *ci = V + (size_t)i * n + r0;
        double *ci1 = V + (size_t)(i + 1) * n + r0;
        if (cc != i + 1) {
            if (cc >= 0) {
                double *pc = V + (size_t)cc * n + r0;
#pragma GCC unroll 16
                for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(pc + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, cr[v]);
            }
#pragma GCC unroll 16
            for (int v = 0; v < NV; ++v) cr[v] = _mm512_maskz_loadu_pd(v == NV - 1 ? tail : (__mmask8)0xff, ci1 + 8 * v);
        }
#pragma GCC unroll 16
        for (int v = 0; v < NV; ++v) {
            const __mmask8 mk = v == NV - 1 ? tail : (__mmask8)0xff;
            const __m512d a = _mm512_maskz_loadu_pd(mk, ci + 8 * v), h = cr[v];
            const __m512d o = _mm512_add_pd(VIO_MUL512(vs, a), VIO_MUL512(vc, h));       // ci1[k] = s * a + c * h
            cr[v] = _mm512_sub_pd(VIO_MUL512(vc, a), VIO_MUL512(vs, h));                // carry[k] = c * a - s * h
            _mm512_mask_storeu_pd(ci1 + 8 * v, mk, o);
        }
        cc = i;
    }
#pragma GCC unroll 16
    for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(carry_io + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, cr[v]);
    carry_col = cc;
}
static bool ql_apply_wide(double *V, int n, int r0, int r1, const Rot *rots, size_t from, size_t to, double *carry_io, int &carry_col) {
    static const bool have = __builtin_cpu_supports("avx512f") && std::getenv("VIO_NO_AVX512") == nullptr;
    if (!have) return false;
    const int nb = r1 - r0;
    switch ((nb + 7) / 8) {
#define VIO_QL_CASE(NV) case NV: ql_apply_512<NV>(V, n, r0, nb, rots, from, to, carry_io, carry_col); return true;
        VIO_QL_CASE(1) VIO_QL_CASE(2) VIO_QL_CASE(3) VIO_QL_CASE(4) VIO_QL_CASE(5) VIO_QL_CASE(6)
        VIO_QL_CASE(7) VIO_QL_CASE(8) VIO_QL_CASE(9) VIO_QL_CASE(10) VIO_QL_CASE(11) VIO_QL_CASE(12)
#undef VIO_QL_CASE
        default: return false;
    }
}
#endif

// rotations [from, to) applied to rows [r0, r1) of V.  A sweep's rotations walk down the column pairs (i, i + 1), (i - 1, i), ...: the column a
// rotation leaves as `at(k, i)` is the next one's `at(k, i + 1)` and stays in `carry` (carry_col: which column it is, -1: none).
VIO_CLONES
void ql_apply_narrow(double *V, int n, int r0, int r1, const Rot *__restrict rots, size_t from, size_t to, double *__restrict carry_io, int &carry_col) {
    const int nb = r1 - r0;
    double carry[QL_MAXB];
    int cc = carry_col;
    for (int k = 0; k < nb; ++k) carry[k] = carry_io[k];
    for (size_t q = from; q < to; ++q) {
        const int i = rots[q].i;
        const double c = rots[q].c, s = rots[q].s;
        double *__restrict ci = V + (size_t)i * n + r0;
        double *__restrict ci1 = V + (size_t)(i + 1) * n + r0;
        if (cc != i + 1) {
            if (cc >= 0) { double *__restrict pc = V + (size_t)cc * n + r0; for (int k = 0; k < nb; ++k) pc[k] = carry[k]; }
            for (int k = 0; k < nb; ++k) carry[k] = ci1[k];
        }
        for (int k = 0; k < nb; ++k) {
            const double h = carry[k], a = ci[k];
            ci1[k] = s * a + c * h;
            carry[k] = c * a - s * h;
        }
        cc = i;
    }
    for (int k = 0; k < nb; ++k) carry_io[k] = carry[k];
    carry_col = cc;
}
void ql_apply(double *V, int n, int r0, int r1, const Rot *rots, size_t from, size_t to, double *carry_io, int &carry_col) {
#ifdef VIO_QL_AVX512
    if (r1 - r0 > 0 && ql_apply_wide(V, n, r0, r1, rots, from, to, carry_io, carry_col)) return;
#endif
    ql_apply_narrow(V, n, r0, r1, rots, from, to, carry_io, carry_col);
}
void ql_flush(double *V, int n, int r0, int r1, const double *carry, int &carry_col) {
    if (carry_col >= 0) { double *cc = V + (size_t)carry_col * n + r0; for (int k = 0; k < r1 - r0; ++k) cc[k] = carry[k]; }
    carry_col = -1;
}
// tql2 on (d, e) alone, recording its rotations (the statements of the loop this replaces, in their order, minus the update of V)
void ql_generate(QlJob &J) {
    const int n = J.n;
    double *d = J.d, *e = J.e;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::ldexp(1.0, -52);
    size_t nr = 0;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { J.ok = false; break; }
                if (nr + (size_t)(m - l) > J.cap) { J.overflow.store(true); J.avail.store(nr, std::memory_order_release); J.done.store(true, std::memory_order_release); return; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = vio_hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = vio_hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    J.rots[nr].i = i; J.rots[nr].c = c; J.rots[nr].s = s;
                    ++nr;
                }
                J.avail.store(nr, std::memory_order_release);
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    J.avail.store(nr, std::memory_order_release);
    J.done.store(true, std::memory_order_release);
}
#ifdef VIO_QL_AVX512
// One thread, an AVX-512 host, at most 96 rows: ql_generate's loop with every rotation applied where it is generated, the carried column in NV
// vector registers.  The iteration's scalar chain — a square root and two divisions a rotation, each waiting for the one before — leaves the
// vector pipes idle and the rotation's 6 n flops need nothing of the chain but (c, s): the core runs the two side by side (generation alone
// 57 us, application alone 50 at 75 rows on the EPYC 9575F; together, here, the longer of the two).  The 256-bit form of this fusion, with the
// carried column on the stack, lost (337 against 210 us, round 6); with the column in registers it wins.  The statements of ql_generate in
// their order, the element operations of ql_apply_512: the same bits.
template <int NV>
VIO_512_ATTR void ql_fused_512(QlJob &J) {
#if defined(__clang__)
#pragma clang fp contract(off)      // (the scalar chain: this function may use fused instructions, ql_generate's build may not)
#endif
    const int n = J.n;
    double *d = J.d, *e = J.e, *V = J.V;
    const __mmask8 tail = (n & 7) ? (__mmask8)((1u << (n & 7)) - 1u) : (__mmask8)0xff;
    __m512d cr[NV];
#pragma GCC unroll 16
    for (int v = 0; v < NV; ++v) cr[v] = _mm512_setzero_pd();
    int cc = -1;
    int pi = -1;                    // the rotation that still waits for its columns: (pi, pcs, psn)
    double pcs = 1.0, psn = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::ldexp(1.0, -52);
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { J.ok = false; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = vio_hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = vio_hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    // The rotation generated one step ago goes on the columns now (ql_apply_512's element operations), BEHIND this step's scalar
                    // chain in program order: the core's schedulers serve the oldest ready instruction first, so the chain — a square root and two
                    // divisions waiting for each other — keeps its pace and the 6 n vector flops of the previous rotation fill the pipes under
                    // its latencies (with the rotation applied in front of the next step's chain the two added up: 97 us at 75 rows against
                    // 57 + 50 apart).  The rotations reach V in the order they are generated, one step late.
                    asm volatile("" ::: "memory");           // (the compiler keeps the order too: the chain's statements above, the columns below)
                    if (pi >= 0) {
                        const double *ci = V + (size_t)pi * n;
                        double *ci1 = V + (size_t)(pi + 1) * n;
                        if (cc != pi + 1) {
                            if (cc >= 0) {
                                double *pc = V + (size_t)cc * n;
#pragma GCC unroll 16
                                for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(pc + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, cr[v]);
                            }
#pragma GCC unroll 16
                            for (int v = 0; v < NV; ++v) cr[v] = _mm512_maskz_loadu_pd(v == NV - 1 ? tail : (__mmask8)0xff, ci1 + 8 * v);
                        }
                        const __m512d vc = _mm512_set1_pd(pcs), vs = _mm512_set1_pd(psn);
#pragma GCC unroll 16
                        for (int v = 0; v < NV; ++v) {
                            const __mmask8 mk = v == NV - 1 ? tail : (__mmask8)0xff;
                            const __m512d a = _mm512_maskz_loadu_pd(mk, ci + 8 * v), hh = cr[v];
                            const __m512d o = _mm512_add_pd(VIO_MUL512(vs, a), VIO_MUL512(vc, hh));
                            cr[v] = _mm512_sub_pd(VIO_MUL512(vc, a), VIO_MUL512(vs, hh));
                            _mm512_mask_storeu_pd(ci1 + 8 * v, mk, o);
                        }
                        cc = pi;
                    }
                    pi = i; pcs = c; psn = s;
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    if (pi >= 0) {                  // the last rotation
        const double *ci = V + (size_t)pi * n;
        double *ci1 = V + (size_t)(pi + 1) * n;
        if (cc != pi + 1) {
            if (cc >= 0) {
                double *pc = V + (size_t)cc * n;
#pragma GCC unroll 16
                for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(pc + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, cr[v]);
            }
#pragma GCC unroll 16
            for (int v = 0; v < NV; ++v) cr[v] = _mm512_maskz_loadu_pd(v == NV - 1 ? tail : (__mmask8)0xff, ci1 + 8 * v);
        }
        const __m512d vc = _mm512_set1_pd(pcs), vs = _mm512_set1_pd(psn);
#pragma GCC unroll 16
        for (int v = 0; v < NV; ++v) {
            const __mmask8 mk = v == NV - 1 ? tail : (__mmask8)0xff;
            const __m512d a = _mm512_maskz_loadu_pd(mk, ci + 8 * v), hh = cr[v];
            const __m512d o = _mm512_add_pd(VIO_MUL512(vs, a), VIO_MUL512(vc, hh));
            cr[v] = _mm512_sub_pd(VIO_MUL512(vc, a), VIO_MUL512(vs, hh));
            _mm512_mask_storeu_pd(ci1 + 8 * v, mk, o);
        }
        cc = pi;
    }
    if (cc >= 0) {
        double *pc = V + (size_t)cc * n;
#pragma GCC unroll 16
        for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(pc + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, cr[v]);
    }
}
bool ql_fused_wide(QlJob &J) {
    if (!have_avx512() || J.n < 24 || J.n > 96 || std::getenv("VIO_NO_QL_FUSION") != nullptr) return false;
    switch ((J.n + 7) / 8) {
#define VIO_QF_CASE(NV) case NV: ql_fused_512<NV>(J); return true;
        VIO_QF_CASE(3) VIO_QF_CASE(4) VIO_QF_CASE(5) VIO_QF_CASE(6) VIO_QF_CASE(7) VIO_QF_CASE(8) VIO_QF_CASE(9) VIO_QF_CASE(10) VIO_QF_CASE(11) VIO_QF_CASE(12)
#undef VIO_QF_CASE
        default: return false;
    }
}
#endif
// participant i of nt: 0 generates, then applies to its (small) block; the others apply to theirs as the rotations arrive
void ql_participant(void *arg, int i, int nt) {
    QlJob &J = *(QlJob *)arg;
    const int n = J.n;
    if (nt <= 1) {
        // (one thread: the whole iteration first, then the rotations block by block — on the round's host 125 + 85 us for 75 rows where the
        //  routine replaced, which applied every rotation where it was generated, took 260; generating and applying in ONE loop with the
        //  carried column, so that the core could overlap the dependent hypot / division chain with the vector work, measured 337: dropped)
#ifdef VIO_QL_AVX512
        if (ql_fused_wide(J)) return;
#endif
        ql_generate(J);
        if (J.overflow.load()) return;
        const size_t total = J.avail.load(std::memory_order_acquire);
        double carry[QL_MAXB];
        for (int r0 = 0; r0 < n; r0 += QL_MAXB) {
            int cc = -1;
            const int r1 = std::min(n, r0 + QL_MAXB);
            for (int k = 0; k < r1 - r0; ++k) carry[k] = 0.0;
            ql_apply(J.V, n, r0, r1, J.rots, 0, total, carry, cc);
            ql_flush(J.V, n, r0, r1, carry, cc);
        }
        return;
    }
    // rows: the producer keeps a quarter of a consumer's share (it starts applying when the iteration is over)
    const int units = 4 * (nt - 1) + 1;
    auto cut = [&](int t) { return t <= 0 ? 0 : (t >= nt ? n : (int)(((int64_t)n * (1 + 4 * (t - 1)) / units + 3) & ~3)); };
    int r0 = std::min(n, cut(i)), r1 = std::min(n, cut(i + 1));
    if (i == nt - 1) r1 = n;
    if (i == 0) ql_generate(J);
    if (r1 <= r0) return;
    double carry[QL_MAXB];
    size_t seen = 0;
    std::vector<int> bcc((size_t)(r1 - r0 + QL_MAXB - 1) / QL_MAXB, -1);
    std::vector<double> bcarry((size_t)bcc.size() * QL_MAXB);
    for (;;) {
        const bool fin = J.done.load(std::memory_order_acquire);
        const size_t av = J.avail.load(std::memory_order_acquire);
        if (J.overflow.load(std::memory_order_relaxed)) return;
        if (av > seen) {
            int bi = 0;
            for (int b0 = r0; b0 < r1; b0 += QL_MAXB, ++bi) {
                const int b1 = std::min(r1, b0 + QL_MAXB);
                ql_apply(J.V, n, b0, b1, J.rots, seen, av, bcarry.data() + (size_t)bi * QL_MAXB, bcc[bi]);
            }
            seen = av;
        } else if (fin) break;
    }
    int bi = 0;
    for (int b0 = r0; b0 < r1; b0 += QL_MAXB, ++bi) ql_flush(J.V, n, b0, std::min(r1, b0 + QL_MAXB), bcarry.data() + (size_t)bi * QL_MAXB, bcc[bi]);
    (void)carry;
}
#ifdef VIO_QL_AVX512
// acc[c0 .. c0 + ncols) = sum over q ascending of x[q] Y[q][c] (Y: rows of ld doubles), the accumulators in NV vector registers for the
// whole sum: the additions of every entry in the order of the scalar loops this stands in for, one load per eight products instead of an
// accumulator round trip per term (AVX-512 hosts: see ql_apply_512)
template <int NV>
VIO_512_ATTR void axpy_rows_512(const double *x, const double *Y, int nq, int ld, int c0, int ncols, double *acc) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const __mmask8 tail = (ncols & 7) ? (__mmask8)((1u << (ncols & 7)) - 1u) : (__mmask8)0xff;
    __m512d a[NV];
#pragma GCC unroll 16
    for (int v = 0; v < NV; ++v) a[v] = _mm512_setzero_pd();
    for (int q = 0; q < nq; ++q) {
        const __m512d xq = _mm512_set1_pd(x[q]);
        const double *y = Y + (size_t)q * ld + c0;
#pragma GCC unroll 16
        for (int v = 0; v < NV; ++v) a[v] = _mm512_add_pd(a[v], VIO_MUL512(xq, _mm512_maskz_loadu_pd(v == NV - 1 ? tail : (__mmask8)0xff, y + 8 * v)));
    }
#pragma GCC unroll 16
    for (int v = 0; v < NV; ++v) _mm512_mask_storeu_pd(acc + c0 + 8 * v, v == NV - 1 ? tail : (__mmask8)0xff, a[v]);
}
// acc[0 .. n) = sum over q ascending of x[q] Y[q][.]; false: no AVX-512 here
bool axpy_rows_wide(const double *x, const double *Y, int nq, int ld, int n, double *acc) {
    if (!have_avx512()) return false;
    for (int c0 = 0; c0 < n; c0 += 96) {
        const int ncols = std::min(96, n - c0);
        switch ((ncols + 7) / 8) {
#define VIO_AX_CASE(NV) case NV: axpy_rows_512<NV>(x, Y, nq, ld, c0, ncols, acc); break;
            VIO_AX_CASE(1) VIO_AX_CASE(2) VIO_AX_CASE(3) VIO_AX_CASE(4) VIO_AX_CASE(5) VIO_AX_CASE(6)
            VIO_AX_CASE(7) VIO_AX_CASE(8) VIO_AX_CASE(9) VIO_AX_CASE(10) VIO_AX_CASE(11) VIO_AX_CASE(12)
#undef VIO_AX_CASE
        }
    }
    return true;
}
// r[0 .. len) -= g[.] * s   (a product, then a difference)
VIO_512_ATTR void sub_scaled_512(double *r, const double *g, double s, int len) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const __m512d sv = _mm512_set1_pd(s);
    int j = 0;
    for (; j + 8 <= len; j += 8) _mm512_storeu_pd(r + j, _mm512_sub_pd(_mm512_loadu_pd(r + j), VIO_MUL512(_mm512_loadu_pd(g + j), sv)));
    if (j < len) {
        const __mmask8 mk = (__mmask8)((1u << (len - j)) - 1u);
        _mm512_mask_storeu_pd(r + j, mk, _mm512_sub_pd(_mm512_maskz_loadu_pd(mk, r + j), VIO_MUL512(_mm512_maskz_loadu_pd(mk, g + j), sv)));
    }
}
// r[0 .. len) -= D[.] * er + E[.] * dr   (two products, their sum, a difference: tred2's rank-2 update of one row)
VIO_512_ATTR void rank2_row_512(double *r, const double *D, const double *E, double er, double dr, int len) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const __m512d ev = _mm512_set1_pd(er), dv = _mm512_set1_pd(dr);
    int j = 0;
    for (; j + 8 <= len; j += 8) {
        const __m512d t = _mm512_add_pd(VIO_MUL512(_mm512_loadu_pd(D + j), ev), VIO_MUL512(_mm512_loadu_pd(E + j), dv));
        _mm512_storeu_pd(r + j, _mm512_sub_pd(_mm512_loadu_pd(r + j), t));
    }
    if (j < len) {
        const __mmask8 mk = (__mmask8)((1u << (len - j)) - 1u);
        const __m512d t = _mm512_add_pd(VIO_MUL512(_mm512_maskz_loadu_pd(mk, D + j), ev), VIO_MUL512(_mm512_maskz_loadu_pd(mk, E + j), dv));
        _mm512_mask_storeu_pd(r + j, mk, _mm512_sub_pd(_mm512_maskz_loadu_pd(mk, r + j), t));
    }
}
// tred2 — both phases — with the work matrix ROW-major and FULL (both triangles kept: the rank-2 update of entry (k, j) and of (j, k) are sums of
// the same two products, so the two stay bit for bit equal) on a host with AVX-512:
//   * the symmetric product p = A v is the sum over the rows k ascending of d[k] A[k][.], eight columns a register — every p_j takes its terms in
//     the order the column-wise loops of symmetric_eigen give it (row part j' < j ascending, the diagonal, the column part k > j ascending);
//   * the rank-2 update and the accumulation's update go a row at a time; the accumulation's dot products of column i + 1 with the columns
//     j <= i are one sum over the rows, every column's additions over k ascending.
// The same value in every entry as the loops in symmetric_eigen; leaves V column-major as they do.  false: not on this host.
bool tred2_wide(int n, const double *Ain, double *d, double *e, double *V) {
    if (!have_avx512() || n < 24) return false;
    static thread_local std::vector<double> Rv, uv, gv;
    if (Rv.size() < (size_t)n * n) Rv.resize((size_t)n * n);
    if (uv.size() < (size_t)n + 8) { uv.resize((size_t)n + 8); gv.resize((size_t)n + 8); }
    double *R = Rv.data(), *u = uv.data(), *G = gv.data();
    auto at = [&](int i, int j) -> double & { return R[(size_t)i * n + j]; };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) { at(i, j) = Ain[(size_t)i * n + j]; at(j, i) = at(i, j); }
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; at(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            axpy_rows_wide(d, R, i, n, i, e);                               // e = A d over the active block
            for (int j = 0; j < i; ++j) at(j, i) = d[j];                    // (the Householder vector, for the second phase)
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int r = 0; r < i; ++r) rank2_row_512(R + (size_t)r * n, d, e, e[r], d[r], i);
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        at(n - 1, i) = at(i, i);
        at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) { u[k] = at(k, i + 1); d[k] = u[k] / h; }
            axpy_rows_wide(u, R, i + 1, n, i + 1, G);
            for (int k = 0; k <= i; ++k) sub_scaled_512(R + (size_t)k * n, G, d[k], i + 1);
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) V[(size_t)j * n + i] = R[(size_t)i * n + j];
    return true;
}
#endif
}  // namespace

// Householder tridiagonalisation followed by the implicit-shift QL iteration.  The work matrix is kept column-major
// (at(i, j) = V[j n + i]): every inner loop of the two phases walks down a column.
// (compiled twice, AVX2 and baseline, dispatched at load time: same operations in the same order, wider vectors)
VIO_CLONES
bool symmetric_eigen_legacy(int n, const double *Ain, double *d, double *Vout) {
    std::vector<double> Vv((size_t)n * n), ev(n);
    double *V = Vv.data(), *e = ev.data();
    auto at = [&](int i, int j) -> double & { return V[(size_t)j * n + i]; };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) { at(i, j) = Ain[(size_t)i * n + j]; at(j, i) = at(i, j); }
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; at(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                at(j, i) = f;
                g = e[j] + at(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += at(k, j) * d[k]; e[k] += at(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
                d[j] = at(i - 1, j);
                at(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        at(n - 1, i) = at(i, i);
        at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
                for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = at(n - 1, j); at(n - 1, j) = 0.0; }
    at(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::ldexp(1.0, -52);
    bool ok = true;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { ok = false; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = vio_hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = vio_hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = at(k, i + 1);
                        at(k, i + 1) = s * at(k, i) + c * h;
                        at(k, i) = c * at(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; ++j)
            if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(at(j, i), at(j, k));
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Vout[(size_t)i * n + j] = at(i, j);
    return ok;
}

VIO_CLONES
bool symmetric_eigen(int n, const double *Ain, double *d, double *Vout, const Par *par) {
#ifdef VIO_TAIL_TIMING
    static double et[6] = {0}; static int ecalls = 0;
    auto enow = [] { return std::chrono::steady_clock::now(); };
    auto E0 = enow();
#define ET(k) do { auto E1 = enow(); if (n >= 32) et[k] += std::chrono::duration<double, std::micro>(E1 - E0).count(); E0 = E1; } while (0)
#else
#define ET(k) do { } while (0)
#endif
    // (the work matrix: scratch of the calling thread, every entry written before it is read — a fresh vector was an allocation and n^2 zeros a call)
    static thread_local std::vector<double> Vv, ev;
    if (Vv.size() < (size_t)n * n) Vv.resize((size_t)n * n);
    if (ev.size() < (size_t)n) ev.resize((size_t)n);
    double *V = Vv.data(), *e = ev.data();
    auto at = [&](int i, int j) -> double & { return V[(size_t)j * n + i]; };
#ifdef VIO_QL_AVX512
    const bool accumulated = tred2_wide(n, Ain, d, e, V);          // (both phases, row-major, on an AVX-512 host: the same values)
#else
    const bool accumulated = false;
#endif
    if (!accumulated) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) { at(i, j) = Ain[(size_t)i * n + j]; at(j, i) = at(i, j); }
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    }
    ET(0);
    for (int i = n - 1; i > 0 && !accumulated; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; at(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            // (the symmetric product p = A v, one column at a time in the routine this replaces: a dependent chain of i - j additions per
            //  column.  Two columns side by side — column j + 1 starts from e[j + 1], which column j's first step has just completed —:
            //  every sum takes its terms in the order it always did, two chains are in flight instead of one)
            {
                int j = 0;
                for (; j + 1 < i; j += 2) {
                    const double f0 = d[j], f1 = d[j + 1];
                    const double *c0 = &at(0, j), *c1 = &at(0, j + 1);
                    at(j, i) = f0;
                    at(j + 1, i) = f1;
                    double g0 = e[j] + c0[j] * f0;
                    g0 += c0[j + 1] * d[j + 1];
                    e[j + 1] += c0[j + 1] * f0;
                    double g1 = e[j + 1] + c1[j + 1] * f1;
                    for (int k = j + 2; k <= i - 1; ++k) {
                        const double a0 = c0[k], a1 = c1[k], dk = d[k];
                        g0 += a0 * dk;
                        g1 += a1 * dk;
                        double ek = e[k];
                        ek += a0 * f0;
                        ek += a1 * f1;
                        e[k] = ek;
                    }
                    e[j] = g0;
                    e[j + 1] = g1;
                }
                for (; j < i; ++j) {
                    f = d[j];
                    at(j, i) = f;
                    g = e[j] + at(j, j) * f;
                    for (int k = j + 1; k <= i - 1; ++k) { g += at(k, j) * d[k]; e[k] += at(k, j) * f; }
                    e[j] = g;
                }
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
                d[j] = at(i - 1, j);
                at(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    ET(1);
    for (int i = 0; i < n - 1 && !accumulated; ++i) {
        at(n - 1, i) = at(i, i);
        at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
            // (the columns are independent of each other: four dot products side by side, each over k ascending as before)
            {
                const double *u = &at(0, i + 1);
                int j = 0;
                for (; j + 3 <= i; j += 4) {
                    double *c0 = &at(0, j), *c1 = &at(0, j + 1), *c2 = &at(0, j + 2), *c3 = &at(0, j + 3);
                    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
                    for (int k = 0; k <= i; ++k) { const double uk = u[k]; g0 += uk * c0[k]; g1 += uk * c1[k]; g2 += uk * c2[k]; g3 += uk * c3[k]; }
                    for (int k = 0; k <= i; ++k) { const double dk = d[k]; c0[k] -= g0 * dk; c1[k] -= g1 * dk; c2[k] -= g2 * dk; c3[k] -= g3 * dk; }
                }
                for (; j <= i; ++j) {
                    double g = 0.0;
                    for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
                    for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
                }
            }
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = at(n - 1, j); at(n - 1, j) = 0.0; }
    at(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    // The QL iteration on (d, e) does not read the eigenvector matrix: thread 0 runs it and RECORDS the plane rotations (column pair, c, s) in
    // the order it generates them; the rotations are applied to V by row blocks — a row sees the same rotations in the same order whoever
    // applies them, so the result does not depend on the number of threads — while the iteration is still running (pipeline: `avail`).
    ET(2);
    // (the record: 3 n^2 rotations — an iteration takes about 0.85 n^2 — in a buffer the thread keeps from call to call: a fresh, zeroed
    //  megabyte per call cost the tail tens of microseconds in page faults)
    const size_t cap = (size_t)3 * n * n + 64;
    static thread_local std::vector<Rot> rots;
    if (rots.size() < cap) rots.resize(cap);
    QlJob job;
    job.V = V; job.n = n; job.rots = rots.data(); job.cap = cap; job.d = d; job.e = e;
    const int want = (par && par->run_n && n >= 32) ? std::min(par->width, (n + 15) / 16 + 1) : 1;
    if (want > 1) par->run_n(par->ctx, want, ql_participant, &job);
    else ql_participant(&job, 0, 1);
    if (job.overflow.load()) return symmetric_eigen_legacy(n, Ain, d, Vout);      // (more than 3 n^2 rotations: the routine as it was takes over)
    bool ok = job.ok;
    ET(3);
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; ++j)
            if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(at(j, i), at(j, k));
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Vout[(size_t)i * n + j] = at(i, j);
    ET(4);
#ifdef VIO_TAIL_TIMING
    if (n >= 32 && ++ecalls % 50 == 0) std::fprintf(stderr, "[eigen timing n=%d, avg us] copy-in %.1f | tred2 reduce %.1f | accumulate %.1f | ql (generate + apply) %.1f | sort, copy-out %.1f | generate alone %.1f\n", n, et[0] / ecalls, et[1] / ecalls, et[2] / ecalls, et[3] / ecalls, et[4] / ecalls, 0.0);
#endif
    return ok;
}

#ifdef VIO_QL_AVX512
// inverse15 on an AVX-512 host: a row of the 15 x 15 work matrices is two vector registers (8 + 7 lanes).  The elimination updates a row at a
// time under the mask of the columns behind the pivot; the 15 right-hand sides of the identity are solved side by side — row i of X is the
// vector of x_i over the columns c.  Every element goes through the operations of inverse15_narrow in their order (a product, then a
// difference; the same quotients): the same bits, 3 us -> 0.4 us an IMU edge, ten edges a frame.
VIO_512_ATTR static void inverse15_512(const double *cov, double *info) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    constexpr int n = 15;
    alignas(64) double lu[n * 16];
    int piv[n];
    const __mmask8 m7 = 0x7f;
    for (int i = 0; i < n; ++i) {
        _mm512_store_pd(lu + 16 * i, _mm512_loadu_pd(cov + n * i));
        _mm512_store_pd(lu + 16 * i + 8, _mm512_maskz_loadu_pd(m7, cov + n * i + 8));
    }
    for (int k = 0; k < n; ++k) {
        int row = k;
        double big = std::fabs(lu[16 * k + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(lu[16 * i + k]) > big) { big = std::fabs(lu[16 * i + k]); row = i; }
        piv[k] = row;
        if (big != 0) {
            if (row != k) {
                const __m512d a0 = _mm512_load_pd(lu + 16 * k), a1 = _mm512_load_pd(lu + 16 * k + 8);
                _mm512_store_pd(lu + 16 * k, _mm512_load_pd(lu + 16 * row)); _mm512_store_pd(lu + 16 * k + 8, _mm512_load_pd(lu + 16 * row + 8));
                _mm512_store_pd(lu + 16 * row, a0); _mm512_store_pd(lu + 16 * row + 8, a1);
            }
            for (int i = k + 1; i < n; ++i) lu[16 * i + k] /= lu[16 * k + k];
        }
        // rows below the pivot, columns behind it: lu[i][j] -= lu[i][k] * lu[k][j]
        const unsigned cols = 0x7fffu & ~((2u << k) - 1u);               // bits k + 1 .. 14
        const __mmask8 c0 = (__mmask8)(cols & 0xffu), c1 = (__mmask8)(cols >> 8);
        const __m512d p0 = _mm512_load_pd(lu + 16 * k), p1 = _mm512_load_pd(lu + 16 * k + 8);
        for (int i = k + 1; i < n; ++i) {
            const __m512d f = _mm512_set1_pd(lu[16 * i + k]);
            if (c0) _mm512_mask_store_pd(lu + 16 * i, c0, _mm512_sub_pd(_mm512_load_pd(lu + 16 * i), VIO_MUL512(f, p0)));
            if (c1) _mm512_mask_store_pd(lu + 16 * i + 8, c1, _mm512_sub_pd(_mm512_load_pd(lu + 16 * i + 8), VIO_MUL512(f, p1)));
        }
    }
    // X = the identity's rows, swapped as the elimination swapped them; then L, then U, all 15 columns at once
    __m512d x0[n], x1[n];
    int perm[n];
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) if (piv[k] != k) std::swap(perm[k], perm[piv[k]]);
    for (int i = 0; i < n; ++i) {
        const int c = perm[i];                                           // row i of X is the unit vector e_c
        x0[i] = _mm512_maskz_mov_pd((__mmask8)(c < 8 ? (1u << c) : 0u), _mm512_set1_pd(1.0));
        x1[i] = _mm512_maskz_mov_pd((__mmask8)(c >= 8 ? (1u << (c - 8)) : 0u), _mm512_set1_pd(1.0));
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            const __m512d f = _mm512_set1_pd(lu[16 * i + j]);
            x0[i] = _mm512_sub_pd(x0[i], VIO_MUL512(f, x0[j]));
            x1[i] = _mm512_sub_pd(x1[i], VIO_MUL512(f, x1[j]));
        }
    for (int i = n - 1; i >= 0; --i) {
        for (int j = i + 1; j < n; ++j) {
            const __m512d f = _mm512_set1_pd(lu[16 * i + j]);
            x0[i] = _mm512_sub_pd(x0[i], VIO_MUL512(f, x0[j]));
            x1[i] = _mm512_sub_pd(x1[i], VIO_MUL512(f, x1[j]));
        }
        const __m512d dd = _mm512_set1_pd(lu[16 * i + i]);
        x0[i] = _mm512_div_pd(x0[i], dd);
        x1[i] = _mm512_div_pd(x1[i], dd);
    }
    for (int i = 0; i < n; ++i) {
        _mm512_storeu_pd(info + n * i, x0[i]);
        _mm512_mask_storeu_pd(info + n * i + 8, m7, x1[i]);
    }
}
#endif
void inverse15(const double *cov, double *info) {
#ifdef VIO_QL_AVX512
    if (have_avx512()) { inverse15_512(cov, info); return; }
#endif
    inverse15_narrow(cov, info);
}

// rows [0, count) in contiguous pieces of at least `grain` on the helper threads of `par` (all of them on the caller without any)
namespace {
template <typename F> struct RowsJob { F *f; int count; };
template <typename F> void rows_piece(void *arg, int i, int n) {
    RowsJob<F> &j = *(RowsJob<F> *)arg;
    const int a0 = (int)((int64_t)j.count * i / n), a1 = (int)((int64_t)j.count * (i + 1) / n);
    if (a1 > a0) (*j.f)(a0, a1);
}
template <typename F> void par_rows(const Par *par, int count, int grain, F &f) {
    const int want = (par && par->run_n) ? std::min(par->width, count / std::max(grain, 1)) : 1;
    if (want <= 1) { if (count > 0) f(0, count); return; }
    RowsJob<F> j{&f, count};
    par->run_n(par->ctx, want, rows_piece<F>, &j);
}
}  // namespace

// A scratch array of the calling thread that keeps its storage from call to call (every user writes what it reads: nothing relies on zeros; a
// fresh std::vector per call is an allocation and 45 KB of zero fill each, eight of them in one marginalize_tail)
namespace {
struct Scratch {
    std::vector<double> v;
    double *get(size_t n) { if (v.size() < n) v.resize(n); return v.data(); }
};
}  // namespace

// rows [a0, a1) of H_prior = (V S) V^T over the kept eigenpairs (problem.cc:775-778): entry (a, c) = sum over q ascending of VS[a][q] VKt[q][c], a
// whole row of c at a time with q outside — every entry's additions in the order of the dot product they replace, nl independent chains
// instead of one dependent one (a function of its own so that it exists in the AVX2 clone: a lambda inside marginalize_tail does not)
VIO_CLONES
void prior_product_rows(int a0, int a1, const double *__restrict VS, const double *__restrict VKt, int nk, int nl, const int *__restrict live,
                        double *__restrict Hout, int n2, double *__restrict acc) {
    for (int a = a0; a < a1; ++a) {
#ifdef VIO_QL_AVX512
        if (axpy_rows_wide(VS + (size_t)a * nk, VKt, nk, nl, nl, acc)) {
            double *__restrict ho = Hout + (size_t)live[a] * n2;
            for (int c = 0; c < nl; ++c) ho[live[c]] = std::fabs(acc[c]) > 1e-9 ? acc[c] : 0.0;     // problem.cc:778
            continue;
        }
#endif
        for (int c = 0; c < nl; ++c) acc[c] = 0.0;
        const double *__restrict x = VS + (size_t)a * nk;
        int q = 0;
        for (; q + 3 < nk; q += 4) {            // (four terms per pass over the accumulators, added one after the other: q ascending for every entry)
            const double x0 = x[q], x1 = x[q + 1], x2 = x[q + 2], x3 = x[q + 3];
            const double *__restrict y0 = VKt + (size_t)q * nl, *__restrict y1 = y0 + nl, *__restrict y2 = y1 + nl, *__restrict y3 = y2 + nl;
            for (int c = 0; c < nl; ++c) {
                double t = acc[c];
                t += x0 * y0[c];
                t += x1 * y1[c];
                t += x2 * y2[c];
                t += x3 * y3[c];
                acc[c] = t;
            }
        }
        for (; q < nk; ++q) {
            const double xq = x[q];
            const double *__restrict y = VKt + (size_t)q * nl;
            for (int c = 0; c < nl; ++c) acc[c] += xq * y[c];
        }
        double *__restrict ho = Hout + (size_t)live[a] * n2;
        for (int c = 0; c < nl; ++c) ho[live[c]] = std::fabs(acc[c]) > 1e-9 ? acc[c] : 0.0;     // problem.cc:778
    }
}

// The two moves of problem.cc:721-745 (speed-bias of the marginalised frame to the bottom, then its pose) as one index map:
// entry (i, j) of the reordered matrix is H[order[i]][order[j]] of the matrix that came in.
static void marg_order(int n, int frame, int *order) {
    int o1[171], o2[171];
    auto move = [n](int idx, int dim, int *o) {
        int q = 0;
        for (int i = 0; i < n; ++i) if (i < idx || i >= idx + dim) o[q++] = i;
        for (int i = idx; i < idx + dim; ++i) o[q++] = i;
    };
    move(12 + 15 * frame, 9, o1);       // larger index first: speed-bias, then pose
    move(6 + 15 * frame, 6, o2);
    for (int i = 0; i < n; ++i) order[i] = o1[o2[i]];
}

VIO_CLONES
int marginalize_tail(double *Hin, double *bin, int frame, double *Hout, double *bout, double *errout, double *jtout, const Par *par) {
    constexpr int n = 171, m2 = 15, n2 = n - m2;
    // Round 4: nothing of the 171 x 171 matrix is copied or permuted in memory (it was, twice: 0.14 of the tail's 0.44 ms on the build
    // host); the reordered matrix is read through `order`, and everything below works on the rows that are not exactly zero.  The sums
    // are the ones the full-size version formed for those rows, in the same order: the outputs are bit-identical.
#ifdef VIO_TAIL_TIMING
    static double tt[8] = {0}; static int tcalls = 0;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tus = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    auto T0 = tnow();
#define TT(k) do { auto T1 = tnow(); tt[k] += tus(T0, T1); T0 = T1; } while (0)
#else
#define TT(k) do { } while (0)
#endif
    int order[n];
    marg_order(n, frame, order);
    auto Hp_ = [&](int i, int j) -> double { return Hin[(size_t)order[i] * n + order[j]]; };
    const double eps = 1e-8;
    double Amm[m2 * m2], ev[m2], V[m2 * m2], Ainv[m2 * m2];
    for (int i = 0; i < m2; ++i)
        for (int j = 0; j < m2; ++j) Amm[i * m2 + j] = 0.5 * (Hp_(n2 + i, n2 + j) + Hp_(n2 + j, n2 + i));
    symmetric_eigen(m2, Amm, ev, V);
    double evinv[m2];                                             // (the quotient once per eigenvalue, not once per term: 3 375 divisions before)
    for (int k = 0; k < m2; ++k) evinv[k] = ev[k] > eps ? 1.0 / ev[k] : 0.0;
    for (int i = 0; i < m2; ++i)
        for (int j = 0; j < m2; ++j) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += V[i * m2 + k] * evinv[k] * V[j * m2 + k];
            Ainv[i * m2 + j] = s;
        }
    TT(0);
    // Arr - Arm Amm^+ Amr, brr - Arm Amm^+ bmm (problem.cc:758-762).  A row (column) of the kept block that is exactly zero all the way — a
    // frame neither the marginalised frame's landmarks nor the old prior reach: 117 of the 156 at tracks of four frames — has a zero row
    // of tempB and yields zeros whatever it is multiplied with: only the others are formed.
    int rowlive[n2], nr = 0;
    for (int i = 0; i < n2; ++i) {
        const double *hr = Hin + (size_t)order[i] * n;            // (the whole row, in whatever order: any non-zero?)
        int anyi = 0;                                             // (no early exit: a dead row is read to its end anyway, and the plain loop vectorises)
        for (int j = 0; j < n; ++j) anyi |= (hr[j] != 0.0);
        bool any = anyi != 0;
        for (int j = n2; j < n && !any; ++j) any = Hp_(j, i) != 0.0;      // the marginalised rows' entries in column i (Amr)
        if (any) rowlive[nr++] = i;
    }
    TT(1);
    static thread_local Scratch s_tempB, s_Hpc, s_Amr, s_Hc, s_Vc, s_VS, s_VK, s_VKt;
    double *tempB = s_tempB.get((size_t)std::max(nr, 1) * m2), *Hpc = s_Hpc.get((size_t)std::max(nr, 1) * std::max(nr, 1)), *Amr = s_Amr.get((size_t)m2 * std::max(nr, 1));
    std::vector<double> bp(n2);
    for (int k = 0; k < m2; ++k)
        for (int c = 0; c < nr; ++c) Amr[(size_t)k * nr + c] = Hp_(n2 + k, rowlive[c]);
    for (int i = 0; i < n2; ++i) bp[i] = bin[order[i]] - 0.0;
    // (rows of the kept block are independent of each other: shared out over the helper threads when there are any)
    auto schur_rows = [&](int a0, int a1) {
        for (int a = a0; a < a1; ++a) {
            const int i = rowlive[a];
            for (int j = 0; j < m2; ++j) {
                double s = 0;
                for (int k = 0; k < m2; ++k) s += Hp_(i, n2 + k) * Ainv[k * m2 + j];
                tempB[(size_t)a * m2 + j] = s;
            }
            {   // (the 15-term sums of a row's entries side by side, k outside: the same additions per entry)
                double *hrow = &Hpc[(size_t)a * nr];
#ifdef VIO_QL_AVX512
                if (!axpy_rows_wide(&tempB[(size_t)a * m2], Amr, m2, nr, nr, hrow))
#endif
                {
                    for (int c = 0; c < nr; ++c) hrow[c] = 0.0;
                    for (int k = 0; k < m2; ++k) {
                        const double tb = tempB[(size_t)a * m2 + k];
                        const double *am = &Amr[(size_t)k * nr];
                        for (int c = 0; c < nr; ++c) hrow[c] += tb * am[c];
                    }
                }
                for (int c = 0; c < nr; ++c) hrow[c] = Hp_(i, rowlive[c]) - hrow[c];
            }
            double s = 0;
            for (int k = 0; k < m2; ++k) s += tempB[(size_t)a * m2 + k] * bin[order[n2 + k]];
            bp[i] = bin[order[i]] - s;
        }
    };
    par_rows(par, nr, 24, schur_rows);
    TT(2);
    // Eigen-decomposition of the reduced system (problem.cc:766).  Rows and columns that are exactly zero — frames the
    // marginalised frame's landmarks and the old prior do not reach: 90 of the 156 in the steady state of a window with
    // tracks of 4 frames — are eigenvectors e_i of eigenvalue 0 already and fall under the 1e-8 cut whatever basis a solver
    // picks for them, so only the block they leave is decomposed (cost ~ n^3).
    int live[n2], lpos[n2], nl = 0;      // live: index in the 156-system; lpos: its place in rowlive
    for (int a = 0; a < nr; ++a) {
        bool any = false;
        for (int c = 0; c < nr && !any; ++c) any = Hpc[(size_t)a * nr + c] != 0.0 || Hpc[(size_t)c * nr + a] != 0.0;
        if (any) { live[nl] = rowlive[a]; lpos[nl] = a; ++nl; }
    }
    const int nz = n2 - nl;
    double *Hc = s_Hc.get((size_t)std::max(nl, 1) * std::max(nl, 1)), *Vc = s_Vc.get((size_t)std::max(nl, 1) * std::max(nl, 1));
    std::vector<double> evc(std::max(nl, 1));
    for (int a = 0; a < nl; ++a)
        for (int c = 0; c < nl; ++c) Hc[(size_t)a * nl + c] = Hpc[(size_t)lpos[a] * nr + lpos[c]];
    TT(3);
    if (nl > 0) symmetric_eigen(nl, Hc, evc.data(), Vc, par);
    TT(4);
    // In the 156-system the eigenvalues are: nz zeros (the unit vectors of the dead indices), then the live block's, ascending: eigenpair
    // k of the live block is number nz + k.  (A negative eigenvalue of the live block would sort before the zeros in the reference;
    // both are below the cut.)  kept: the live block's eigenvalues above the cut, the only ones the three products below see.
    int kept[n2], nk = 0;
    for (int k = 0; k < nl; ++k) if (evc[k] > eps) kept[nk++] = k;
    std::fill(jtout, jtout + (size_t)n2 * n2, 0.0);
    std::fill(errout, errout + n2, 0.0);
    for (int q = 0; q < nk; ++q) {
        const int k = kept[q], i = nz + k;
        const double sinv = std::sqrt(1.0 / evc[k]);
        double *row = jtout + (size_t)i * n2;
        for (int a = 0; a < nl; ++a) row[live[a]] = sinv * Vc[(size_t)a * nl + k];
        double s = 0;                                       // err_prior = -Jt_prior_inv b (problem.cc:774); the other rows of Jt are zero
        for (int a = 0; a < nl; ++a) s += -row[live[a]] * bp[live[a]];
        errout[i] = s;
    }
    TT(5);
    {   // H_prior = J^T J with J = sqrt(S) V^T (problem.cc:775-777): sum over the kept k of V_ik s_k V_jk, k ascending; the
        // kept eigenvectors live on the live indices only, every other entry of the product is an exact zero
        double *VS = s_VS.get((size_t)std::max(nl, 1) * std::max(nk, 1)), *VK = s_VK.get((size_t)std::max(nl, 1) * std::max(nk, 1));
        for (int a = 0; a < nl; ++a)
            for (int q = 0; q < nk; ++q) { VK[(size_t)a * nk + q] = Vc[(size_t)a * nl + kept[q]]; VS[(size_t)a * nk + q] = VK[(size_t)a * nk + q] * evc[kept[q]]; }
        std::fill(Hout, Hout + (size_t)n2 * n2, 0.0);
        // (entry (a, c) = sum over q ascending of VS[a][q] VK[c][q]: formed for a whole row of c at once, the q loop outside — every entry's
        //  additions in the same order as the dot product they replace, but 75 independent chains instead of one dependent one)
        double *VKt = s_VKt.get((size_t)std::max(nk, 1) * std::max(nl, 1));
        for (int c = 0; c < nl; ++c)
            for (int q = 0; q < nk; ++q) VKt[(size_t)q * nl + c] = VK[(size_t)c * nk + q];
        auto prior_rows = [&](int a0, int a1) {
            std::vector<double> accv((size_t)std::max(nl, 1));
            prior_product_rows(a0, a1, VS, VKt, nk, nl, live, Hout, n2, accv.data());
        };
        par_rows(par, nl, 16, prior_rows);
    }
    std::memcpy(bout, bp.data(), sizeof(double) * n2);
    TT(6);
#ifdef VIO_TAIL_TIMING
    if (++tcalls % 50 == 0) { std::fprintf(stderr, "[tail timing, avg us] Amm eigen %.1f | live-row scan %.1f | Schur rows %.1f | live block %.1f | eigen %.1f | Jt, err %.1f | H_prior %.1f\n", tt[0] / tcalls, tt[1] / tcalls, tt[2] / tcalls, tt[3] / tcalls, tt[4] / tcalls, tt[5] / tcalls, tt[6] / tcalls); }
#endif
    return nl;
}

// ---- IntegrationBase ---------------------------------------------------------------------------------------
namespace {
struct M3 { double m[9]; };
inline M3 mul(const M3 &a, const M3 &b) {
    M3 c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
inline M3 hat(const double *v) { return M3{{0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}}; }
inline M3 rot_of(const double *q) {          // Eigen toRotationMatrix on (x,y,z,w), no normalisation
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    return M3{{1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)}};
}
inline void apply(const M3 &R, const double *v, double *o) {
    for (int i = 0; i < 3; ++i) o[i] = R.m[3 * i] * v[0] + R.m[3 * i + 1] * v[1] + R.m[3 * i + 2] * v[2];
}
inline void put(double *M, int ld, int r0, int c0, const M3 &B, double s) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r0 + i) * ld + c0 + j] = B.m[3 * i + j] * s;
}
void matmul(int m, int k, int n, const double *A, const double *B, double *C) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
}  // namespace

void preintegrate(const double *acc_first, const double *gyr_first, const double *ba, const double *bg, int count,
                  const double *dts, const double *accs, const double *gyrs, double acc_n, double gyr_n, double acc_w,
                  double gyr_w, double *sum_dt_out, double *dp_out, double *dq_out, double *dv_out, double *jac,
                  double *cov) {
    double a0[3] = {acc_first[0], acc_first[1], acc_first[2]}, g0[3] = {gyr_first[0], gyr_first[1], gyr_first[2]};
    double noise[18 * 18] = {0};
    for (int i = 0; i < 3; ++i) {
        noise[19 * i] = acc_n * acc_n; noise[19 * (3 + i)] = gyr_n * gyr_n; noise[19 * (6 + i)] = acc_n * acc_n;
        noise[19 * (9 + i)] = gyr_n * gyr_n; noise[19 * (12 + i)] = acc_w * acc_w; noise[19 * (15 + i)] = gyr_w * gyr_w;
    }
    std::fill(jac, jac + 225, 0.0);
    std::fill(cov, cov + 225, 0.0);
    for (int i = 0; i < 15; ++i) jac[16 * i] = 1.0;
    double dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1}, sum_dt = 0;
    const M3 I3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    for (int s = 0; s < count; ++s) {
        const double h = dts[s];
        const double *a1 = accs + 3 * s, *g1 = gyrs + 3 * s;
        double ua0[3], ua1[3], w[3], x0[3], x1[3];
        for (int k = 0; k < 3; ++k) { x0[k] = a0[k] - ba[k]; x1[k] = a1[k] - ba[k]; w[k] = 0.5 * (g0[k] + g1[k]) - bg[k]; }
        const M3 Rd = rot_of(dq);
        apply(Rd, x0, ua0);
        // result_delta_q = delta_q * Quaterniond(1, w*dt/2): rotated with BEFORE the normalize() of propagate()
        const double iq[4] = {w[0] * h / 2, w[1] * h / 2, w[2] * h / 2, 1.0};
        const double rq[4] = {dq[3] * iq[0] + dq[0] * iq[3] + dq[1] * iq[2] - dq[2] * iq[1],
                              dq[3] * iq[1] + dq[1] * iq[3] + dq[2] * iq[0] - dq[0] * iq[2],
                              dq[3] * iq[2] + dq[2] * iq[3] + dq[0] * iq[1] - dq[1] * iq[0],
                              dq[3] * iq[3] - dq[0] * iq[0] - dq[1] * iq[1] - dq[2] * iq[2]};
        const M3 Rr = rot_of(rq);
        apply(Rr, x1, ua1);
        double rp[3], rv[3];
        for (int k = 0; k < 3; ++k) {
            const double ua = 0.5 * (ua0[k] + ua1[k]);
            rp[k] = dp[k] + dv[k] * h + 0.5 * ua * h * h;
            rv[k] = dv[k] + ua * h;
        }
        const M3 Rw = hat(w), Ra0 = hat(x0), Ra1 = hat(x1);
        M3 ImW;
        for (int k = 0; k < 9; ++k) ImW.m[k] = I3.m[k] - Rw.m[k] * h;
        const M3 RdA0 = mul(Rd, Ra0), RrA1 = mul(Rr, Ra1), RrA1I = mul(RrA1, ImW);
        double F[225] = {0}, V[15 * 18] = {0};
        M3 B;
        put(F, 15, 0, 0, I3, 1.0);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.25 * RdA0.m[k] * h * h + -0.25 * RrA1I.m[k] * h * h;
        put(F, 15, 0, 3, B, 1.0);
        put(F, 15, 0, 6, I3, h);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.25 * (Rd.m[k] + Rr.m[k]) * h * h;
        put(F, 15, 0, 9, B, 1.0);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.25 * RrA1.m[k] * h * h * -h;
        put(F, 15, 0, 12, B, 1.0);
        put(F, 15, 3, 3, ImW, 1.0);
        put(F, 15, 3, 12, I3, -1.0 * h);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.5 * RdA0.m[k] * h + -0.5 * RrA1I.m[k] * h;
        put(F, 15, 6, 3, B, 1.0);
        put(F, 15, 6, 6, I3, 1.0);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.5 * (Rd.m[k] + Rr.m[k]) * h;
        put(F, 15, 6, 9, B, 1.0);
        for (int k = 0; k < 9; ++k) B.m[k] = -0.5 * RrA1.m[k] * h * -h;
        put(F, 15, 6, 12, B, 1.0);
        put(F, 15, 9, 9, I3, 1.0);
        put(F, 15, 12, 12, I3, 1.0);
        put(V, 18, 0, 0, Rd, 0.25 * h * h);
        for (int k = 0; k < 9; ++k) B.m[k] = 0.25 * -RrA1.m[k] * h * h * 0.5 * h;
        put(V, 18, 0, 3, B, 1.0);
        put(V, 18, 0, 6, Rr, 0.25 * h * h);
        put(V, 18, 0, 9, B, 1.0);
        put(V, 18, 3, 3, I3, 0.5 * h);
        put(V, 18, 3, 9, I3, 0.5 * h);
        put(V, 18, 6, 0, Rd, 0.5 * h);
        for (int k = 0; k < 9; ++k) B.m[k] = 0.5 * -RrA1.m[k] * h * 0.5 * h;
        put(V, 18, 6, 3, B, 1.0);
        put(V, 18, 6, 6, Rr, 0.5 * h);
        put(V, 18, 6, 9, B, 1.0);
        put(V, 18, 9, 12, I3, h);
        put(V, 18, 12, 15, I3, h);
        double T1[225], T2[225], FT[225], VN[15 * 18], VT[18 * 15], T3[225];
        matmul(15, 15, 15, F, jac, T1);
        std::memcpy(jac, T1, sizeof(T1));
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 15; ++j) FT[15 * i + j] = F[15 * j + i];
        matmul(15, 15, 15, F, cov, T1);
        matmul(15, 15, 15, T1, FT, T2);
        matmul(15, 18, 18, V, noise, VN);
        for (int i = 0; i < 18; ++i)
            for (int j = 0; j < 15; ++j) VT[15 * i + j] = V[18 * j + i];
        matmul(15, 18, 15, VN, VT, T3);
        for (int k = 0; k < 225; ++k) cov[k] = T2[k] + T3[k];
        const double nq = std::sqrt(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]);
        for (int k = 0; k < 4; ++k) dq[k] = rq[k] / nq;
        for (int k = 0; k < 3; ++k) { dp[k] = rp[k]; dv[k] = rv[k]; a0[k] = a1[k]; g0[k] = g1[k]; }
        sum_dt += h;
    }
    *sum_dt_out = sum_dt;
    for (int k = 0; k < 3; ++k) { dp_out[k] = dp[k]; dv_out[k] = dv[k]; }
    for (int k = 0; k < 4; ++k) dq_out[k] = dq[k];
}

}  // namespace vio_host
