// Diagnostic micro-benchmark (host only, not part of the product): the marginalisation's dense tail (csrc/host_dense.cpp: marginalize_tail) and its
// eigen-solver on 1 .. 7 threads of the library's own pool (csrc/vio_plan.cpp), with the check that the outputs do not depend on the thread count
// and that the eigen-solver returns the bits of the routine it replaced (symmetric_eigen_legacy).
//   g++ -O3 -std=c++17 -I visual-inertial-odometry_amd/csrc tools/microbench/host_tail.cpp visual-inertial-odometry_amd/csrc/host_dense.cpp \
//       visual-inertial-odometry_amd/csrc/vio_plan.cpp -o tools/microbench/host_tail -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "host_dense.h"
#include "vio_plan.h"
static void run_n(void *ctx, int want, void (*fn)(void *, int, int), void *arg) { vio_plan::pool_run_n((vio_plan::HostPool *)ctx, want, fn, arg); }
template <typename F> static double best_us(F f, int reps = 60) {
    double b = 1e18;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        f();
        b = std::min(b, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return b;
}
int main(int argc, char **argv) {
    const int n = 171, want_live = argc > 1 ? atoi(argv[1]) : 75, maxw = argc > 2 ? atoi(argv[2]) : 7;
    srand(1);
    // the support a steady-state prior has: ext, frame 0 (marginalised), the poses of frames 1..10, the speed-bias of frame 1
    std::vector<int> idx;
    for (int i = 0; i < 6; ++i) idx.push_back(i);
    for (int i = 0; i < 15; ++i) idx.push_back(6 + i);
    for (int f = 1; f <= 10; ++f) for (int i = 0; i < 6; ++i) idx.push_back(6 + 15 * f + i);
    for (int i = 0; i < 9; ++i) idx.push_back(6 + 15 + 6 + i);
    while ((int)idx.size() > want_live + 15) idx.pop_back();
    const int m = (int)idx.size();
    std::vector<double> A((size_t)m * 200), H((size_t)n * n, 0.0), b(n, 0.0);
    for (auto &x : A) x = (rand() / (double)RAND_MAX - 0.5) * 100;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) { double s = 0; for (int k = 0; k < 200; ++k) s += A[(size_t)i * 200 + k] * A[(size_t)j * 200 + k]; H[(size_t)idx[i] * n + idx[j]] = s; }
    for (int i = 0; i < m; ++i) b[idx[i]] = rand() / (double)RAND_MAX;
    vio_plan::HostPool *pool = vio_plan::pool_create(6);
    std::vector<double> Ho(156 * 156), bo(156), eo(156), jo(156 * 156), Ho1, bo1, eo1, jo1;
    int nl = 0, bad = 0;
    for (int w = 1; w <= maxw; ++w) {
        vio_host::Par par{pool, run_n, w};
        std::vector<double> Hc = H, bc = b;
        nl = vio_host::marginalize_tail(Hc.data(), bc.data(), 0, Ho.data(), bo.data(), eo.data(), jo.data(), w == 1 ? nullptr : &par);
        if (w == 1) { Ho1 = Ho; bo1 = bo; eo1 = eo; jo1 = jo; }
        else if (memcmp(Ho.data(), Ho1.data(), Ho.size() * 8) || memcmp(bo.data(), bo1.data(), bo.size() * 8) || memcmp(eo.data(), eo1.data(), eo.size() * 8) || memcmp(jo.data(), jo1.data(), jo.size() * 8)) ++bad;
    }
    std::vector<double> M((size_t)nl * nl), e1(nl), V1((size_t)nl * nl), e2(nl), V2((size_t)nl * nl);
    for (int i = 0; i < nl; ++i) for (int j = 0; j < nl; ++j) M[(size_t)i * nl + j] = H[(size_t)idx[15 + 6 > i ? i : i] * n + idx[j]];
    for (int i = 0; i < nl; ++i) for (int j = 0; j < i; ++j) M[(size_t)j * nl + i] = M[(size_t)i * nl + j];
    vio_host::symmetric_eigen_legacy(nl, M.data(), e1.data(), V1.data());
    for (int w = 1; w <= maxw; ++w) {
        vio_host::Par par{pool, run_n, w};
        vio_host::symmetric_eigen(nl, M.data(), e2.data(), V2.data(), w == 1 ? nullptr : &par);
        if (memcmp(e1.data(), e2.data(), e1.size() * 8) || memcmp(V1.data(), V2.data(), V1.size() * 8)) ++bad;
    }
    printf("live rows %d; outputs that differ from the one-thread / legacy ones: %d\n", nl, bad);
    printf("symmetric_eigen(%d): legacy %.1f us;", nl, best_us([&] { vio_host::symmetric_eigen_legacy(nl, M.data(), e1.data(), V1.data()); }));
    for (int w = 1; w <= maxw; ++w) { vio_host::Par par{pool, run_n, w}; printf(" %d thr %.1f", w, best_us([&] { vio_host::symmetric_eigen(nl, M.data(), e2.data(), V2.data(), w == 1 ? nullptr : &par); })); }
    printf(" us\nmarginalize_tail:");
    for (int w = 1; w <= maxw; ++w) {
        vio_host::Par par{pool, run_n, w};
        printf(" %d thr %.1f", w, best_us([&] { std::vector<double> Hc = H, bc = b; vio_host::marginalize_tail(Hc.data(), bc.data(), 0, Ho.data(), bo.data(), eo.data(), jo.data(), w == 1 ? nullptr : &par); }));
    }
    printf(" us\n");
    vio_plan::pool_destroy(pool);
    return bad != 0;
}
