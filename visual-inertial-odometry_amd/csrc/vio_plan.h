// vio_plan.h — the part of the library's host side that needs no device: the pass over a caller's observation list and the planner
// that cuts a window's landmarks into workgroup items (Estimator::problemSolve's graph build, VM/src/estimator.cpp:909-1034, restated as
// flat tables).  Plain C++ (no HIP): vio_api.cpp calls it with the context's mirrors and uploads what it returns; tests/test_host_units.py
// compiles it with g++ alone — under AddressSanitizer / UBSan in the VIO_TEST_SANITIZE=1 tier — and drives it with refused, shuffled and
// ragged lists.
#ifndef VIO_PLAN_H
#define VIO_PLAN_H

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "vio_types.h"

namespace vio_plan {

constexpr int NF = VIO_NF;

struct Pattern {
    int use_ext, host, K, nb, host_slot, G;
    int btype_i[VIO_MAXNB], bk_i[VIO_MAXNB];
    int8_t target[VIO_MAXK], tslot[VIO_MAXK], cam_block[VIO_MAXNB];
    int n_rows, lds_doubles;
};

// One pass over an inverse-depth observation list (vio_set_observations / vio_commit_observations): any index out of range?  is it
// landmark-major, as estimator.cpp:975-1016 emits it?  and if so: do the edges of a landmark — neighbours in such a list — share host
// frame and host observation (edge_reprojection.cc:24: pts_i is the landmark's)?  pts_i_lm ([2 N], resized when it is not): the host
// observation by landmark, noted on the way; `changed`: some landmark's differs from what pts_i_lm held before the call.
struct ScanResult {
    bool bad = false;               // an index out of range (bad_index: the first such edge)
    int64_t bad_index = -1;
    bool lm_major = false;
    bool consistent = false;
    bool changed = false;
};
ScanResult scan_observations(int64_t N, int64_t m, const int32_t *lm, const int32_t *host, const int32_t *target, const double *pi,
                             std::vector<double> &pts_i_lm);
// The pass in pieces, for callers that run it on several threads (vio_set_observations: 80 000 edges are 2.2 MB to read, DRAM-bound on one
// core): scan_range over [e0, e1) — the edge before e0 is looked at, not written — gives the flags of that piece; the pieces' flags are
// OR-ed and handed to scan_finish.  Valid for landmark-major lists (a landmark's run starts in exactly one piece, which notes its host
// observation); when the OR-ed `unsorted` is set the caller runs scan_observations itself (the serial pass's last-writer semantics).
// pts_i_lm must have its size (2 N) before the pieces run.
struct ScanFlags { unsigned bad = 0, unsorted = 0, incons = 0, changed = 0; };
ScanFlags scan_range(int64_t N, int64_t e0, int64_t e1, const int32_t *lm, const int32_t *host, const int32_t *target, const double *pi, double *pts_i_lm);
ScanResult scan_finish(const ScanFlags &f, int64_t N, int64_t m, const int32_t *lm, const int32_t *host, const int32_t *target);

// A few parked helper threads for such passes (host memory bandwidth, not compute).  pool_create: nullptr when no thread can be created —
// pool_run then runs everything on the caller.  pool_run: fn(arg, i) for i = 0 .. n - 1 (n <= helpers + 1), i = 0 on the caller; returns
// when all are done.  One pool_run at a time per pool.
struct HostPool;
HostPool *pool_create(int helpers);
void pool_destroy(HostPool *p);
int pool_width(const HostPool *p);          // helpers + 1 (1 for nullptr)
void pool_run(HostPool *p, int n, void (*fn)(void *arg, int i), void *arg);
// the same for passes that coordinate among themselves (the eigen-solver's pipeline): fn(arg, i, n) with the number of participants the run
// really has — `want` at most, 1 (everything on the caller) when the pool is busy or absent
void pool_run_n(HostPool *p, int want, void (*fn)(void *arg, int i, int n), void *arg);

// The process's host threads (round 6: until then every context created its own — pool_create(3) with its first long observation list and a
// worker for its marginalisation tails: a batch of 256 contexts parked 1 024 threads).  ONE pool of SHARED_POOL_HELPERS helpers and ONE
// background worker per process, created when first wanted, reference-counted by the contexts (vio_create / vio_destroy): the last
// shared_release joins them.  A run that finds the pool busy runs on its caller (pool_run); background jobs queue in order.
constexpr int SHARED_POOL_HELPERS = 6;
void shared_acquire();
void shared_release();
HostPool *shared_pool();                    // nullptr while no context holds a reference or when no thread could be created
int shared_threads_alive();                 // helpers + background worker (tests: <= SHARED_POOL_HELPERS + 1 whatever the number of contexts)
struct BgTicket {                           // one per submitter (a context): at most one job in flight per ticket
    std::mutex mu;
    std::condition_variable cv;
    bool pending = false;
    void (*fn)(void *) = nullptr;
    void *arg = nullptr;
};
void bg_submit(BgTicket *t, void (*fn)(void *), void *arg);      // runs fn(arg) on the background worker (on the caller if there is none)
void bg_wait(BgTicket *t);                                        // returns when the ticket's job is done (at once if none is pending)

// the same for an XYZ list (landmark, observing frame): range, and landmark-major with a landmark's frames ascending
ScanResult scan_observations_xyz(int64_t N, int64_t m, const int32_t *lm, const int32_t *frame);

// LDS a workgroup item needs, in doubles (the kernels' own formulas: lin_lds_doubles / xyz_lds_doubles of vio_kernels.hip)
typedef int (*LdsFn)(int G, int K, int nb, int use_ext);
typedef int (*LdsXyzFn)(int G, int K);
// staging memory for the arrays that go to the device as they are (the HIP library hands out its pinned arena)
typedef void *(*AllocFn)(void *user, size_t bytes);

struct HostPool;
struct Input {
    int64_t N = 0, M = 0;
    HostPool *pool = nullptr;              // helper threads for the per-landmark pass (or null: the caller's thread does it all)
    const int32_t *olm = nullptr, *ohost = nullptr, *otarget = nullptr;       // observation -> landmark / host frame / target (XYZ: observing) frame
    const double *pts_i = nullptr;         // [M][2] host observation per edge (lists the scan did not vouch for), or null
    const double *pts_i_lm = nullptr;      // [N][2] host observation per landmark (vouched lists)
    const double *pts_j = nullptr;         // [M][2] (XYZ, lists that are not landmark-major: gathered into item order here)
    bool lm_major = false, vouched = false;
    int marg = 0, use_ext = 0;
    int throughput = 0;                    // VIO_ITEMS_THROUGHPUT: the largest items the LDS budget holds
    int g_max = 0, g_min = 8, n_cus = 256;
    int lin_threads = 1024, lin_threads_full = 1024, lds_budget = 0;
    LdsFn lds = nullptr;
    LdsXyzFn lds_xyz = nullptr;
    int imu_item_lds = 0;                  // floor of max_lds_doubles (the IMU workgroups' need)
};

struct Output {
    int status = 0;                        // 0, or a vio_status value (VIO_ERR_UNSUPPORTED = -5, VIO_ERR_HIP = -2 for a failed allocation)
    std::string err;
    std::vector<Pattern> patterns;
    std::vector<ItemDesc> items;
    std::vector<int32_t> sorted_to_orig;
    std::vector<int32_t> obs_idx;          // the list's CSR by landmark (lists that are not landmark-major), else empty
    int64_t Ns = 0, Ms = 0;
    size_t slab_doubles = 0, lw_doubles = 0;
    int max_lds_doubles = 0;
    double *pts_i = nullptr;               // [Ns][2] host observations in sorted landmark order            (AllocFn memory)
    int32_t *first = nullptr;              // [Ns] where sorted landmark s's observations start in the list / its CSR     (AllocFn memory)
    double *pts_j = nullptr;               // XYZ slow path: [Ms][2] in item order                                  (AllocFn memory)
    double t_us[4] = {0, 0, 0, 0};         // observation lists, pattern keys, sort + sizing, items (diagnostic)
};

void build_pattern_tables(Pattern &pt, int g_max, int threads, int lds_budget, LdsFn lds);
bool plan_invdepth(const Input &in, Output &out, AllocFn alloc, void *user);
bool plan_xyz(const Input &in, Output &out, AllocFn alloc, void *user);
// k_reduce's inverted lists: list b < 78 = camera block pair (P, Q), 78 + P = block P's vectors, 90 = chi2 / max h_ll (vio_kernels.hip)
void build_reduce_lists(const std::vector<ItemDesc> &items, std::vector<int32_t> &list_off, std::vector<int32_t> &list);

}  // namespace vio_plan
#endif
