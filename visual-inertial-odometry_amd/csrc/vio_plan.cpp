// vio_plan.cpp — see vio_plan.h.  No HIP in this file.
#include "vio_plan.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <unordered_map>

namespace vio_plan {

namespace {
constexpr int ERR_UNSUPPORTED = -5, ERR_ALLOC = -2;
bool fail(Output &out, int status, const std::string &msg) {
    out.status = status;
    out.err = msg;
    return false;
}
double us_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

// Landmarks per item.  One k_linearize workgroup per item, one workgroup per CU (LDS), so the kernel takes
// rounds x (time of a workgroup), rounds = ceil((items + IMU workgroups) / CUs), and a workgroup of g landmarks takes
// ~ (70 + g) x 170 cycles (measured 48..80, tools/diag_wg_timeline.py).  A workgroup too many doubles the kernel:
// 20 000 landmarks in 7 patterns are 252 + 10 workgroups at 80 landmarks per item and 245 + 10 at 82 — one round on
// the 256 CUs of an MI355X instead of two (20 -> 12 us).  Items of a pattern are then evened out.
int best_item_size(const Input &in, const std::vector<Pattern> &patterns, const std::vector<int64_t> &n_of) {
    const int cus = std::max(1, in.n_cus);
    int best_g = 0;
    double best_cost = 0.0;
    for (int g = (in.throughput ? 128 : in.g_min); g <= 128; ++g) {      // (throughput: the largest items the LDS holds)
        int64_t blocks = VIO_NF - 1;
        int g_eff = 1;
        for (size_t q = 0; q < patterns.size(); ++q) {
            const int gp = std::min(g, patterns[q].G);
            const int64_t ni = (n_of[q] + gp - 1) / gp;
            blocks += ni;
            if (ni) g_eff = std::max<int>(g_eff, (int)((n_of[q] + ni - 1) / ni));
        }
        const double cost = (double)((blocks + cus - 1) / cus) * (70.0 + g_eff);
        if (best_g == 0 || cost < best_cost) { best_g = g; best_cost = cost; }
        // (one round already at the smallest size: a larger g keeps the one round and only lengthens its workgroups — g_eff does not fall
        //  as g grows — so the search would end where it started: the windows an Estimator produces, a few hundred landmarks, leave here)
        if (g == in.g_min && !in.throughput && blocks <= cus) break;
    }
    return best_g;
}
}  // namespace

ScanFlags scan_range(int64_t N, int64_t e0, int64_t e1, const int32_t *lm, const int32_t *host, const int32_t *target, const double *pi, double *pl) {
    ScanFlags f;
    const uint32_t un = (uint32_t)std::min<int64_t>(N, INT32_MAX);
    int32_t prev = e0 > 0 ? lm[e0 - 1] : -1;
    for (int64_t e = e0; e < e1; ++e) {
        const int32_t l = lm[e];
        f.bad |= (unsigned)((uint32_t)l >= un) | (unsigned)((uint32_t)host[e] >= (uint32_t)NF) | (unsigned)((uint32_t)target[e] >= (uint32_t)NF) |
                 (unsigned)(host[e] == target[e]);
        f.unsorted |= (unsigned)(l < prev);
        if (e > 0 && l == prev) f.incons |= (unsigned)(host[e] != host[e - 1]) | (unsigned)(pi[2 * e] != pi[2 * e - 2]) | (unsigned)(pi[2 * e + 1] != pi[2 * e - 1]);
        else if ((uint32_t)l < un) {
            f.changed |= (unsigned)(pl[2 * (size_t)l] != pi[2 * e]) | (unsigned)(pl[2 * (size_t)l + 1] != pi[2 * e + 1]);
            pl[2 * (size_t)l] = pi[2 * e]; pl[2 * (size_t)l + 1] = pi[2 * e + 1];
        }
        prev = l;
    }
    return f;
}
ScanResult scan_finish(const ScanFlags &f, int64_t N, int64_t m, const int32_t *lm, const int32_t *host, const int32_t *target) {
    ScanResult r;
    r.changed = f.changed != 0;
    if (f.bad)
        for (int64_t e = 0; e < m; ++e)
            if (lm[e] < 0 || lm[e] >= N || host[e] < 0 || host[e] >= NF || target[e] < 0 || target[e] >= NF || host[e] == target[e]) {
                r.bad = true; r.bad_index = e;
                return r;
            }
    r.lm_major = !f.unsorted;
    r.consistent = !f.unsorted && !f.incons;
    return r;
}
ScanResult scan_observations(int64_t N, int64_t m, const int32_t *lm, const int32_t *host, const int32_t *target, const double *pi,
                             std::vector<double> &pts_i_lm) {
    unsigned resized = 0;
    if (pts_i_lm.size() != 2 * (size_t)N) { pts_i_lm.assign(2 * (size_t)N, 0.0); resized = 1; }
    ScanFlags f = scan_range(N, 0, m, lm, host, target, pi, pts_i_lm.data());
    f.changed |= resized;
    return scan_finish(f, N, m, lm, host, target);
}

// ---- helper threads ----
// A pool is a few parked threads and ONE run at a time: a caller that finds it busy (another context's pass, the marginalisation's
// background tail) runs its tasks itself, in order — the same results, nobody waits for anybody.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    uint64_t gen = 0;                       // bumped by every pool_run
    int n = 0;                              // tasks of the current run (task i >= 1 belongs to helper i - 1)
    void (*fn)(void *, int) = nullptr;
    void (*fn_n)(void *, int, int) = nullptr;
    void *arg = nullptr;
    std::atomic<int> remaining{0};
    std::atomic<uint64_t> gen_hint{0};
    std::atomic<bool> busy{false};          // a run is in flight (try-lock of the callers)
    bool quit = false;
};
static void pool_loop(HostPool *p, int me) {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        if (p->gen == seen && !p->quit) {
            // (a pass is often followed by another within microseconds — the stages of the marginalisation's tail —: look for it a little
            //  while before parking; gen_hint mirrors gen for readers without the mutex)
            lk.unlock();
            for (int spin = 0; spin < 4000 && p->gen_hint.load(std::memory_order_acquire) == seen; ++spin) { }
            lk.lock();
        }
        p->cv.wait(lk, [&] { return p->gen != seen || p->quit; });
        if (p->quit) return;
        seen = p->gen;
        if (me + 1 < p->n) {
            void (*fn)(void *, int) = p->fn;
            void (*fn_n)(void *, int, int) = p->fn_n;
            void *arg = p->arg;
            const int n = p->n;
            lk.unlock();
            if (fn_n) fn_n(arg, me + 1, n); else fn(arg, me + 1);
            lk.lock();
            if (p->remaining.fetch_sub(1) == 1) p->cv_done.notify_all();
        }
    }
}
HostPool *pool_create(int helpers) {
    HostPool *p = nullptr;
    try {
        p = new HostPool;
        for (int i = 0; i < helpers; ++i) p->th.emplace_back(pool_loop, p, i);
    } catch (...) {
        if (p) pool_destroy(p);
        return nullptr;
    }
    return p;
}
void pool_destroy(HostPool *p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->quit = true; }
    p->cv.notify_all();
    for (auto &t : p->th) if (t.joinable()) t.join();
    delete p;
}
int pool_width(const HostPool *p) { return p ? (int)p->th.size() + 1 : 1; }
static void pool_dispatch(HostPool *p, int n, void (*fn)(void *, int), void (*fn_n)(void *, int, int), void *arg) {
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->n = n; p->fn = fn; p->fn_n = fn_n; p->arg = arg;
        p->remaining.store(n - 1);
        ++p->gen;
        p->gen_hint.store(p->gen, std::memory_order_release);
    }
    p->cv.notify_all();
    if (fn_n) fn_n(arg, 0, n); else fn(arg, 0);
    for (int spin = 0; spin < 2000 && p->remaining.load() != 0; ++spin) { }      // (a helper is usually a few microseconds behind)
    if (p->remaining.load() != 0) {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return p->remaining.load() == 0; });
    }
    p->busy.store(false, std::memory_order_release);
}
void pool_run(HostPool *p, int n, void (*fn)(void *arg, int i), void *arg) {
    if (p && n > 1) {
        bool expected = false;
        if (p->busy.compare_exchange_strong(expected, true, std::memory_order_acquire)) { pool_dispatch(p, std::min(n, pool_width(p)), fn, nullptr, arg); return; }
    }
    for (int i = 0; i < n; ++i) fn(arg, i);
}
void pool_run_n(HostPool *p, int want, void (*fn)(void *arg, int i, int n), void *arg) {
    if (p && want > 1 && pool_width(p) > 1) {
        bool expected = false;
        if (p->busy.compare_exchange_strong(expected, true, std::memory_order_acquire)) { pool_dispatch(p, std::min(want, pool_width(p)), nullptr, fn, arg); return; }
    }
    fn(arg, 0, 1);
}

// ---- the process's shared pool and its background worker ----
namespace {
std::mutex g_shared_mu;
HostPool *g_shared = nullptr;
int g_shared_refs = 0;
bool g_shared_tried = false;

struct BgWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<BgTicket *> queue;
    bool quit = false;
};
BgWorker *g_bg = nullptr;
bool g_bg_tried = false;
std::atomic<int> g_threads_alive{0};

void bg_loop(BgWorker *w) {
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return !w->queue.empty() || w->quit; });
        if (w->queue.empty() && w->quit) return;
        BgTicket *t = w->queue.front();
        w->queue.erase(w->queue.begin());
        lk.unlock();
        t->fn(t->arg);
        {
            // (notified under the ticket's mutex: the waiter cannot leave bg_wait — and, say, destroy the ticket — before this thread is
            //  done with it; found by the ThreadSanitizer tier)
            std::lock_guard<std::mutex> tl(t->mu);
            t->pending = false;
            t->cv.notify_all();
        }
        lk.lock();
    }
}
}  // namespace

void shared_acquire() {
    std::lock_guard<std::mutex> lk(g_shared_mu);
    ++g_shared_refs;
}
void shared_release() {
    HostPool *dead = nullptr;
    BgWorker *bg = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_shared_mu);
        if (g_shared_refs > 0 && --g_shared_refs == 0) {
            dead = g_shared; g_shared = nullptr; g_shared_tried = false;
            bg = g_bg; g_bg = nullptr; g_bg_tried = false;
        }
    }
    if (dead) { g_threads_alive.fetch_sub((int)dead->th.size()); pool_destroy(dead); }
    if (bg) {
        { std::lock_guard<std::mutex> lk(bg->mu); bg->quit = true; }
        bg->cv.notify_all();
        if (bg->th.joinable()) bg->th.join();
        g_threads_alive.fetch_sub(1);
        delete bg;
    }
}
HostPool *shared_pool() {
    std::lock_guard<std::mutex> lk(g_shared_mu);
    if (!g_shared && !g_shared_tried && g_shared_refs > 0) {
        g_shared_tried = true;
        unsigned hw = std::thread::hardware_concurrency();
        int helpers = SHARED_POOL_HELPERS;
        if (hw > 0 && (int)hw - 1 < helpers) helpers = (int)hw - 1;
        if (helpers > 0) g_shared = pool_create(helpers);
        if (g_shared) g_threads_alive.fetch_add((int)g_shared->th.size());
    }
    return g_shared;
}
int shared_threads_alive() { return g_threads_alive.load(); }

void bg_submit(BgTicket *t, void (*fn)(void *), void *arg) {
    t->fn = fn; t->arg = arg;
    BgWorker *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_shared_mu);
        if (!g_bg && !g_bg_tried && g_shared_refs > 0) {
            g_bg_tried = true;
            try {
                g_bg = new BgWorker;
                g_bg->th = std::thread(bg_loop, g_bg);
                g_threads_alive.fetch_add(1);
            } catch (...) {
                delete g_bg;
                g_bg = nullptr;
            }
        }
        w = g_bg;
        if (w) {
            // (queued under g_shared_mu: shared_release cannot take the worker away between the look and the push)
            { std::lock_guard<std::mutex> tl(t->mu); t->pending = true; }
            std::lock_guard<std::mutex> wl(w->mu);
            w->queue.push_back(t);
        }
    }
    if (!w) { fn(arg); return; }            // (no worker: the job is done when the call returns)
    w->cv.notify_one();
}
void bg_wait(BgTicket *t) {
    std::unique_lock<std::mutex> lk(t->mu);
    t->cv.wait(lk, [&] { return !t->pending; });
}

ScanResult scan_observations_xyz(int64_t N, int64_t m, const int32_t *lm, const int32_t *frame) {
    ScanResult r;
    unsigned bad = 0, unordered = 0;
    const uint32_t un = (uint32_t)std::min<int64_t>(N, INT32_MAX);
    int32_t pl = -1, pf = -1;
    for (int64_t e = 0; e < m; ++e) {
        const int32_t l = lm[e], f = frame[e];
        bad |= (unsigned)((uint32_t)l >= un) | (unsigned)((uint32_t)f >= (uint32_t)NF);
        unordered |= (unsigned)(l < pl) | ((unsigned)(l == pl) & (unsigned)(f <= pf));
        pl = l; pf = f;
    }
    if (bad)
        for (int64_t e = 0; e < m; ++e)
            if (lm[e] < 0 || lm[e] >= N || frame[e] < 0 || frame[e] >= NF) { r.bad = true; r.bad_index = e; return r; }
    r.lm_major = !unordered;
    return r;
}

// Block types of a pattern, its slab size and the largest G that fits the LDS budget.  The slab of an item holds,
// in this order: the 6x6 blocks of the pattern-local pairs (p <= q, p-major), then per block the direct b, the Schur
// correction of b and the direct diagonal, then chi2 and max h_ll (k_linearize writes it, k_reduce's lists index it).
void build_pattern_tables(Pattern &pt, int g_max, int threads, int lds_budget, LdsFn lds) {
    const int nb = pt.nb, K = pt.K;
    int *type = pt.btype_i, *kof = pt.bk_i;
    for (int p = 0; p < nb; ++p) {
        type[p] = (pt.use_ext && p == 0) ? 0 : (p == pt.host_slot ? 1 : 2);
        kof[p] = 15;
    }
    for (int k = 0; k < K; ++k) kof[pt.tslot[k]] = k;
    pt.n_rows = item_nbp(nb) * 6 + 3 * nb;
    int G = std::max(1, std::min(g_max, threads / K));                     // one thread per observation in k_linearize's phase 1
    G = std::max(1, std::min(G, 7 * threads / (6 * nb + 2)));              // k_linearize stages the item's Schur rows with 7 loads per thread
    if (G > 1 && lds(G, K, nb, pt.use_ext) > lds_budget) {
        // the largest G the LDS holds (the size grows with G: by bisection — one G at a time this loop was most of a small window's pattern pass)
        int lo = 1, hi = G;                                                // lo fits (or is 1), hi does not
        while (hi - lo > 1) {
            const int mid = (lo + hi) / 2;
            if (lds(mid, K, nb, pt.use_ext) > lds_budget) hi = mid; else lo = mid;
        }
        G = lo;
    }
    pt.G = G;
    pt.lds_doubles = lds(G, K, nb, pt.use_ext);
}

// The pattern key of every landmark — (count, host, targets in observation order), 4 bits a field — and the per-landmark checks, over a range
// of landmarks: the part of the planner's pattern pass that needs nothing but the landmark's own edges (pieces of it run on helper threads,
// Input::pool).  packed[l] = 0: the landmark is not part of this graph (MargOldFrame: not hosted in frame 0, or without observations).
namespace {
enum { PK_OK = 0, PK_NO_OBS, PK_TOO_MANY, PK_HOST, PK_TWICE };
struct PatKeys {
    const Input *in;
    const int64_t *obs_off;
    const int32_t *obs_idx;          // null: landmark-major (the list is its own CSR)
    bool vouched;
    uint64_t *packed;
    int pieces;
    int64_t N;
    int err[8];
    int64_t err_l[8];
};
void pat_keys_piece(void *arg, int i) {
    PatKeys &j = *(PatKeys *)arg;
    const Input &in = *j.in;
    j.err[i] = PK_OK; j.err_l[i] = -1;
    const int64_t l0 = j.N * i / j.pieces, l1 = j.N * (i + 1) / j.pieces;
    for (int64_t l = l0; l < l1; ++l) {
        const int64_t o0 = j.obs_off[l], n = j.obs_off[l + 1] - o0;
        const int32_t *ix = j.obs_idx ? j.obs_idx + o0 : nullptr;
        auto edge = [&](int64_t k) { return ix ? (int64_t)ix[k] : o0 + k; };
        j.packed[l] = 0;
        if (n == 0) {
            if (in.marg) continue;
            j.err[i] = PK_NO_OBS; j.err_l[i] = l; return;
        }
        const int64_t e0 = edge(0);
        const int h = in.ohost[e0];
        if (in.marg && h != 0) continue;          // MargOldFrame keeps landmarks hosted in frame 0 only (estimator.cpp:762-764)
        if (n > VIO_MAXK) { j.err[i] = PK_TOO_MANY; j.err_l[i] = l; return; }
        uint64_t packed = (uint64_t)n | ((uint64_t)h << 4);
        unsigned seen = 1u << h;
        for (int64_t k = 0; k < n; ++k) {
            const int64_t e = edge(k);
            if (!j.vouched && (in.ohost[e] != h || in.pts_i[2 * e] != in.pts_i[2 * e0] || in.pts_i[2 * e + 1] != in.pts_i[2 * e0 + 1])) { j.err[i] = PK_HOST; j.err_l[i] = l; return; }
            const int t = in.otarget[e];
            if (seen & (1u << t)) { j.err[i] = PK_TWICE; j.err_l[i] = l; return; }
            seen |= 1u << t;
            packed |= (uint64_t)t << (4 * (k + 2));
        }
        j.packed[l] = packed;
    }
}
}  // namespace

bool plan_invdepth(const Input &in, Output &out, AllocFn alloc, void *user) {
    const int64_t N = in.N, M = in.M;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    const auto tp0 = tnow();
    out.status = 0; out.err.clear();
    // observations of each landmark, in the caller's order (CSR; this runs once per frame on the host, so no
    // per-landmark allocations and no tree lookups: 20 000 landmarks take well under a millisecond)
    // (a landmark-major list — what the reference's loop emits — is its own CSR: observation k of landmark l is obs_off[l] + k)
    struct ObsRange { const int32_t *p; size_t n; int32_t base; size_t size() const { return n; } bool empty() const { return n == 0; }
                      int32_t operator[](size_t i) const { return p ? p[i] : base + (int32_t)i; } };
    std::vector<int64_t> obs_off(N + 1, 0);
    for (int64_t e = 0; e < M; ++e) ++obs_off[in.olm[e] + 1];
    for (int64_t l = 0; l < N; ++l) obs_off[l + 1] += obs_off[l];
    const bool lm_major = in.lm_major;
    std::vector<int32_t> &obs_idx = out.obs_idx;
    obs_idx.clear();
    if (!lm_major) {
        obs_idx.resize(std::max<int64_t>(M, 1));
        std::vector<int64_t> fill(obs_off.begin(), obs_off.end() - 1);
        for (int64_t e = 0; e < M; ++e) obs_idx[fill[in.olm[e]]++] = (int32_t)e;
    }
    auto obs_of = [&](int64_t l) { return ObsRange{lm_major ? nullptr : obs_idx.data() + obs_off[l], (size_t)(obs_off[l + 1] - obs_off[l]), (int32_t)obs_off[l]}; };
    const auto tp1 = tnow();
    // pattern of each landmark: (host, targets in observation order) packed 4 bits a frame
    std::unordered_map<uint64_t, int> pattern_id;
    uint64_t pat_cache_key[256];
    int pat_cache_id[256];
    for (int q = 0; q < 256; ++q) pat_cache_id[q] = -1;
    out.patterns.clear();
    std::vector<int32_t> lm_pattern(N, -1);
    const bool vouched = lm_major && in.vouched;
    if (!vouched && M > 0 && !in.pts_i) return fail(out, ERR_UNSUPPORTED, "the per-edge host observations are needed for a list the scan did not vouch for");
    std::vector<uint64_t> packed_of((size_t)std::max<int64_t>(N, 1));
    {
        PatKeys pk;
        pk.in = &in; pk.obs_off = obs_off.data(); pk.obs_idx = lm_major ? nullptr : obs_idx.data(); pk.vouched = vouched; pk.packed = packed_of.data(); pk.N = N;
        pk.pieces = (in.pool && N >= 4096) ? std::min(pool_width(in.pool), 8) : 1;
        pool_run(pk.pieces > 1 ? in.pool : nullptr, pk.pieces, pat_keys_piece, &pk);
        // the first landmark (in list order) that was refused names the error, as one pass over the landmarks would
        int code = PK_OK;
        int64_t at = -1;
        for (int i = 0; i < pk.pieces; ++i) if (pk.err[i] != PK_OK && (at < 0 || pk.err_l[i] < at)) { code = pk.err[i]; at = pk.err_l[i]; }
        if (code == PK_NO_OBS) return fail(out, ERR_UNSUPPORTED, "landmark without observations (its 1x1 Hessian block would be singular)");
        if (code == PK_TOO_MANY) return fail(out, ERR_UNSUPPORTED, "more than 10 observations of one landmark");
        if (code == PK_HOST) return fail(out, ERR_UNSUPPORTED, "edges of one landmark must share host frame and host observation");
        if (code == PK_TWICE) return fail(out, ERR_UNSUPPORTED, "two observations of one landmark in the same frame");
    }
    for (int64_t l = 0; l < N; ++l) {
        const uint64_t packed = packed_of[l];
        if (packed == 0) continue;
        int id = -1;
        const unsigned hslot = (unsigned)((packed * 0x9E3779B97F4A7C15ull) >> 56);      // 256 slots in front of the map
        if (pat_cache_id[hslot] >= 0 && pat_cache_key[hslot] == packed) id = pat_cache_id[hslot];
        else {
            auto itp = pattern_id.find(packed);
            if (itp != pattern_id.end()) { id = itp->second; pat_cache_key[hslot] = packed; pat_cache_id[hslot] = id; }
        }
        if (id < 0) {
            id = (int)out.patterns.size();
            pattern_id[packed] = id;
            pat_cache_key[hslot] = packed; pat_cache_id[hslot] = id;
            Pattern pt;
            std::memset(&pt, 0, sizeof(pt));
            const int K = (int)(packed & 15), h = (int)((packed >> 4) & 15);
            pt.use_ext = in.use_ext; pt.host = h; pt.K = K;
            bool seen[NF] = {false};
            seen[h] = true;
            for (int k = 0; k < K; ++k) seen[(packed >> (4 * (k + 2))) & 15] = true;
            int frames[NF], nfr = 0;
            for (int f = 0; f < NF; ++f) if (seen[f]) frames[nfr++] = f;
            pt.nb = nfr + pt.use_ext;
            int p = 0;
            if (pt.use_ext) pt.cam_block[p++] = 0;
            for (int q = 0; q < nfr; ++q) { const int f = frames[q]; if (f == h) pt.host_slot = p; pt.cam_block[p++] = (int8_t)(1 + f); }
            for (int k = 0; k < K; ++k) {
                const int t = (int)((packed >> (4 * (k + 2))) & 15);
                pt.target[k] = (int8_t)t;
                for (int q = 0; q < pt.nb; ++q) if (pt.cam_block[q] == 1 + t) pt.tslot[k] = (int8_t)q;
            }
            build_pattern_tables(pt, in.g_max > 0 ? in.g_max : 128, in.lin_threads, in.lds_budget, in.lds);      // pt.G = the most landmarks the LDS holds
            out.patterns.push_back(pt);
        }
        lm_pattern[l] = id;
    }
    const auto tp2 = tnow();
    // sort landmarks by pattern (stable in the original index): counting sort, pattern-major, original index inside a pattern
    {
        std::vector<int64_t> start(out.patterns.size() + 1, 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) ++start[lm_pattern[l] + 1];
        for (size_t q = 0; q < out.patterns.size(); ++q) start[q + 1] += start[q];
        out.sorted_to_orig.assign((size_t)start[out.patterns.size()], 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) out.sorted_to_orig[start[lm_pattern[l]]++] = (int32_t)l;
    }
    out.Ns = (int64_t)out.sorted_to_orig.size();
    {
        std::vector<int64_t> n_of(out.patterns.size(), 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) ++n_of[lm_pattern[l]];
        const int best_g = best_item_size(in, out.patterns, n_of);
        for (size_t q = 0; q < out.patterns.size(); ++q) {
            Pattern &pt = out.patterns[q];
            const int gp = std::min(best_g, pt.G);
            const int64_t ni = std::max<int64_t>(1, (n_of[q] + gp - 1) / gp);
            pt.G = (int)std::max<int64_t>(1, (n_of[q] + ni - 1) / ni);
            pt.lds_doubles = in.lds(pt.G, pt.K, pt.nb, pt.use_ext);
        }
    }
    // items
    const auto tp3 = tnow();
    out.items.clear();
    // host observations in sorted landmark order, written straight into the caller's staging
    out.pts_i = (double *)alloc(user, 2 * (size_t)std::max<int64_t>(out.Ns, 1) * 8);
    out.first = (int32_t *)alloc(user, (size_t)std::max<int64_t>(out.Ns, 1) * 4);
    if (!out.pts_i || !out.first) return fail(out, ERR_ALLOC, "staging allocation failed");
    out.slab_doubles = 0; out.lw_doubles = 0; out.max_lds_doubles = in.imu_item_lds;
    int64_t s = 0, obs_base = 0;
    while (s < out.Ns) {
        const int id = lm_pattern[out.sorted_to_orig[s]];
        const Pattern &pt = out.patterns[id];
        int64_t e = s;
        while (e < out.Ns && e - s < pt.G && lm_pattern[out.sorted_to_orig[e]] == id) ++e;
        ItemDesc it;
        std::memset(&it, 0, sizeof(it));
        it.lm_base = (int32_t)s; it.G = (int32_t)(e - s); it.K = pt.K; it.nb = pt.nb; it.host = pt.host;
        it.host_slot = pt.host_slot; it.use_ext = pt.use_ext; it.obs_base = (int32_t)obs_base;
        it.out_base = (int32_t)out.slab_doubles; it.lw_base = (int32_t)out.lw_doubles;
        std::memcpy(it.target, pt.target, sizeof(it.target)); std::memcpy(it.tslot, pt.tslot, sizeof(it.tslot));
        std::memcpy(it.cam_block, pt.cam_block, sizeof(it.cam_block));
        for (int p = 0; p < pt.nb; ++p) { it.btype[p] = (int8_t)pt.btype_i[p]; it.bk[p] = (int8_t)pt.bk_i[p]; }
        it.n_rows = pt.n_rows; it.lds_doubles = pt.lds_doubles;
        out.items.push_back(it);
        out.max_lds_doubles = std::max(out.max_lds_doubles, pt.lds_doubles);
        out.slab_doubles += (size_t)item_out_count(pt.nb);
        out.slab_doubles = (out.slab_doubles + 1) & ~(size_t)1;
        out.lw_doubles += (size_t)item_lw_fields(pt.nb) * it.G;
        for (int g = 0; g < it.G; ++g) {
            const int32_t l = out.sorted_to_orig[s + g];
            const ObsRange ob = obs_of(l);
            const double *hp = vouched ? &in.pts_i_lm[2 * (size_t)l] : &in.pts_i[2 * (size_t)ob[0]];
            out.pts_i[2 * (s + g)] = hp[0]; out.pts_i[2 * (s + g) + 1] = hp[1];
            out.first[s + g] = (int32_t)obs_off[l];          // (the target observations follow on the device: k_gather_obs)
        }
        obs_base += (int64_t)it.G * it.K;
        s = e;
    }
    out.Ms = obs_base;
    const auto tp4 = tnow();
    out.t_us[0] = us_between(tp0, tp1); out.t_us[1] = us_between(tp1, tp2); out.t_us[2] = us_between(tp2, tp3); out.t_us[3] = us_between(tp3, tp4);
    return true;
}

// The plan of a window of XYZ landmarks (vio_kernels_xyz.h): a pattern is the set of frames a landmark is seen from,
// one pattern block per frame, no host frame and no extrinsic block.
// marg: Problem::Marginalize's graph (problem.cc:617-637): the edges connected to the pose of frame 0 and the landmarks they touch —
// a landmark seen from frame 0 enters with that one observation, whatever else observes it.
bool plan_xyz(const Input &in, Output &out, AllocFn alloc, void *user) {
    const int64_t N = in.N, M = in.M;
    out.status = 0; out.err.clear();
    out.obs_idx.clear();
    // A landmark-major list with ascending frames (what the scan found) is its own CSR and observation k of a landmark is its
    // pattern's k-th frame: no table of observations by (landmark, frame), and the observations go to the device as listed
    // (k_gather_obs puts them into item order).  Any other list: the table, and the gather on the host.
    const bool fast = in.lm_major;
    const int marg = in.marg;
    std::vector<int64_t> obs_off;
    if (fast) {
        obs_off.assign((size_t)N + 1, 0);
        for (int64_t e = 0; e < M; ++e) ++obs_off[in.olm[e] + 1];
        for (int64_t l = 0; l < N; ++l) obs_off[l + 1] += obs_off[l];
    }
    // observation of landmark l in frame f: obs_at[l * NF + f] (or -1)
    std::vector<int32_t> obs_at;
    if (!fast) {
        obs_at.assign((size_t)std::max<int64_t>(N, 1) * NF, -1);
        for (int64_t e = 0; e < M; ++e) {
            if (marg && in.otarget[e] != 0) continue;
            int32_t &slot = obs_at[(size_t)in.olm[e] * NF + in.otarget[e]];
            if (slot >= 0) return fail(out, ERR_UNSUPPORTED, "two observations of one landmark in the same frame");
            slot = (int32_t)e;
        }
    }
    std::vector<int32_t> mask(N, 0), pat_of_mask(1 << NF, -1), lm_pattern(N, -1);
    out.patterns.clear();
    for (int64_t l = 0; l < N; ++l) {
        int m = 0;
        if (fast) {
            for (int64_t e = obs_off[l]; e < obs_off[l + 1]; ++e) m |= 1 << in.otarget[e];
            if (marg) m &= 1;              // Problem::Marginalize's graph: the observation frame 0 has of the landmark, nothing else
        } else
            for (int f = 0; f < NF; ++f) if (obs_at[(size_t)l * NF + f] >= 0) m |= 1 << f;
        if (m == 0) {
            if (marg) continue;
            return fail(out, ERR_UNSUPPORTED, "landmark without observations (its 3x3 Hessian block would be singular)");
        }
        mask[l] = m;
        if (pat_of_mask[m] < 0) {
            pat_of_mask[m] = (int)out.patterns.size();
            Pattern pt;
            std::memset(&pt, 0, sizeof(pt));
            pt.host = -1; pt.host_slot = -1;
            int p = 0;
            for (int f = 0; f < NF; ++f) if ((m >> f) & 1) {
                pt.cam_block[p] = (int8_t)(1 + f);
                if (p < VIO_MAXK) { pt.target[p] = (int8_t)f; pt.tslot[p] = (int8_t)p; }
                pt.btype_i[p] = 2; pt.bk_i[p] = p;
                ++p;
            }
            pt.K = pt.nb = p;
            pt.n_rows = item_nbp(pt.nb) * 6 + 3 * pt.nb;
            int G = std::max(1, std::min(in.g_max > 0 ? in.g_max : 128, in.lin_threads_full / pt.K));
            while (G > 1 && in.lds_xyz(G, pt.K) > in.lds_budget) --G;
            pt.G = G;
            pt.lds_doubles = in.lds_xyz(G, pt.K);
            out.patterns.push_back(pt);
        }
        lm_pattern[l] = pat_of_mask[m];
    }
    {   // counting sort: pattern-major, original index inside a pattern
        std::vector<int64_t> start(out.patterns.size() + 1, 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) ++start[lm_pattern[l] + 1];
        for (size_t q = 0; q < out.patterns.size(); ++q) start[q + 1] += start[q];
        out.sorted_to_orig.assign((size_t)start[out.patterns.size()], 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) out.sorted_to_orig[start[lm_pattern[l]]++] = (int32_t)l;
    }
    out.Ns = (int64_t)out.sorted_to_orig.size();
    {   // landmarks per item: whole rounds of the device's CUs, as plan_invdepth does, within what the LDS holds per pattern
        std::vector<int64_t> n_of(out.patterns.size(), 0);
        for (int64_t l = 0; l < N; ++l) if (lm_pattern[l] >= 0) ++n_of[lm_pattern[l]];
        const int best_g = best_item_size(in, out.patterns, n_of);
        for (size_t q = 0; q < out.patterns.size(); ++q) {
            Pattern &pt = out.patterns[q];
            const int gp = std::min(best_g, pt.G);
            const int64_t ni = std::max<int64_t>(1, (n_of[q] + gp - 1) / gp);
            pt.G = (int)std::max<int64_t>(1, (n_of[q] + ni - 1) / ni);
            pt.lds_doubles = in.lds_xyz(pt.G, pt.K);
        }
    }
    out.items.clear();
    // observations in item order, written straight into the caller's staging (every observation of the window has a place)
    out.pts_i = nullptr;
    out.pts_j = fast ? nullptr : (double *)alloc(user, 2 * (size_t)std::max<int64_t>(M, 1) * 8);
    out.first = fast ? (int32_t *)alloc(user, (size_t)std::max<int64_t>(out.Ns, 1) * 4) : nullptr;
    if (fast ? !out.first : !out.pts_j) return fail(out, ERR_ALLOC, "staging allocation failed");
    out.slab_doubles = 0; out.lw_doubles = 0; out.max_lds_doubles = in.imu_item_lds;
    int64_t s = 0, obs_base = 0;
    while (s < out.Ns) {
        const int id = lm_pattern[out.sorted_to_orig[s]];
        const Pattern &pt = out.patterns[id];
        int64_t e = s;
        while (e < out.Ns && e - s < pt.G && lm_pattern[out.sorted_to_orig[e]] == id) ++e;
        ItemDesc it;
        std::memset(&it, 0, sizeof(it));
        it.lm_base = (int32_t)s; it.G = (int32_t)(e - s); it.K = pt.K; it.nb = pt.nb; it.host = -1;
        it.host_slot = -1; it.use_ext = 0; it.obs_base = (int32_t)obs_base;
        it.out_base = (int32_t)out.slab_doubles; it.lw_base = (int32_t)out.lw_doubles;
        std::memcpy(it.target, pt.target, sizeof(it.target)); std::memcpy(it.tslot, pt.tslot, sizeof(it.tslot));
        std::memcpy(it.cam_block, pt.cam_block, sizeof(it.cam_block));
        for (int p = 0; p < pt.nb; ++p) { it.btype[p] = 2; it.bk[p] = (int8_t)p; }
        it.n_rows = pt.n_rows; it.lds_doubles = pt.lds_doubles;
        out.items.push_back(it);
        out.max_lds_doubles = std::max(out.max_lds_doubles, pt.lds_doubles);
        out.slab_doubles += (size_t)item_out_count(pt.nb);
        out.slab_doubles = (out.slab_doubles + 1) & ~(size_t)1;
        out.lw_doubles += (size_t)9 * it.G;            // H_ll (6), b_l (3): W is formed again where it is needed
        for (int g = 0; g < it.G; ++g) {
            const int32_t l = out.sorted_to_orig[s + g];
            if (fast) { out.first[s + g] = (int32_t)obs_off[l]; continue; }      // (marg: the frame-0 observation is the landmark's first)
            for (int k = 0; k < it.K; ++k) {
                const int32_t oe = obs_at[(size_t)l * NF + (pt.cam_block[k] - 1)];
                const int64_t o = obs_base + (int64_t)k * it.G + g;
                out.pts_j[2 * o] = in.pts_j[2 * oe]; out.pts_j[2 * o + 1] = in.pts_j[2 * oe + 1];
            }
        }
        obs_base += (int64_t)it.G * it.K;
        s = e;
    }
    out.Ms = obs_base;
    return true;
}

void build_reduce_lists(const std::vector<ItemDesc> &items, std::vector<int32_t> &list_off, std::vector<int32_t> &list) {
    // (two passes over the items — sizes, then entries at their places — instead of 91 growing vectors: a list's entries in item order, as before;
    //  the vectors' allocations were a third of a small window's plan upload)
    const int n_lists = VIO_NPAIR + VIO_NCB + 1;
    int32_t count[n_lists];
    for (int b = 0; b < n_lists; ++b) count[b] = 0;
    for (const ItemDesc &it : items) {
        for (int p = 0; p < it.nb; ++p) {
            const int P = it.cam_block[p];
            for (int q = p; q < it.nb; ++q) {
                const int Q = it.cam_block[q];
                ++count[P * VIO_NCB - P * (P - 1) / 2 + (Q - P)];
            }
            count[VIO_NPAIR + P] += 2;
        }
        ++count[n_lists - 1];
    }
    list_off.assign(n_lists + 1, 0);
    for (int b = 0; b < n_lists; ++b) list_off[b + 1] = list_off[b] + count[b];
    list.resize((size_t)list_off[n_lists]);
    int32_t at[n_lists];
    for (int b = 0; b < n_lists; ++b) at[b] = list_off[b];
    for (const ItemDesc &it : items) {
        for (int p = 0; p < it.nb; ++p) {
            const int P = it.cam_block[p];
            for (int q = p; q < it.nb; ++q) {
                const int Q = it.cam_block[q];
                list[at[P * VIO_NCB - P * (P - 1) / 2 + (Q - P)]++] = it.out_base + item_pair_index(it.nb, p, q) * 36;
            }
            list[at[VIO_NPAIR + P]++] = it.out_base + item_nbp(it.nb) * 36 + p * 6;
            list[at[VIO_NPAIR + P]++] = it.nb * 6;
        }
        list[at[n_lists - 1]++] = it.out_base + it.n_rows * 6;
    }
}

}  // namespace vio_plan
