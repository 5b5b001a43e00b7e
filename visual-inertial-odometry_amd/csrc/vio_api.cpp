// vio_api.cpp — host side of libvio_hip.so: the C ABI of include/vio_backend.h over the gfx950 kernels.
//
// Mirrors the call sequence of the reference's Estimator::problemSolve / MargOldFrame / MargNewFrame
// (VM/src/estimator.cpp:693-1073) against Problem (VM/src/backend/problem.cc); see the header for the
// per-function citations.  No CPU fallback: every compute entry point launches HIP kernels and fails with
// VIO_ERR_HIP / VIO_ERR_NO_DEVICE when it cannot.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/vio_backend.h"
#include "host_dense.h"
#include "vio_plan.h"
#include "vio_types.h"

struct ReduceTables {
    const int32_t *list_off;
    const int32_t *list;
    const double *slab;
    double *vis;
    const double *step_part;
    int32_t n_step;
    int32_t gate;
    const LmState *lm;
    const double *jtinv;
    const double *bprior;
    double *errprior;
    int32_t lm_loop;             // 1 (vio_solve's loop): step_part / jtinv count only while lm->pending; bprior / errprior are the bases of
                                 // the two copies and the step's copy is lm->cur ^ 1
};
void vio_launch_errprior(const DeviceTables &T, hipStream_t s);
void vio_launch_prepare(const DeviceTables &T, hipStream_t s);
void vio_launch_linearize(const DeviceTables &T, int n_blocks, size_t lds_bytes, int threads, int use_ext, hipStream_t s);
void vio_launch_reduce(const ReduceTables &R, hipStream_t s);
void vio_launch_reduce_assemble(const ReduceTables &R, const DeviceTables &T, hipStream_t s);
void vio_launch_assemble(const DeviceTables &T, hipStream_t s);
void vio_launch_gather_obs(const ItemDesc *items, int n_items, const int32_t *first, const int32_t *obs_idx, const double *raw, double *out, hipStream_t s);
void vio_launch_gather_landmarks(const LmState *lm, const double *src, int ns_src, double *dst, int ns_dst, const int32_t *map, hipStream_t s);
void vio_launch_pose_solve(const DeviceTables &T, size_t lds_bytes, hipStream_t s);
void vio_launch_backsub(const DeviceTables &T, int mode, hipStream_t s);
void vio_launch_step_sum(const DeviceTables &T, int mode, hipStream_t s);
void vio_launch_triangulate(const TriTables &Q, hipStream_t s);
void vio_launch_lm_decide(const DeviceTables &T, int mode, int sum_local, hipStream_t s);
void vio_launch_init_lm(const DeviceTables &T, int max_iter, hipStream_t s);
void vio_launch_set_lambda(LmState *lm, double lambda, hipStream_t s);
void vio_launch_flip(LmState *lm, hipStream_t s);
void vio_launch_batch_lm(const DeviceTables *tabs, int B, int lm_dim, int max_blocks, size_t lin_lds, int lin_threads, int any_prior, size_t ps_lds,
                         int what, int max_iter, int order, hipStream_t s);
void vio_launch_batch_gn(const DeviceTables *tabs, int B, int lm_dim, int max_blocks, size_t lin_lds, int lin_threads, int test_prev, int any_prior,
                         int parity, size_t ps_lds, int order, hipStream_t s, int ev_kernel, hipEvent_t *ev);
void vio_launch_chain_pre(const DeviceTables &T, hipStream_t s);
void vio_launch_prior_simg(const DeviceTables &T, hipStream_t s);
int vio_chain_s_doubles();
int vio_chain_pre_lds_doubles();
int vio_chain_prior_flags();
void vio_chain_imu_map(uint32_t *out);
void vio_launch_chain_solve_test(const double *img, double lambda, double *x_nat, double *lds_dump, hipStream_t s);
int vio_chain_image_doubles();
int vio_chain_y_offset();
int vio_chain_lds_core_doubles();
void vio_chain_entry_pos(int i, int j, int *p1, int *p2);
int vio_chain_dim(int i);
int vio_set_kernel_attributes();
int lin_lds_doubles_host(int G, int K, int nb, int use_ext);
int lin_threads_host();
int lin_threads_half_host();
int xyz_lds_doubles_host(int G, int K);

namespace {

constexpr int NF = VIO_NUM_FRAMES, PD = VIO_POSE_DIM, PRD = VIO_PRIOR_DIM;
constexpr int LDS_BUDGET_DOUBLES = (160 * 1024 - 512) / 8;   // per linearize workgroup: 160 KiB per CU on gfx950, 112 bytes of it static
constexpr int LDS_BUDGET_HALF_DOUBLES = (80 * 1024 - 512) / 8;   // two workgroups to a CU (k_linearize_h: plans of the throughput policy)
constexpr int POSE_SOLVE_TILED = 66 * 272 + 192;   // PS_PACKED of vio_kernels.hip: 66 tiles of 16x17 + the rhs row
constexpr int POSE_SOLVE_LDS = (POSE_SOLVE_TILED + 176 + 272 + 272 + 192 + 176 + 112 + 176 + 184) * 8 + 176 * 4 + 64;

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool view = false;          // a piece of another allocation (alloc_fixed's frame block): never freed, never grown
    void view_of(void *base, size_t count) { p = (T *)base; n = count; view = true; }
    hipError_t resize(size_t count) {
        if (count == 0) count = 1;
        if (count <= n) return hipSuccess;
        // hipFree waits for the whole device: a buffer that has had to grow once (the next frame's graph is a little larger than
        // this one's) gets a quarter of headroom, so that a stream of frames stops reallocating after its first few
        // (a view that has to grow becomes an allocation of its own: the block it was a piece of is not this buffer's to free)
        if (p && !view) { hipFree(p); count += count / 4; }
        p = nullptr; n = 0; view = false;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() { if (p && !view) hipFree(p); p = nullptr; n = 0; view = false; }
};

// Pinned staging for the uploads of a frame (plan tables, observations, states, prior): the copies out of it are truly
// asynchronous, so activate() does not wait for the device at all (two hipStreamSynchronize per activation before: ~0.2 ms
// of a 2.6 ms frame).  begin() waits for the copies of the previous use (long done in practice), then hands out pieces;
// a piece that does not fit opens a new chunk, and the next begin() makes one chunk of the total.
struct HostArena {
    struct Chunk { char *p; size_t cap, off; };
    std::vector<Chunk> chunks;
    hipEvent_t ev = nullptr;
    bool pending = false;
    hipError_t begin() {
        if (pending) { hipError_t e = hipEventSynchronize(ev); if (e != hipSuccess) return e; pending = false; }
        if (chunks.size() > 1) {
            size_t tot = 0;
            for (Chunk &c : chunks) { tot += c.cap; hipHostFree(c.p); }
            chunks.clear();
            Chunk c{nullptr, tot + tot / 4, 0};
            hipError_t e = hipHostMalloc((void **)&c.p, c.cap, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            chunks.push_back(c);
        }
        for (Chunk &c : chunks) c.off = 0;
        return hipSuccess;
    }
    void *alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (chunks.empty() || chunks.back().off + bytes > chunks.back().cap) {
            Chunk c{nullptr, std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20), 0};
            if (hipHostMalloc((void **)&c.p, c.cap, hipHostMallocDefault) != hipSuccess) return nullptr;
            chunks.push_back(c);
        }
        Chunk &c = chunks.back();
        void *r = c.p + c.off;
        c.off += bytes;
        return r;
    }
    template <typename T> T *put(const T *src, size_t count) {
        T *d = (T *)alloc(std::max<size_t>(count, 1) * sizeof(T));
        if (d && count) std::memcpy(d, src, count * sizeof(T));
        return d;
    }
    hipError_t end(hipStream_t st) {
        if (!ev) { hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); if (e != hipSuccess) return e; }
        hipError_t e = hipEventRecord(ev, st);
        pending = e == hipSuccess;
        return e;
    }
    void release(bool stream_alive) {
        if (pending && ev && stream_alive) hipEventSynchronize(ev);       // (otherwise the caller has waited for the device)
        for (Chunk &c : chunks) hipHostFree(c.p);
        chunks.clear();
        if (ev) hipEventDestroy(ev);
        ev = nullptr; pending = false;
    }
};

// A host mirror the DMA engine reads directly: the observations of a window are uploaded as the caller listed them, straight out
// of this buffer, and put into item order by a kernel (k_gather_obs) instead of by a host loop over 80 000 observations.
struct PinnedVec {
    double *p = nullptr;
    size_t n = 0, cap = 0;
    const double *data() const { return p; }
    size_t size() const { return n; }
    double operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    bool resize_uninitialized(size_t k) {
        if (k > cap) {
            double *q = nullptr;
            const size_t c2 = k + k / 4;
            if (hipHostMalloc((void **)&q, std::max<size_t>(c2, 1) * 8, hipHostMallocDefault) != hipSuccess) return false;
            if (p) hipHostFree(p);
            p = q; cap = c2;
        }
        n = k;
        return true;
    }
    bool assign(const double *a, const double *b) {
        const size_t k = (size_t)(b - a);
        if (k > cap) {
            double *q = nullptr;
            const size_t c2 = k + k / 4;
            if (hipHostMalloc((void **)&q, std::max<size_t>(c2, 1) * 8, hipHostMallocDefault) != hipSuccess) return false;
            if (p) hipHostFree(p);
            p = q; cap = c2;
        }
        if (k) std::memcpy(p, a, k * 8);
        n = k;
        return true;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; n = cap = 0; }
};

using vio_plan::Pattern;

// Everything that depends on the graph topology (which landmark is seen from where).
struct Plan {
    bool valid = false;
    int marg = 0, use_ext = 0, lm_dim = 1;
    int64_t Ns = 0, Ms = 0;
    std::vector<int32_t> sorted_to_orig;       // landmark permutation
    std::vector<ItemDesc> items;
    std::vector<Pattern> patterns;
    std::vector<int32_t> list_off, list;
    size_t slab_doubles = 0, lw_doubles = 0;
    int max_lds_doubles = 0;
    int lin_threads = 0;                       // k_linearize's workgroup width for this plan: lin_threads_host(), or lin_threads_half_host() (two workgroups to a CU)
    DevBuf<ItemDesc> d_items;
    DevBuf<int32_t> d_list_off, d_list, d_first, d_obs_idx;      // d_first / d_obs_idx: k_gather_obs's view of the caller's observation list
    DevBuf<char> d_tables;      // d_items, d_list_off, d_list, d_first and (inverse depth) d_pts_i are views of it: one upload per plan (upload_plan)
    DevBuf<double> d_pts_i, d_pts_j, d_invd, d_slab, d_lw, d_dxl, d_step_part;
    void release() {
        d_items.release(); d_list_off.release(); d_list.release(); d_first.release(); d_obs_idx.release(); d_tables.release();
        d_pts_i.release(); d_pts_j.release(); d_invd.release(); d_slab.release(); d_lw.release(); d_dxl.release();
        d_step_part.release();
        valid = false;
    }
};

}  // namespace

struct MargResult {                 // what the helper thread of vio_marginalize_begin leaves for vio_marginalize_end
    std::vector<double> H, b, err, jt;
    int kind = 0, live_rows = 0;
    bool finite = true;
    double tail_us = 0;
};

struct vio_ctx {
    vio_config cfg;
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // host mirrors of the inputs (original landmark order)
    double h_state[STATE_STRIDE];
    int lm_dim = 1;                            // 1: inverse depths; 3: XYZ points (h_invd [N][3], h_otarget = observing frame, h_pts_j = observation)
    std::vector<double> h_invd, h_pts_i, h_pts_i_lm;      // h_pts_i_lm [N][2]: the host observation by landmark (lists vio_set_observations vouches for)
    PinnedVec h_pts_j;                         // pinned: uploaded as it is (raw_pts_valid: the device copy d_raw_pts_j is current)
    bool raw_pts_valid = false;
    std::vector<int32_t> h_olm, h_ohost, h_otarget;   // observation -> landmark / host frame / target frame
    bool imu_valid[VIO_WINDOW_SIZE];
    std::vector<double> h_pre;                 // [10][PRE_STRIDE]
    int has_prior = 0;
    std::vector<double> h_bprior, h_errprior;
    PinnedVec h_Hprior, h_Jtinv;               // pinned: the two matrices of the prior (430 KB) go up as they lie here (round 6: they were copied into the staging arena first)
    HostArena arena;                           // pinned staging of the uploads
    double *marg_stage = nullptr;              // pinned: H_marg (171 x 171) and b_marg for the host tail of vio_marginalize
    vio_plan::BgTicket marg_ticket;            // the dense tail of a marginalisation on the process's background worker (vio_marginalize_begin / _end)
    int marg_job_frame = 0;
    bool shared_ref = false;                   // this context holds a reference on the process's helper threads (vio_plan::shared_acquire)
    bool marg_pending = false;
    MargResult marg_out;
    double *pull_stage = nullptr;              // pinned staging of the landmark read-back (pull_from_device)
    size_t pull_cap = 0;
    bool prior_dirty = true, imu_dirty = true; // h_Hprior / h_Jtinv resp. h_pre newer than their device copies
    // dirty tracking
    bool dirty_inputs = true;                  // host mirrors newer than the device
    unsigned ahead = 0;                        // what the device holds newer than the host mirrors: 1 states, 2 landmarks, 4 b_prior / err_prior
    bool topo_dirty = true;
    bool obs_mapped = false;                   // between vio_map_observations and vio_commit_observations: no list
    bool obs_map_stale = false;                // ... and vio_set_landmarks has changed the landmark count meanwhile: the mapped arrays are gone
    bool obs_consistent = false;               // ... and vio_set_observations has seen that they share host frame and host observation
    bool obs_lm_major = false;                 // the observations of a landmark are consecutive and the landmarks ascend (vio_set_observations)
    bool linearized = false;
    bool lin_fresh = false;                    // vio_linearize was the last call that touched the device state: vio_solve starts from its system
    hipEvent_t lin_ev[2] = {nullptr, nullptr}; // vio_solve's timing of the first linearisation (vio_solve_report.hessian_ms)
    float last_lin_ms = 0;                     // ... the last such timing: reported when the linearisation is already there
    bool pairtab_valid = false;
    bool stepwise_updated = false;
    double gn_lambda = -1.0;
    int solve_order = VIO_ORDER_CHAIN;         // vio_set_solve_order (VIO_SOLVE_ORDER=eigen|chain overrides the default at creation)
    bool prior_chain_ok = true;                // H_prior couples no two speed-bias blocks that are not neighbours (the chain order's storage)
    bool test_in_solve = false;                // three-launch path: the step test owed by the last k_reduce_c goes to the next k_pose_solve_c
    int gn_split = 0;                          // the GN iteration being enqueued has its speed-bias chain pre-eliminated: k_pose_solve_cs follows
    int pg_layout = -1;                        // which order's image d_Pg holds (the two layouts rely on different never-written zeros)
    bool want_natural_hs = false;              // set by vio_get_schur_system: re-run k_assemble with the natural-order copy
    bool natural_hs_valid = false;
    int g_max = 0;                             // landmarks per item; 0 = automatic (VIO_G_MAX overrides; <= 128: k_backsub has one thread per landmark)
    int g_min = 8;                             // lower end of the automatic choice (VIO_G_MIN overrides)
    int n_cus = 256;                           // CUs of the device (one k_linearize workgroup each)
    Plan solve_plan, marg_plan;
    bool marg_map_valid = false;          // d_gather_map holds the marg plan's landmarks as positions in the solve plan (both plans as they are)
    Plan *active = nullptr;
    // device buffers independent of the topology
    DevBuf<double> d_state, d_pairtab, d_vis, d_pre, d_imu_out, d_Hprior, d_bprior, d_errprior, d_Jtinv, d_Hs, d_bs,
        d_bfull, d_diagfull, d_dx, d_step_tot, d_imu_chi, d_gath, d_step_gath, d_sp_part, d_cfi;
    DevBuf<int32_t> d_imu_valid, d_perm, d_rank, d_gather_map;
    DevBuf<uint32_t> d_imu_map;                // static: where an IMU item's elements go in the chain image (d_chain_pre_item)
    DevBuf<double> d_prior_simg;               // H_prior's speed-bias rows at their places in the chain image
    DevBuf<int32_t> d_prior_flags, d_prior_list;     // ... which tiles / rows of it hold a non-zero, and its non-zero entries as a list
    DevBuf<double> d_prior_cval;
    bool prior_simg_valid = false;
    DevBuf<double> d_raw_pts_j;                // the target observations in the caller's order (k_gather_obs reads them)
    DevBuf<double> d_Pg;
    DevBuf<LmState> d_lm;
    LmState h_lm;
    LmState *h_lm_pin = nullptr;               // pinned landing place of LmState's read-back (the copy is then a DMA the host need not wait in)
    hipEvent_t lm_event = nullptr;             // recorded behind that copy: read_lm_end waits for it, not for what was enqueued after it
    vio_exchange_fn hook = nullptr;
    void *comm = nullptr;                              // ncclComm_t of the native exchange (vio_comm_init)
    int cur_host = -1;                                 // LmState.cur as the host tracks it through GN iterations (-1: unknown)
    uint64_t tables_gen = 1;                           // bumped whenever what make_tables_raw produces may have changed
    // a batch this context leads (vio_batch_gn_iteration): the members' tables as a device array + what it was built from
    std::vector<vio_ctx *> batch_members;
    std::vector<uint64_t> batch_gens;
    std::vector<int> batch_cur0;                       // every member's LmState.cur when the array was built
    DevBuf<DeviceTables> d_batch_tabs;
    DevBuf<char> d_frame_block;                // d_state, d_bprior, d_errprior, d_lm, d_imu_valid are views of it (alloc_fixed)
    int batch_iters = 0;                               // iterations since the array was built (its parity flips every window's cur)
    vio_status flush_status = VIO_OK;                  // what the flush_decide inside the last make_tables returned
    bool decide_pending = false;                       // GN mode: the last step's test has not run yet (k_assemble of the next iteration does it)
    void *hook_user = nullptr;
    double hessian_ms = 0;
    double *ext_vis = nullptr, *ext_step = nullptr;    // caller-owned exchange buffers (vio_bind_exchange_buffers)
    double *ext_gath = nullptr, *ext_step_gath = nullptr;      // caller-owned receive buffers of the all-gather (vio_bind_gather_buffers)
    bool gather_known = false;                         // the caller has asked for / bound the receive buffers: its hook is an all-gather
    DevBuf<unsigned long long> d_dbg;                  // diagnostic builds only
    double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};        // vio_get_host_timing
    int prof_which = -1;
    int prof_every = 1, prof_seen = 0;                 // event pairs around every prof_every-th launch only
    std::vector<hipEvent_t> prof_events;               // pairs
    size_t prof_used = 0;
};

namespace {

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            c->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                 \
            return VIO_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)
#define VIOCHK(expr)                                    \
    do {                                                \
        vio_status s__ = (expr);                        \
        if (s__ != VIO_OK) return s__;                  \
    } while (0)

// every entry point: the context's device, and whatever error an earlier call of anybody's left on this thread forgotten (the
// launch checks below ask hipGetLastError(), which would report it as theirs)
// keep_fresh: entry points that leave the device's state and system as they are (vio_synchronize, the profiling calls)
static inline void enter_device(const vio_ctx *c, bool keep_fresh = false) {
    (void)hipSetDevice(c->cfg.device);
    (void)hipGetLastError();
    if (!keep_fresh) const_cast<vio_ctx *>(c)->lin_fresh = false;
}

vio_status fail(vio_ctx *c, vio_status s, const std::string &msg) {
    c->err = msg;
    return s;
}

// ---- topology preprocessing ----------------------------------------------------------------------------
// The planner itself is host-only code (csrc/vio_plan.cpp: patterns, counting sort, item sizing, items; CPU-tier tests and sanitizers
// reach it without a device); here: its inputs out of the context, its staging out of the pinned arena, its outputs uploaded.
vio_status upload_plan(vio_ctx *c, Plan &pl, const double *pts_i, const double *pts_j, const int32_t *first = nullptr, const int32_t *obs_idx = nullptr, int64_t n_obs_idx = 0);

static void *arena_alloc(void *user, size_t bytes) { return ((HostArena *)user)->alloc(bytes); }
static int lds_fn(int G, int K, int nb, int use_ext) { return lin_lds_doubles_host(G, K, nb, use_ext); }
static int lds_xyz_fn(int G, int K) { return xyz_lds_doubles_host(G, K); }

vio_status build_plan(vio_ctx *c, Plan &pl, int marg) {
    c->marg_map_valid = false;
    pl.valid = false;
    pl.marg = marg; pl.lm_dim = c->lm_dim;
    const bool xyz = c->lm_dim == 3;
    pl.use_ext = xyz ? 0 : (marg ? 1 : (c->cfg.ext_fixed ? 0 : 1));
    // the throughput policy's plans run on k_linearize_h (k_linearize_xyz_h): half the threads, half the LDS, two workgroups to a CU
    const bool half = c->cfg.item_policy == VIO_ITEMS_THROUGHPUT && !std::getenv("VIO_NO_HALF_WIDTH");
    pl.lin_threads = half ? lin_threads_half_host() : lin_threads_host();
    const int64_t M = (int64_t)c->h_olm.size();
    static const bool timing = std::getenv("VIO_HOST_TIMING") != nullptr;
    // the observations leave for the device as the caller listed them, now: the copy runs under the host work below
    if (M && !c->raw_pts_valid && (!xyz || c->obs_lm_major)) {
        HIPCHK(c->d_raw_pts_j.resize(2 * (size_t)M));
        HIPCHK(hipMemcpyAsync(c->d_raw_pts_j.p, c->h_pts_j.data(), 2 * (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
        c->raw_pts_valid = true;
    }
    vio_plan::Input in;
    in.N = (int64_t)c->h_invd.size() / c->lm_dim; in.M = M;
    in.pool = in.M >= 16384 ? vio_plan::shared_pool() : nullptr;      // (the process's helper threads: windows long enough for pieces to pay)
    in.olm = c->h_olm.data(); in.ohost = c->h_ohost.data(); in.otarget = c->h_otarget.data();
    in.pts_i = c->h_pts_i.empty() ? nullptr : c->h_pts_i.data();
    in.pts_i_lm = c->h_pts_i_lm.empty() ? nullptr : c->h_pts_i_lm.data();
    in.pts_j = c->h_pts_j.data();
    in.lm_major = c->obs_lm_major; in.vouched = c->obs_consistent;
    in.marg = marg; in.use_ext = pl.use_ext;
    in.throughput = c->cfg.item_policy == VIO_ITEMS_THROUGHPUT;
    in.g_max = c->g_max; in.g_min = c->g_min; in.n_cus = c->n_cus;
    in.lin_threads = pl.lin_threads; in.lin_threads_full = lin_threads_host();
    in.lds_budget = half ? LDS_BUDGET_HALF_DOUBLES : LDS_BUDGET_DOUBLES;
    in.lds = lds_fn; in.lds_xyz = lds_xyz_fn;
    in.imu_item_lds = IMU_ITEM_LDS_DOUBLES;
    vio_plan::Output out;
    const bool ok = xyz ? vio_plan::plan_xyz(in, out, arena_alloc, &c->arena) : vio_plan::plan_invdepth(in, out, arena_alloc, &c->arena);
    if (!ok) return fail(c, (vio_status)out.status, out.status == VIO_ERR_HIP ? "hipHostMalloc (staging)" : out.err);
    pl.patterns.swap(out.patterns); pl.items.swap(out.items); pl.sorted_to_orig.swap(out.sorted_to_orig);
    pl.Ns = out.Ns; pl.Ms = out.Ms; pl.slab_doubles = out.slab_doubles; pl.lw_doubles = out.lw_doubles; pl.max_lds_doubles = out.max_lds_doubles;
    if (timing) std::fprintf(stderr, "[vio host timing] build_plan: obs lists %.0f us, patterns %.0f us, sort + sizing %.0f us, items + gather %.0f us\n",
                             out.t_us[0], out.t_us[1], out.t_us[2], out.t_us[3]);
    const auto tb = std::chrono::steady_clock::now();
    vio_status st;
    if (xyz && !in.lm_major) st = upload_plan(c, pl, nullptr, out.pts_j);
    else {
        const int32_t *s_idx = out.obs_idx.empty() || in.lm_major ? nullptr : c->arena.put(out.obs_idx.data(), out.obs_idx.size());
        if (!in.lm_major && !xyz && !s_idx) return fail(c, VIO_ERR_HIP, "hipHostMalloc (staging)");
        st = upload_plan(c, pl, out.pts_i, nullptr, out.first, s_idx, (int64_t)out.obs_idx.size());
    }
    if (timing) std::fprintf(stderr, "[vio host timing] build_plan: upload_plan %.0f us of it\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb).count());
    return st;
}

// inverted lists for k_reduce, device buffers, upload: common to both kinds of landmark
// pts_j == nullptr: the observations are on the device in the caller's order (d_raw_pts_j) and k_gather_obs puts them into item order:
// first[s] = place of sorted landmark s's first observation in that list (landmark-major lists), or in obs_idx (the CSR of any other list)
vio_status upload_plan(vio_ctx *c, Plan &pl, const double *pts_i, const double *pts_j, const int32_t *first, const int32_t *obs_idx, int64_t n_obs_idx) {
    vio_plan::build_reduce_lists(pl.items, pl.list_off, pl.list);
    // upload
    const size_t ni = pl.items.size();
    const size_t ld = (size_t)pl.lm_dim;
    // The plan's tables — item descriptors, k_reduce's lists, the gather's `first`, the landmarks' host observations — are pieces of one
    // device block and leave the staging with one copy (five before: a hipMemcpyAsync costs the host 5 us whatever its size).
    const bool with_first = pl.Ms && !pts_j, with_pi = pl.Ns && pl.lm_dim == 1;
    auto up16 = [](size_t b) { return (b + 15) / 16 * 16; };
    const size_t o_items = 0, o_off = up16(o_items + std::max<size_t>(ni, 1) * sizeof(ItemDesc)), o_list = up16(o_off + pl.list_off.size() * 4),
                 o_first = up16(o_list + std::max<size_t>(pl.list.size(), 1) * 4), o_pi = up16(o_first + (with_first ? (size_t)pl.Ns * 4 : 0)),
                 tb_bytes = up16(o_pi + (with_pi ? 2 * (size_t)pl.Ns * 8 : 0));
    pl.d_items.release(); pl.d_list_off.release(); pl.d_list.release(); pl.d_first.release();      // (views: nothing is freed)
    if (with_pi || pl.d_pts_i.view) pl.d_pts_i.release();       // (an allocation of its own only where the plan's landmarks were XYZ before)
    HIPCHK(pl.d_tables.resize(tb_bytes));
    pl.d_items.view_of(pl.d_tables.p + o_items, std::max<size_t>(ni, 1)); pl.d_list_off.view_of(pl.d_tables.p + o_off, pl.list_off.size());
    pl.d_list.view_of(pl.d_tables.p + o_list, std::max<size_t>(pl.list.size(), 1));
    if (with_first) pl.d_first.view_of(pl.d_tables.p + o_first, (size_t)pl.Ns);
    if (with_pi) pl.d_pts_i.view_of(pl.d_tables.p + o_pi, 2 * (size_t)pl.Ns); else HIPCHK(pl.d_pts_i.resize(2 * (size_t)pl.Ns));
    HIPCHK(pl.d_pts_j.resize(2 * (size_t)pl.Ms));
    HIPCHK(pl.d_invd.resize(2 * ld * (size_t)std::max<int64_t>(pl.Ns, 1))); HIPCHK(pl.d_slab.resize(pl.slab_doubles));
    HIPCHK(pl.d_lw.resize(2 * pl.lw_doubles)); HIPCHK(pl.d_dxl.resize(ld * (size_t)pl.Ns));      // lw: two sets, see DeviceTables.lw_set
    HIPCHK(pl.d_step_part.resize(4 * (ni + VIO_WINDOW_SIZE)));
    hipStream_t st = c->stream;
    // everything leaves from the pinned staging: no wait here (activate() marks the staging busy until these copies are done)
    char *s_tb = (char *)c->arena.alloc(tb_bytes);
    if (!s_tb) return fail(c, VIO_ERR_HIP, "hipHostMalloc (staging)");
    if (ni) std::memcpy(s_tb + o_items, pl.items.data(), ni * sizeof(ItemDesc));
    std::memcpy(s_tb + o_off, pl.list_off.data(), pl.list_off.size() * 4);
    if (!pl.list.empty()) std::memcpy(s_tb + o_list, pl.list.data(), pl.list.size() * 4);
    if (with_first) std::memcpy(s_tb + o_first, first, (size_t)pl.Ns * 4);
    if (with_pi) std::memcpy(s_tb + o_pi, pts_i, 2 * (size_t)pl.Ns * 8);
    HIPCHK(hipMemcpyAsync(pl.d_tables.p, s_tb, tb_bytes, hipMemcpyHostToDevice, st));
    if (pl.Ms && pts_j) HIPCHK(hipMemcpyAsync(pl.d_pts_j.p, pts_j, 2 * (size_t)pl.Ms * 8, hipMemcpyHostToDevice, st));
    else if (pl.Ms) {
        if (obs_idx) {
            HIPCHK(pl.d_obs_idx.resize((size_t)n_obs_idx));
            HIPCHK(hipMemcpyAsync(pl.d_obs_idx.p, obs_idx, (size_t)n_obs_idx * 4, hipMemcpyHostToDevice, st));
        }
        vio_launch_gather_obs(pl.d_items.p, (int)ni, pl.d_first.p, obs_idx ? pl.d_obs_idx.p : nullptr, c->d_raw_pts_j.p, pl.d_pts_j.p, st);
        HIPCHK(hipGetLastError());
    }
    pl.valid = true;
    return VIO_OK;
}

// ---- device state management ---------------------------------------------------------------------------
// The small per-frame inputs — states, b_prior, err_prior (two copies each), LmState, the IMU edges' valid flags — are pieces of ONE device
// block in this order, so that a frame uploads them with one copy instead of five (a hipMemcpyAsync costs the host 5 us whatever its size).
constexpr size_t FB_STATE = 0, FB_BPRIOR = FB_STATE + 2 * STATE_STRIDE * 8, FB_ERRPRIOR = FB_BPRIOR + 2 * 176 * 8, FB_LM = FB_ERRPRIOR + 2 * 160 * 8,
                 FB_IV = (FB_LM + sizeof(LmState) + 15) / 16 * 16, FB_BYTES = FB_IV + 16 * 4;
vio_status alloc_fixed(vio_ctx *c) {
    if (!c->d_frame_block.p) {
        HIPCHK(c->d_frame_block.resize(FB_BYTES));
        HIPCHK(hipMemsetAsync(c->d_frame_block.p, 0, FB_BYTES, c->stream));
        c->d_state.view_of(c->d_frame_block.p + FB_STATE, 2 * STATE_STRIDE); c->d_bprior.view_of(c->d_frame_block.p + FB_BPRIOR, 2 * 176);
        c->d_errprior.view_of(c->d_frame_block.p + FB_ERRPRIOR, 2 * 160); c->d_lm.view_of(c->d_frame_block.p + FB_LM, 1);
        c->d_imu_valid.view_of(c->d_frame_block.p + FB_IV, 16);
    }
    HIPCHK(c->d_state.resize(2 * STATE_STRIDE)); HIPCHK(c->d_pairtab.resize(2 * PAIRTAB_STRIDE));
    HIPCHK(c->d_vis.resize(VIS_COUNT)); HIPCHK(c->d_pre.resize(VIO_WINDOW_SIZE * PRE_STRIDE));
    HIPCHK(c->d_imu_out.resize(VIO_WINDOW_SIZE * IMU_OUT)); HIPCHK(c->d_Hprior.resize(PD * PD));
    HIPCHK(c->d_bprior.resize(2 * 176)); HIPCHK(c->d_errprior.resize(2 * 160)); HIPCHK(c->d_Jtinv.resize(PRD * PRD));
    HIPCHK(c->d_Hs.resize(PD * PD)); HIPCHK(c->d_bs.resize(176)); HIPCHK(c->d_bfull.resize(2 * 176));
    HIPCHK(c->d_diagfull.resize(176)); HIPCHK(c->d_dx.resize(176)); HIPCHK(c->d_step_tot.resize(8)); HIPCHK(c->d_sp_part.resize(8));
    HIPCHK(c->d_imu_chi.resize(16)); HIPCHK(c->d_imu_valid.resize(16)); HIPCHK(c->d_lm.resize(1));
    HIPCHK(c->d_gath.resize((size_t)c->cfg.shard_count * VIS_SEND)); HIPCHK(c->d_step_gath.resize((size_t)c->cfg.shard_count * 2));
    HIPCHK(c->d_perm.resize(2 * 176)); HIPCHK(c->d_rank.resize(176)); HIPCHK(c->d_Pg.resize(2 * POSE_SOLVE_TILED));     // two sets (vio_solve's loop)
    {
        std::vector<uint32_t> map(10 * 63 * 9);
        vio_chain_imu_map(map.data());
        HIPCHK(c->d_imu_map.resize(map.size()));
        HIPCHK(hipMemcpy(c->d_imu_map.p, map.data(), map.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c->d_prior_simg.resize((size_t)vio_chain_s_doubles()));
        HIPCHK(c->d_prior_flags.resize((size_t)vio_chain_prior_flags()));
        HIPCHK(c->d_prior_list.resize(128 + (size_t)vio_chain_s_doubles())); HIPCHK(c->d_prior_cval.resize((size_t)vio_chain_s_doubles()));
    }
    HIPCHK(c->d_cfi.resize((size_t)vio_chain_lds_core_doubles()));
    HIPCHK(hipMemsetAsync(c->d_cfi.p, 0, (size_t)vio_chain_lds_core_doubles() * 8, c->stream));
    HIPCHK(hipMemset(c->d_Pg.p, 0, 2 * POSE_SOLVE_TILED * sizeof(double)));   // tile padding (17th column) is never written again (use_pg_layout)
    HIPCHK(hipMemsetAsync(c->d_vis.p, 0, VIS_COUNT * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_step_tot.p, 0, 8 * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_dx.p, 0, 176 * 8, c->stream));
    return VIO_OK;
}

// The chain order (vio_pose_solve_chain.h) stores the speed-bias blocks as a block-tridiagonal chain; a prior that couples two
// blocks that are not neighbours (no reference caller produces one: Marginalize reaches the speed-bias of frame 1 only) is
// solved in Eigen's pivot order instead.
inline int effective_order(const vio_ctx *c) {
    return (c->solve_order == VIO_ORDER_CHAIN && (!c->has_prior || c->prior_chain_ok)) ? VIO_ORDER_CHAIN : VIO_ORDER_EIGEN;
}

// Both images of the pose system keep padding that no kernel ever writes (the 17th column of a tile, the identity rows, the unused
// rows of the chain's right-hand side): cleared when the buffer changes hands, in stream order
inline void use_pg_layout(vio_ctx *c, int order) {
    if (c->pg_layout == order) return;
    (void)hipMemsetAsync(c->d_Pg.p, 0, 2 * (size_t)POSE_SOLVE_TILED * sizeof(double), c->stream);
    c->pg_layout = order;
}

DeviceTables make_tables_raw(vio_ctx *c, Plan &pl) {
    DeviceTables T;
    std::memset(&T, 0, sizeof(T));
    T.items = pl.d_items.p; T.n_items = (int32_t)pl.items.size(); T.n_imu_items = VIO_WINDOW_SIZE;
    T.Ns = (int32_t)pl.Ns; T.lm_dim = pl.lm_dim; T.ext_fixed = c->cfg.ext_fixed; T.loss_type = c->cfg.loss_type; T.marg_mode = pl.marg;
    T.loss_delta = c->cfg.loss_delta; T.sqrt_info = c->cfg.reproj_sqrt_info;
    for (int k = 0; k < 3; ++k) T.gravity[k] = c->cfg.gravity[k];
    T.state = c->d_state.p; T.invd = pl.d_invd.p; T.pts_i = pl.d_pts_i.p; T.pts_j = pl.d_pts_j.p;
    T.pairtab = c->d_pairtab.p; T.slab = pl.d_slab.p; T.lw = pl.d_lw.p; T.lw_set = (int64_t)pl.lw_doubles; T.vis = c->ext_vis ? c->ext_vis : c->d_vis.p; T.pre = c->d_pre.p;
    T.imu_valid = c->d_imu_valid.p; T.imu_out = c->d_imu_out.p; T.imu_chi_try = c->d_imu_chi.p;
    T.pair_slot = nullptr; T.blk_slot = nullptr;
    T.Hprior = c->d_Hprior.p; T.bprior = c->d_bprior.p; T.errprior = c->d_errprior.p; T.Jtinv = c->d_Jtinv.p;
    // Problem always carries a 171x171 prior block (zero before the first marginalisation); err_prior_ exists
    // only once a prior has been set (problem.cc:466,505,554)
    T.has_prior = c->has_prior; T.add_imu_prior = 1; T.natural_hs = (pl.marg || c->want_natural_hs) ? 1 : 0;
    T.Hs = c->d_Hs.p; T.Pg = c->d_Pg.p; T.perm = c->d_perm.p; T.rank = c->d_rank.p; T.bs = c->d_bs.p; T.bfull = c->d_bfull.p; T.diagfull = c->d_diagfull.p; T.dx = c->d_dx.p;
    T.dxl = pl.d_dxl.p; T.step_part = pl.d_step_part.p; T.n_step_blocks = T.n_items + T.n_imu_items;
    T.gn_flags = 0; T.cur_hint = -1; T.lm_gate = 0;
    T.imu_mask = 0;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) T.imu_mask |= (c->imu_valid[k] ? 1 : 0) << k;
    T.chi_part = pl.d_step_part.p + 2 * (size_t)T.n_step_blocks;
    T.step_tot = c->ext_step ? c->ext_step : c->d_step_tot.p; T.lm = c->d_lm.p; T.sp_part = c->d_sp_part.p;
    T.gath = c->ext_gath ? c->ext_gath : c->d_gath.p; T.step_gath = c->ext_step_gath ? c->ext_step_gath : c->d_step_gath.p;
    T.n_shards = (c->hook != nullptr || c->comm != nullptr) ? c->cfg.shard_count : 0;
    T.solve_order = effective_order(c);
    T.cfi = c->d_cfi.p; T.prior_simg = c->d_prior_simg.p; T.prior_flags = c->d_prior_flags.p; T.prior_list = c->d_prior_list.p; T.prior_cval = c->d_prior_cval.p; T.imu_map = c->d_imu_map.p;
    use_pg_layout(c, T.solve_order);
    T.list_off = pl.d_list_off.p; T.list = pl.d_list.p;
#ifdef VIO_STAMPS
    (void)c->d_dbg.resize(16 * (size_t)(T.n_items + T.n_imu_items + 48));
    T.dbg = c->d_dbg.p;
#endif
    return T;
}

vio_status run_exchange(vio_ctx *c, int which);

// GN mode leaves the chi2 evaluation and the step test of iteration i to iteration i+1 (k_linearize computes that chi2
// anyway, k_assemble runs the test); whoever touches the device outside that loop gets them done first, the classic
// way: k_backsub once more in full (the landmark update it repeats is idempotent), then k_lm_decide.
vio_status flush_decide(vio_ctx *c) {
    if (!c->decide_pending || !c->active) return VIO_OK;
    c->decide_pending = false;
    DeviceTables T = make_tables_raw(c, *c->active);
    T.gn_flags = 8;                                          // the GN step left the prior update of its trial slot undone:
    vio_launch_backsub(T, 0, c->stream);                     // b_prior' by k_backsub,
    T.gn_flags = 0;
    if (T.has_prior) vio_launch_errprior(T, c->stream);      // err_prior' by k_errprior
    if (c->hook || c->comm) {
        vio_launch_step_sum(T, 0, c->stream);
        VIOCHK(run_exchange(c, 1));
        vio_launch_lm_decide(T, 1, 0, c->stream);
    } else {
        vio_launch_lm_decide(T, 1, 1, c->stream);
    }
    return VIO_OK;
}

// tables for every path but the GN loop: bring the device's LmState up to date first; the host's idea of `cur` is void
// until the next read_lm
DeviceTables make_tables(vio_ctx *c, Plan &pl) {
    c->flush_status = flush_decide(c);      // checked by the caller right after (MAKE_TABLES)
    c->cur_host = -1;
    return make_tables_raw(c, pl);
}
#define MAKE_TABLES(T, c, pl) DeviceTables T = make_tables((c), (pl)); VIOCHK((c)->flush_status)

// LmState's read-back in two halves: whatever the host enqueues between them (vio_solve: the uploads of the next call's plan)
// runs behind the copy and is not waited for
vio_status read_lm_begin(vio_ctx *c) {
    VIOCHK(flush_decide(c));
    if (!c->h_lm_pin) HIPCHK(hipHostMalloc((void **)&c->h_lm_pin, sizeof(LmState), hipHostMallocDefault));
    if (!c->lm_event) HIPCHK(hipEventCreateWithFlags(&c->lm_event, hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(c->h_lm_pin, c->d_lm.p, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->lm_event, c->stream));
    return VIO_OK;
}
vio_status read_lm_end(vio_ctx *c) {
    // (polling the event with hipEventQuery instead: measured, no difference — 0.5205 / 0.5196 against 0.5212 / 0.5232 ms per Solve(10))
    HIPCHK(hipEventSynchronize(c->lm_event));
    c->h_lm = *c->h_lm_pin;
    c->cur_host = c->h_lm.cur;
    return VIO_OK;
}
vio_status read_lm(vio_ctx *c) {
    VIOCHK(read_lm_begin(c));
    return read_lm_end(c);
}

// bring the host mirrors up to date with the device (states, inverse depths, prior vectors)
vio_status pull_from_device(vio_ctx *c, unsigned mask = 7u) {
    const unsigned need = c->ahead & mask;
    if (!need) return VIO_OK;
    VIOCHK(read_lm(c));
    const int cur = c->h_lm.cur;
    if (need & 1u) HIPCHK(hipMemcpyAsync(c->h_state, c->d_state.p + cur * STATE_STRIDE, STATE_STRIDE * 8, hipMemcpyDeviceToHost, c->stream));
    Plan &pl = c->solve_plan;
    const size_t ld = (size_t)pl.lm_dim;
    double *tmp = nullptr;
    if ((need & 2u) && pl.valid && pl.Ns) {
        // the landmarks come back through a pinned buffer of the context's (a pageable destination is staged by the runtime)
        const size_t cnt = ld * (size_t)pl.Ns;
        if (cnt > c->pull_cap) {
            if (c->pull_stage) hipHostFree(c->pull_stage);
            c->pull_stage = nullptr; c->pull_cap = 0;
            HIPCHK(hipHostMalloc((void **)&c->pull_stage, (cnt + cnt / 4) * 8, hipHostMallocDefault));
            c->pull_cap = cnt + cnt / 4;
        }
        tmp = c->pull_stage;
        HIPCHK(hipMemcpyAsync(tmp, pl.d_invd.p + (size_t)cur * ld * pl.Ns, cnt * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if ((need & 4u) && c->has_prior) {
        HIPCHK(hipMemcpyAsync(c->h_bprior.data(), c->d_bprior.p + cur * 176, PD * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->h_errprior.data(), c->d_errprior.p + cur * 160, PRD * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (tmp)
        for (int64_t s = 0; s < pl.Ns; ++s)
            for (size_t k = 0; k < ld; ++k) c->h_invd[ld * pl.sorted_to_orig[s] + k] = tmp[k * pl.Ns + s];      // device: coordinate-major
    c->ahead &= ~need;
    return VIO_OK;
}

// upload the host mirrors into copy 0 and reset the LM state
vio_status push_to_device(vio_ctx *c, Plan &pl) {
    hipStream_t st = c->stream;
    HostArena &A = c->arena;
    const size_t ld = (size_t)pl.lm_dim;
    // states, b_prior, err_prior, LmState and the IMU flags: an image of the device's frame block (its second copies — trial state, trial
    // b_prior / err_prior: written by the kernels before anybody reads them — go out as zeros), one copy
    char *s_fb = (char *)A.alloc(FB_BYTES);
    double *s_invd = (double *)A.alloc(ld * (size_t)std::max<int64_t>(pl.Ns, 1) * 8);
    if (!s_fb || !s_invd) return fail(c, VIO_ERR_HIP, "hipHostMalloc (staging)");
    std::memset(s_fb, 0, FB_BYTES);
    std::memcpy(s_fb + FB_STATE, c->h_state, (size_t)STATE_STRIDE * 8);
    std::memcpy(s_fb + FB_BPRIOR, c->h_bprior.data(), (size_t)PD * 8);
    std::memcpy(s_fb + FB_ERRPRIOR, c->h_errprior.data(), (size_t)PRD * 8);
    std::memset(&c->h_lm, 0, sizeof(LmState));
    c->h_lm.ni = 2; c->h_lm.lambda = -1; c->h_lm.finite = 1; c->h_lm.last_chi = 1e20;
    std::memcpy(s_fb + FB_LM, &c->h_lm, sizeof(LmState));
    {
        int32_t *iv = (int32_t *)(s_fb + FB_IV);
        for (int k = 0; k < VIO_WINDOW_SIZE; ++k) iv[k] = c->imu_valid[k] ? 1 : 0;
    }
    HIPCHK(hipMemcpyAsync(c->d_frame_block.p, s_fb, FB_BYTES, hipMemcpyHostToDevice, st));
    for (int64_t s = 0; s < pl.Ns; ++s)
        for (size_t k = 0; k < ld; ++k) s_invd[k * pl.Ns + s] = c->h_invd[ld * pl.sorted_to_orig[s] + k];
    if (pl.Ns) HIPCHK(hipMemcpyAsync(pl.d_invd.p, s_invd, ld * (size_t)pl.Ns * 8, hipMemcpyHostToDevice, st));
    // the pre-integrations and the two matrices of the prior never change on the device: uploaded when the caller set them
    if (c->imu_dirty) {
        double *s_pre = A.put(c->h_pre.data(), (size_t)VIO_WINDOW_SIZE * PRE_STRIDE);
        if (!s_pre) return fail(c, VIO_ERR_HIP, "hipHostMalloc (staging)");
        HIPCHK(hipMemcpyAsync(c->d_pre.p, s_pre, VIO_WINDOW_SIZE * PRE_STRIDE * 8, hipMemcpyHostToDevice, st));
        c->imu_dirty = false;
    }
    if (c->prior_dirty) {
        // (straight out of the pinned mirrors; vio_set_prior waits for the arena's event — recorded behind these copies — before it writes them again)
        HIPCHK(hipMemcpyAsync(c->d_Hprior.p, c->h_Hprior.p, PD * PD * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c->d_Jtinv.p, c->h_Jtinv.p, PRD * PRD * 8, hipMemcpyHostToDevice, st));
        c->prior_dirty = false;
        c->prior_simg_valid = false;
    }
    c->decide_pending = false; c->cur_host = 0;      // a fresh LmState: nothing of the old one is owed
    return VIO_OK;
}

// make `pl` the active plan with the host mirrors uploaded
vio_status activate(vio_ctx *c, Plan &pl, int marg) {
    if (c->obs_mapped) return fail(c, VIO_ERR_BAD_ARG, "vio_map_observations without vio_commit_observations: the context holds no observation list");
    const bool need_build = !pl.valid || c->topo_dirty;
    const bool switching = c->active != &pl;
    if (need_build || switching || c->dirty_inputs) {
        static const bool timing = std::getenv("VIO_HOST_TIMING") != nullptr;      // diagnostic: where a frame's host time goes
        const auto t0 = std::chrono::steady_clock::now();
        VIOCHK(pull_from_device(c));
        HIPCHK(c->arena.begin());                   // (waits for the copies of the previous activation: long done)
        const auto t1 = std::chrono::steady_clock::now();
        if (need_build) {
            if (c->topo_dirty) { c->solve_plan.valid = false; c->marg_plan.valid = false; c->topo_dirty = false; }
            VIOCHK(build_plan(c, pl, marg));
        }
        const auto t2 = std::chrono::steady_clock::now();
        VIOCHK(push_to_device(c, pl));
        HIPCHK(c->arena.end(c->stream));
        {
            const auto t3 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            c->timing[0] = us(t0, t1); c->timing[1] = us(t1, t2); c->timing[2] = us(t2, t3);
            if (timing) std::fprintf(stderr, "[vio host timing] activate(marg=%d): pull %.0f us, build_plan+upload %.0f us, push %.0f us\n", marg, us(t0, t1), us(t1, t2), us(t2, t3));
        }
        ++c->tables_gen;
        c->active = &pl;
        c->dirty_inputs = false;
        c->linearized = false;
        c->pairtab_valid = false;
        c->stepwise_updated = false;
        c->gn_lambda = -1.0;
    }
    return VIO_OK;
}

struct ProfScope {       // records an event pair around one kernel launch when that kernel is being profiled
    vio_ctx *c;
    bool on;
    ProfScope(vio_ctx *ctx, int id) : c(ctx), on(ctx->prof_which == id) {
        if (!on) return;
        if (c->prof_seen++ % c->prof_every != 0) { on = false; return; }
        if (c->prof_used + 2 > c->prof_events.size()) {
            for (int k = 0; k < 2; ++k) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } c->prof_events.push_back(e); }
        }
        (void)hipEventRecord(c->prof_events[c->prof_used], c->stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(c->prof_events[c->prof_used + 1], c->stream);
        c->prof_used += 2;
    }
};

// ---- RCCL, resolved at run time: the library has no link-time dependency on it -------------------------------
struct RcclId128 { char b[128]; };          // ncclUniqueId: passed by value to ncclCommInitRank
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(RcclId128 *id) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, RcclId128 id, int rank) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t s) = nullptr;
    int (*CommCount)(void *comm, int *count) = nullptr;
    int (*CommUserRank)(void *comm, int *rank) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

RcclApi *rccl_api(std::string &err) {
    static RcclApi api;
    static bool tried = false;
    if (tried) { if (!api.lib) err = "librccl.so could not be loaded"; return api.lib ? &api : nullptr; }
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) { err = std::string("dlopen librccl.so: ") + dlerror(); return nullptr; }
    *(void **)&api.GetUniqueId = dlsym(api.lib, "ncclGetUniqueId");
    *(void **)&api.CommInitRank = dlsym(api.lib, "ncclCommInitRank");
    *(void **)&api.CommDestroy = dlsym(api.lib, "ncclCommDestroy");
    *(void **)&api.AllGather = dlsym(api.lib, "ncclAllGather");
    *(void **)&api.GetErrorString = dlsym(api.lib, "ncclGetErrorString");
    *(void **)&api.CommCount = dlsym(api.lib, "ncclCommCount");
    *(void **)&api.CommUserRank = dlsym(api.lib, "ncclCommUserRank");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
        err = "librccl.so lacks ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllGather";
        dlclose(api.lib); api.lib = nullptr;
        return nullptr;
    }
    return &api;
}

inline bool sharded(const vio_ctx *c) { return c->hook != nullptr || c->comm != nullptr; }
// k_reduce_c + k_pose_solve_c instead of k_reduce, k_assemble_c, k_pose_solve_c: unsharded, chain order, no natural-order copy wanted
// (VIO_FOUR_LAUNCHES=1: the four-launch path, for A/B)
inline bool three_launch(const vio_ctx *c, const DeviceTables &T) {
    static const bool off = std::getenv("VIO_FOUR_LAUNCHES") != nullptr;
    return !off && !sharded(c) && T.solve_order == VIO_ORDER_CHAIN && !T.natural_hs && !T.marg_mode;
}

// The exchange of a sharded window is an all-gather, not an all-reduce: every rank receives every rank's slab and the kernels
// that read a sum form it in rank order (d_vis / d_step_tot of vio_kernels.hip) — identical bits on every rank by construction,
// independent of the collective library's algorithm (an all-reduce leaves the order of the additions to RCCL).
// which == 0: vis[0 .. VIS_SEND) of every shard -> gath[rank][..]   (24 KB per rank; max |h_ll| rides in its last slot)
// which == 1: the two step scalars of the stepwise path -> step_gath[rank][0..1]
vio_status run_exchange(vio_ctx *c, int which) {
    if (c->comm) {
        std::string err;
        RcclApi *api = rccl_api(err);
        if (!api) return fail(c, VIO_ERR_HIP, err);
        const double *send = which == 0 ? (c->ext_vis ? c->ext_vis : c->d_vis.p) : (c->ext_step ? c->ext_step : c->d_step_tot.p);
        double *recv = which == 0 ? (c->ext_gath ? c->ext_gath : c->d_gath.p) : (c->ext_step_gath ? c->ext_step_gath : c->d_step_gath.p);
        const int rc = api->AllGather(send, recv, which == 0 ? (size_t)VIS_SEND : 2, /*ncclDouble*/ 8, c->comm, c->stream);
        if (rc != 0) return fail(c, VIO_ERR_HIP, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"));
        return VIO_OK;
    }
    if (!c->hook) return VIO_OK;
    // Since ABI version 3 the hook is an ALL-GATHER into the receive buffers (it was an in-place all-reduce before).  A hook written
    // against the old contract would return 0 and leave the receive buffers empty: refused until the caller has shown that it knows
    // where the receive side is (vio_gather_buffers / vio_bind_gather_buffers).
    if (!c->gather_known)
        return fail(c, VIO_ERR_BAD_ARG, "exchange hook: the exchange is an all-gather into the buffers of vio_gather_buffers (VIO_ABI_VERSION >= 3); "
                                        "call vio_gather_buffers or vio_bind_gather_buffers before the first exchange");
    if (c->hook(c->hook_user, which) != 0) return fail(c, VIO_ERR_HIP, "exchange hook failed");
    return VIO_OK;
}

// prepare (if needed) + linearize + reduce + [exchange] + assemble at the current state
// gn = true: the GN loop (vio_gn_iteration): `cur` comes from the host; when a step is waiting for its test, k_reduce also
// sums its gain-ratio partials and k_assemble runs the test on the chi2 this very linearisation computes
// gate != 0: one slot of vio_solve's device-driven loop (the kernels skip themselves when it is not their turn)
vio_status enqueue_linearize(vio_ctx *c, Plan &pl, bool gn = false, int gate = 0) {
    DeviceTables T = gn ? make_tables_raw(c, pl) : make_tables(c, pl);
    if (!gn) VIOCHK(c->flush_status);
    T.lm_gate = gate;
    const bool test_prev = gn && c->decide_pending;
    if (gn) T.cur_hint = c->cur_host;
    if (test_prev) T.gn_flags = 2;
    if (!c->pairtab_valid && pl.lm_dim == 1) { vio_launch_prepare(T, c->stream); c->pairtab_valid = true; }
    // The GN loop's split solve (VIO_GN_SPLIT=1, off by default): lambda is the caller's, so the speed-bias chain of
    // this iteration's system — IMU factors, prior and lambda, nothing of the landmarks — can be eliminated by one more workgroup of
    // k_linearize's grid (it forms the IMU items itself: their workgroups are not launched) while k_pose_solve_cs starts at the camera block.
    // Bit-identical, and measured as a wash at 20 000 landmarks: the solve loses 8.1 us (30.2 -> 22.1), the linearisation gains the 8.5 us
    // by which the chain workgroup (IMU items 6 us, assembly 4 us, elimination 9 us, store) outlasts the items' 13.7 us.  2: diagnostic form.
    static const int split_mode = std::getenv("VIO_GN_SPLIT") ? std::atoi(std::getenv("VIO_GN_SPLIT")) : 0;
    c->gn_split = 0;
    const bool split = split_mode == 1 && gn && three_launch(c, T) && pl.lm_dim == 1 && pl.lin_threads == lin_threads_host() && lin_threads_host() == 1024 &&
                       T.n_items >= 1;
    size_t lin_lds = (size_t)pl.max_lds_doubles * 8;
    int lin_blocks = T.n_items + T.n_imu_items;
    DeviceTables TL = T;
    if (split) {
        if (T.has_prior && !c->prior_simg_valid) { vio_launch_prior_simg(T, c->stream); c->prior_simg_valid = true; }
        TL.gn_flags |= 16;
        TL.n_imu_items = 0;
        TL.n_step_blocks = T.n_items;          // (the rows of b_prior' are shared out over the item workgroups alone)
        lin_blocks = 1 + T.n_items;
        lin_lds = std::max(lin_lds, (size_t)vio_chain_pre_lds_doubles() * 8);
        c->gn_split = 1;
    }
    { ProfScope ps(c, VIO_K_LINEARIZE); vio_launch_linearize(TL, lin_blocks, lin_lds, pl.lin_threads, pl.use_ext, c->stream); }
    // with a prior, the previous GN step left err_prior to this k_reduce (k_pose_solve wrote b_prior' only)
    const bool err_prev = test_prev && T.has_prior;
    // sharded + gated slot: the all-reduce below runs whether the slot is live or not (every rank enqueues the same
    // collectives), in place on vis; k_reduce therefore always runs and puts this rank's own sums back first (a skipped
    // re-linearisation leaves the slabs as they were), so the buffer never accumulates the sum of sums
    ReduceTables R{pl.d_list_off.p, pl.d_list.p, pl.d_slab.p, T.vis, test_prev ? T.step_part : nullptr, T.n_items, sharded(c) ? 0 : gate, T.lm,
                   err_prev ? T.Jtinv : nullptr, err_prev ? T.bprior + c->cur_host * 176 : nullptr,
                   err_prev ? T.errprior + c->cur_host * 160 : nullptr};
    // Three launches per iteration where nothing sits between the sums and the assembly: the GN loop of an unsharded window in the
    // chain order, nobody waiting for the natural-order matrix.  k_reduce_c writes the image; the step test moves to the head of the
    // k_pose_solve_c that follows (enqueue_trial).
    c->test_in_solve = false;
    // (round 6: also the linearisation that opens vio_solve / vio_linearize — nothing of the loops' flags is needed for the fused sums;
    //  VIO_FIRST_FOUR_LAUNCHES=1 keeps k_reduce + k_assemble_c there, for A/B)
    static const bool first_four = std::getenv("VIO_FIRST_FOUR_LAUNCHES") != nullptr;
    if (three_launch(c, T) && (gn || (gate == 0 && !first_four))) {
        { ProfScope ps(c, VIO_K_REDUCE); vio_launch_reduce_assemble(R, T, c->stream); }
        // (VIO_GN_SPLIT=2, diagnostic: the chain eliminated in a launch of its own, from the assembled image)
        if (gn && split_mode == 2) { vio_launch_chain_pre(T, c->stream); c->gn_split = 2; }
        if (test_prev) { c->test_in_solve = true; c->decide_pending = false; }
        HIPCHK(hipGetLastError());
        c->linearized = true;
        c->natural_hs_valid = false;
        return VIO_OK;
    }
    { ProfScope ps(c, VIO_K_REDUCE); vio_launch_reduce(R, c->stream); }
    VIOCHK(run_exchange(c, 0));
    if (test_prev) { T.gn_flags = 1; c->decide_pending = false; }
    { ProfScope ps(c, VIO_K_ASSEMBLE); vio_launch_assemble(T, c->stream); }
    HIPCHK(hipGetLastError());
    c->linearized = true;
    c->natural_hs_valid = T.natural_hs != 0;
    return VIO_OK;
}

// ComputeLambdaInitLM; with shards, max |h_ll| of every rank came with its slab of the last exchange (k_init_lm takes the max)
vio_status enqueue_init_lm(vio_ctx *c, const DeviceTables &T, int max_iter) {
    vio_launch_init_lm(T, max_iter, c->stream);
    HIPCHK(hipGetLastError());
    return VIO_OK;
}

vio_status enqueue_trial(vio_ctx *c, Plan &pl, int mode, bool gn = false, int gate = 0) {
    DeviceTables T = gn ? make_tables_raw(c, pl) : make_tables(c, pl);
    if (!gn) VIOCHK(c->flush_status);
    T.lm_gate = gate;
    T.gn_flags = 4;         // bit 2: k_pose_solve leaves the prior update to the kernels that follow (all paths now)
    if (gn) T.cur_hint = c->cur_host;      // GN: the update rides with the next k_linearize / k_reduce (or with flush_decide)
    if (gn && c->test_in_solve) { T.gn_flags |= 1; c->test_in_solve = false; }      // three-launch path: the previous step's test at this kernel's head
    if (gn && c->gn_split) { T.gn_flags |= 16; c->gn_split = 0; }                   // ... and the chain of this system is eliminated already
    static const bool no_early = std::getenv("VIO_NO_EARLY_START") != nullptr;      // diagnostic / tests: k_pose_solve_c's prologue with its barriers (bit 5)
    if (no_early) T.gn_flags |= 32;
    { ProfScope ps(c, VIO_K_POSE_SOLVE); vio_launch_pose_solve(T, POSE_SOLVE_LDS, c->stream); }
    if (gn) {
        // the step is accepted whatever chi2 turns out to be: the landmark back-substitution, the chi2 of the new state and
        // the step test all belong to the next iteration's k_linearize / k_assemble (flush_decide does them the
        // classic way when anybody else asks first)
        c->decide_pending = true;
        c->cur_host ^= 1;
        HIPCHK(hipGetLastError());
        c->ahead = 7u;
        return VIO_OK;
    }
    // the prior update of an LM trial (problem.cc:466-475) is spread over the kernels that follow, like a flushed GN step's:
    // b_prior' by k_backsub's workgroups, err_prior' by k_errprior (one CU alone needs ~10 us for the two matrices)
    T.gn_flags = 8;
    { ProfScope ps(c, VIO_K_BACKSUB); vio_launch_backsub(T, 0, c->stream); }
    T.gn_flags = 0;
    if (T.has_prior) vio_launch_errprior(T, c->stream);
    if (sharded(c)) {
        vio_launch_step_sum(T, 0, c->stream);
        VIOCHK(run_exchange(c, 1));
        ProfScope ps(c, VIO_K_LM_DECIDE);
        vio_launch_lm_decide(T, mode, 0, c->stream);
    } else {
        ProfScope ps(c, VIO_K_LM_DECIDE);
        vio_launch_lm_decide(T, mode, 1, c->stream);
    }
    HIPCHK(hipGetLastError());
    c->ahead = 7u;
    return VIO_OK;
}

// One slot of vio_solve's loop.  Problem::Solve tests a step by evaluating chi2 at the trial state and, when the step is good,
// linearises there (problem.cc:210-233); here the linearisation comes first — chi2 is a by-product of it — so an iteration is the
// GN loop's four launches: k_linearize at the trial state (the step's landmark back-substitution and b_prior' rows in its head),
// k_reduce (+ err_prior'), [exchange], k_assemble (chi2 and the gain ratio's denominator into LmState), k_pose_solve (the
// verdict; on accept the next step from the system just assembled; on reject nothing: the next slot linearises at the kept
// state again, which a rejected trial costs here instead of a second evaluation).  `first`: the step from the initial linearisation.
vio_status enqueue_lm_slot(vio_ctx *c, Plan &pl, bool first) {
    DeviceTables T = make_tables_raw(c, pl);
    bool test_in_solve = false;
    T.cur_hint = -2;
    T.lm_gate = 2;
    if (!first) {
        T.gn_flags = 2;
        { ProfScope ps(c, VIO_K_LINEARIZE); vio_launch_linearize(T, T.n_items + T.n_imu_items, (size_t)pl.max_lds_doubles * 8, pl.lin_threads, pl.use_ext, c->stream); }
        // (sharded: k_reduce always runs, see enqueue_linearize)
        ReduceTables R{pl.d_list_off.p, pl.d_list.p, pl.d_slab.p, T.vis, T.step_part, T.n_items, sharded(c) ? 0 : 2, T.lm,
                       T.has_prior ? T.Jtinv : nullptr, T.has_prior ? T.bprior : nullptr, T.has_prior ? T.errprior : nullptr, 1};
        if (three_launch(c, T)) {
            { ProfScope ps(c, VIO_K_REDUCE); vio_launch_reduce_assemble(R, T, c->stream); }
            test_in_solve = true;
        } else {
            { ProfScope ps(c, VIO_K_REDUCE); vio_launch_reduce(R, c->stream); }
            VIOCHK(run_exchange(c, 0));
            T.gn_flags = 1;
            { ProfScope ps(c, VIO_K_ASSEMBLE); vio_launch_assemble(T, c->stream); }
        }
    }
    T.gn_flags = 4 | (test_in_solve ? 1 : 0);
    { ProfScope ps(c, VIO_K_POSE_SOLVE); vio_launch_pose_solve(T, POSE_SOLVE_LDS, c->stream); }
    HIPCHK(hipGetLastError());
    c->ahead = 7u;
    return VIO_OK;
}

// MargOldFrame straight after a solve (the frame loop): its plan (the landmarks hosted in frame 0) and their positions in the solve
// plan depend on the graph only, not on the solve's result.  vio_solve calls this between enqueueing its slots and waiting for
// them, so the host builds the next call's tables while the device is busy; vio_marginalize calls it again (a no-op then).
vio_status prepare_marg_plan(vio_ctx *c) {
    Plan &mp = c->marg_plan, &sp = c->solve_plan;
    if (c->lm_dim != 1 || c->active != &sp || !sp.valid || c->topo_dirty) return VIO_OK;
    if (mp.valid && c->marg_map_valid) return VIO_OK;
    HIPCHK(c->arena.begin());
    if (!mp.valid) VIOCHK(build_plan(c, mp, 1));
    std::vector<int32_t> pos(c->h_invd.size(), -1);
    for (int64_t q = 0; q < sp.Ns; ++q) pos[sp.sorted_to_orig[q]] = (int32_t)q;
    int32_t *map = (int32_t *)c->arena.alloc((size_t)std::max<int64_t>(mp.Ns, 1) * 4);
    if (!map) return fail(c, VIO_ERR_HIP, "hipHostMalloc (staging)");
    for (int64_t q = 0; q < mp.Ns; ++q) map[q] = pos[mp.sorted_to_orig[q]];
    HIPCHK(c->d_gather_map.resize((size_t)std::max<int64_t>(mp.Ns, 1)));
    if (mp.Ns) HIPCHK(hipMemcpyAsync(c->d_gather_map.p, map, (size_t)mp.Ns * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c->arena.end(c->stream));
    c->marg_map_valid = true;
    return VIO_OK;
}

}  // namespace

// =========================================================================================================
extern "C" {

void vio_default_config(vio_config *cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->ext_fixed = 1;                        // estimate_extrinsic: 0 (VM/config/vio_simulation.yaml:25)
    cfg->loss_type = VIO_LOSS_CAUCHY;          // CauchyLoss(1.0), estimator.cpp:905
    cfg->loss_delta = 1.0;
    cfg->reproj_sqrt_info = 460.0 / 1.5;       // estimator.cpp:42
    cfg->gravity[2] = 9.81;                    // g_norm (vio_simulation.yaml:79)
    cfg->shard_rank = 0;
    cfg->shard_count = 1;
}

vio_status vio_create(const vio_config *cfg, vio_ctx **out) {
    if (!cfg || !out) return VIO_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VIO_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return VIO_ERR_BAD_ARG;
    if (hipSetDevice(cfg->device) != hipSuccess) return VIO_ERR_HIP;
    vio_ctx *c = new vio_ctx();
    vio_plan::shared_acquire(); c->shared_ref = true;       // (the process's helper threads live as long as any context does)
    c->cfg = *cfg;
    if (c->cfg.shard_count < 1) c->cfg.shard_count = 1;
    if (cfg->stream) c->stream = (hipStream_t)cfg->stream;
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { vio_plan::shared_release(); delete c; return VIO_ERR_HIP; }
        c->own_stream = true;
    }
    std::memset(c->h_state, 0, sizeof(c->h_state));
    c->h_state[STATE_EXT + 6] = 1.0;
    for (int i = 0; i < NF; ++i) c->h_state[STATE_POSE + 7 * i + 6] = 1.0;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) c->imu_valid[k] = false;
    c->h_pre.assign(VIO_WINDOW_SIZE * PRE_STRIDE, 0.0);
    c->h_bprior.assign(PD, 0.0); c->h_errprior.assign(PRD, 0.0);
    if (!c->h_Hprior.resize_uninitialized((size_t)PD * PD) || !c->h_Jtinv.resize_uninitialized((size_t)PRD * PRD)) { vio_plan::shared_release(); c->h_Hprior.release(); c->h_Jtinv.release(); if (c->own_stream) hipStreamDestroy(c->stream); delete c; return VIO_ERR_HIP; }
    std::memset(c->h_Hprior.p, 0, (size_t)PD * PD * 8); std::memset(c->h_Jtinv.p, 0, (size_t)PRD * PRD * 8);
    if (const char *e = std::getenv("VIO_G_MAX")) { int v = std::atoi(e); if (v >= 1 && v <= 128) c->g_max = v; }
    if (const char *e = std::getenv("VIO_G_MIN")) { int v = std::atoi(e); if (v >= 1 && v <= 128) c->g_min = v; }
    if (c->g_max > 0) c->g_min = c->g_max;     // a forced size is exactly that size (LDS permitting)
    if (const char *e = std::getenv("VIO_SOLVE_ORDER")) c->solve_order = (e[0] == 'e' || e[0] == 'E' || e[0] == '0') ? VIO_ORDER_EIGEN : VIO_ORDER_CHAIN;
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->cfg.device) == hipSuccess && v > 0) c->n_cus = v; }
    if (vio_set_kernel_attributes() != 0) { c->err = "hipFuncSetAttribute failed"; }
    vio_status s = alloc_fixed(c);
    if (s != VIO_OK) { vio_destroy(c); return s; }
    *out = c;
    return VIO_OK;
}

vio_status vio_set_config(vio_ctx *c, const vio_config *cfg) {
    if (!c || !cfg) return VIO_ERR_BAD_ARG;
    if (cfg->device != c->cfg.device || (cfg->stream && (hipStream_t)cfg->stream != c->stream) || cfg->shard_rank != c->cfg.shard_rank ||
        std::max(cfg->shard_count, 1) != c->cfg.shard_count)
        return fail(c, VIO_ERR_BAD_ARG, "vio_set_config: device, stream and shard fields belong to the context's creation");
    enter_device(c);
    VIOCHK(pull_from_device(c));
    void *keep_stream = c->cfg.stream;
    const bool replan = cfg->ext_fixed != c->cfg.ext_fixed || cfg->item_policy != c->cfg.item_policy;     // the patterns carry an extrinsic block or not; the items' size
    c->cfg = *cfg;
    c->cfg.stream = keep_stream;
    if (c->cfg.shard_count < 1) c->cfg.shard_count = 1;
    if (replan) c->topo_dirty = true;
    c->dirty_inputs = true;
    c->prior_simg_valid = false;        // (ext_fixed decides which prior entries are masked)
    return VIO_OK;
}

static void marg_join(vio_ctx *c);
void vio_destroy(vio_ctx *c) {
    if (!c) return;
    marg_join(c);
    if (c->shared_ref) { vio_plan::shared_release(); c->shared_ref = false; }
    enter_device(c);
    // A stream the caller supplied (vio_config.stream, e.g. the leader's of a batch) may be gone already when this context
    // goes: it is not touched here.  hipFree waits for the device itself, so nothing in flight loses its buffers.
    if (c->own_stream) hipStreamSynchronize(c->stream);
    else hipDeviceSynchronize();
    vio_comm_destroy(c);
    c->solve_plan.release(); c->marg_plan.release();
    c->d_state.release(); c->d_pairtab.release(); c->d_vis.release(); c->d_pre.release(); c->d_imu_out.release();
    c->d_Hprior.release(); c->d_bprior.release(); c->d_errprior.release(); c->d_Jtinv.release(); c->d_Hs.release();
    c->d_bs.release(); c->d_bfull.release(); c->d_diagfull.release(); c->d_dx.release(); c->d_step_tot.release(); c->d_sp_part.release();
    c->d_imu_chi.release(); c->d_imu_valid.release(); c->d_lm.release(); c->d_perm.release(); c->d_Pg.release(); c->d_cfi.release(); c->d_imu_map.release(); c->d_prior_simg.release(); c->d_prior_flags.release(); c->d_prior_list.release(); c->d_prior_cval.release();
    c->d_batch_tabs.release(); c->d_frame_block.release(); c->d_rank.release(); c->d_gather_map.release(); c->d_gath.release(); c->d_step_gath.release();
    c->arena.release(c->own_stream);
    c->h_pts_j.release(); c->d_raw_pts_j.release(); c->h_Hprior.release(); c->h_Jtinv.release();
    if (c->pull_stage) hipHostFree(c->pull_stage);
    if (c->marg_stage) hipHostFree(c->marg_stage);
    if (c->h_lm_pin) hipHostFree(c->h_lm_pin);
    if (c->lm_event) hipEventDestroy(c->lm_event);
    for (hipEvent_t e : c->lin_ev) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->prof_events) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->stream);
    (void)hipGetLastError();         // (an event of a borrowed stream that is gone may have complained: not the next caller's business)
    delete c;
}

const char *vio_last_error(const vio_ctx *c) { return c ? c->err.c_str() : "null context"; }

vio_status vio_set_window(vio_ctx *c, const double *poses, const double *sb, const double *ext) {
    if (!c || !poses || !sb || !ext) return VIO_ERR_BAD_ARG;
    enter_device(c);
    c->ahead &= ~1u;                 // the states are replaced whole: nothing of the device's to keep
    std::memcpy(c->h_state + STATE_EXT, ext, 7 * 8);
    std::memcpy(c->h_state + STATE_POSE, poses, 77 * 8);
    std::memcpy(c->h_state + STATE_SB, sb, 99 * 8);
    c->dirty_inputs = true;
    return VIO_OK;
}

static vio_status set_landmarks_dim(vio_ctx *c, int64_t n, const double *val, int dim) {
    if (!c || n < 0 || (n > 0 && !val)) return VIO_ERR_BAD_ARG;
    enter_device(c);
    const bool resized = (int64_t)c->h_invd.size() != n * dim || c->lm_dim != dim;
    // The same values as the mirror holds, the mirror current: nothing to do.  (The reference's frame sets the window twice,
    // for Solve and for Marginalize, with the landmarks the solve gave it: estimator.cpp:1083-1092.)
    if (!resized && !(c->ahead & 2u) && (n == 0 || std::memcmp(c->h_invd.data(), val, (size_t)n * dim * 8) == 0)) return VIO_OK;
    c->ahead &= ~2u;                 // the landmarks are replaced whole
    c->h_invd.assign(val, val + n * dim);
    c->lm_dim = dim;
    if (resized) {      // the observation list refers to landmark indices (of this kind): it must be set again
        c->h_olm.clear(); c->h_ohost.clear(); c->h_otarget.clear(); c->h_pts_i.clear(); c->h_pts_j.clear();
        c->obs_lm_major = c->obs_consistent = c->raw_pts_valid = false;
        if (c->obs_mapped) c->obs_map_stale = true;       // the caller's pointers are gone: vio_commit_observations says so
        c->topo_dirty = true;
    }
    c->dirty_inputs = true;
    return VIO_OK;
}

vio_status vio_set_landmarks(vio_ctx *c, int64_t n, const double *invd) { return set_landmarks_dim(c, n, invd, 1); }
vio_status vio_set_landmarks_xyz(vio_ctx *c, int64_t n, const double *xyz) { return set_landmarks_dim(c, n, xyz, 3); }

vio_status vio_set_observations_xyz(vio_ctx *c, int64_t m, const int32_t *lm, const int32_t *frame, const double *pts) {
    if (!c || m < 0 || (m > 0 && (!lm || !frame || !pts))) return VIO_ERR_BAD_ARG;
    if (c->lm_dim != 3) return fail(c, VIO_ERR_BAD_ARG, "vio_set_observations_xyz needs vio_set_landmarks_xyz first");
    const int64_t N = (int64_t)c->h_invd.size() / 3;
    {   // one pass: indices in range?  landmark-major with a landmark's frames ascending (then observation k of a landmark is the k-th
        // frame of its pattern and the list is its own CSR: plan_xyz's fast path, the observations put into item order on the device)?
        const vio_plan::ScanResult r = vio_plan::scan_observations_xyz(N, m, lm, frame);
        if (r.bad) return fail(c, VIO_ERR_BAD_ARG, "observation " + std::to_string(r.bad_index) + " out of range");
        c->obs_lm_major = r.lm_major;
        c->obs_consistent = false;
    }
    // (what the device holds newer than the host mirrors stays there until activate() needs it: the old plan is alive till then)
    if ((int64_t)c->h_olm.size() == m && (m == 0 || (std::memcmp(c->h_olm.data(), lm, (size_t)m * 4) == 0 && std::memcmp(c->h_otarget.data(), frame, (size_t)m * 4) == 0 &&
                                                     std::memcmp(c->h_pts_j.data(), pts, (size_t)m * 16) == 0)))
        return VIO_OK;               // the graph the context already holds: its plans stay
    enter_device(c);
    if (c->arena.pending) { HIPCHK(hipEventSynchronize(c->arena.ev)); c->arena.pending = false; }
    c->h_olm.assign(lm, lm + m); c->h_otarget.assign(frame, frame + m); c->h_ohost.assign((size_t)m, 0);
    if (!c->h_pts_j.assign(pts, pts + 2 * m)) return fail(c, VIO_ERR_HIP, "hipHostMalloc (observations)");
    c->h_pts_i.assign(2 * (size_t)m, 0.0);
    c->raw_pts_valid = false;
    c->topo_dirty = true;
    c->dirty_inputs = true;
    return VIO_OK;
}

// One pass over an observation list (vio_plan::scan_observations): range, landmark-major?, do a landmark's edges share host frame and host
// observation?; the landmark's host observation is noted by landmark on the way (h_pts_i_lm), for the planner, which repeats the
// consistency check per landmark only for the lists this pass does not vouch for (and names the offender).
// *same_pi (optional): every landmark's host observation is the one h_pts_i_lm held before the call (meaningful when the list the context
// holds was vouched for: then that is all there is to compare — the per-edge copies are not kept for such lists).
// The pass and, for vio_set_observations, the copy into the context's mirrors run on a few threads when the list is long (80 000 edges: 2.2 MB
// to read and 2.2 MB to copy, DRAM-bound on one core: 0.20 ms of a 1.4 ms frame; three parked helpers: see vio_plan.h).  Each piece of the
// list scans its edges, compares them with what the mirrors hold (when the sizes allow) and copies where they differ.
struct ObsPass {
    int64_t N, m;
    const int32_t *lm, *host, *target;
    const double *pi, *pj;
    double *pl;
    int32_t *mlm, *mhost, *mtarget;      // the mirrors (null: scan only)
    double *mpj;
    bool comparable;
    int pieces;
    vio_plan::ScanFlags flags[8];
    bool same[8];
};
static void obs_pass_piece(void *arg, int i) {
    ObsPass &o = *(ObsPass *)arg;
    const int64_t e0 = o.m * i / o.pieces, e1 = o.m * (i + 1) / o.pieces;
    o.flags[i] = vio_plan::scan_range(o.N, e0, e1, o.lm, o.host, o.target, o.pi, o.pl);
    o.same[i] = false;
    if (!o.mlm || e1 == e0) { o.same[i] = o.comparable; return; }
    const size_t k = (size_t)(e1 - e0);
    if (o.comparable)
        o.same[i] = std::memcmp(o.mlm + e0, o.lm + e0, k * 4) == 0 && std::memcmp(o.mhost + e0, o.host + e0, k * 4) == 0 &&
                    std::memcmp(o.mtarget + e0, o.target + e0, k * 4) == 0 && std::memcmp(o.mpj + 2 * e0, o.pj + 2 * e0, k * 16) == 0;
    if (!o.same[i]) {
        std::memcpy(o.mlm + e0, o.lm + e0, k * 4); std::memcpy(o.mhost + e0, o.host + e0, k * 4); std::memcpy(o.mtarget + e0, o.target + e0, k * 4);
        std::memcpy(o.mpj + 2 * e0, o.pj + 2 * e0, k * 16);
    }
}
// runs the pass; *r = the scan's verdict, *same_graph = every piece found its edges in the mirrors already
static void run_obs_pass(vio_ctx *c, ObsPass &o, vio_plan::ScanResult *r, bool *same_graph) {
    bool resized = false;
    if (c->h_pts_i_lm.size() != 2 * (size_t)o.N) { c->h_pts_i_lm.assign(2 * (size_t)o.N, 0.0); resized = true; }
    o.pl = c->h_pts_i_lm.data();
    o.pieces = 1;
    vio_plan::HostPool *hp = nullptr;
    if (o.m >= 16384) {
        // pieces only for landmark-major lists (then no two of them note the same landmark's host observation): one look at the landmark
        // indices alone, 0.3 MB, tells
        unsigned unsorted = 0;
        for (int64_t e = 1; e < o.m; ++e) unsorted |= (unsigned)(o.lm[e] < o.lm[e - 1]);
        if (!unsorted) {
            hp = vio_plan::shared_pool();
            o.pieces = std::min(vio_plan::pool_width(hp), 4);          // (four pieces: the pass is bound by one core's memory bandwidth, not by arithmetic)
        }
    }
    vio_plan::pool_run(o.pieces > 1 ? hp : nullptr, o.pieces, obs_pass_piece, &o);
    vio_plan::ScanFlags f;
    bool same = o.comparable;
    for (int i = 0; i < o.pieces; ++i) { f.bad |= o.flags[i].bad; f.unsorted |= o.flags[i].unsorted; f.incons |= o.flags[i].incons; f.changed |= o.flags[i].changed; same = same && o.same[i]; }
    f.changed |= resized ? 1u : 0u;
    *r = vio_plan::scan_finish(f, o.N, o.m, o.lm, o.host, o.target);
    *same_graph = same;
}
static void drop_observations(vio_ctx *c) {
    c->h_olm.clear(); c->h_ohost.clear(); c->h_otarget.clear(); c->h_pts_i.clear(); c->h_pts_j.clear();
    c->obs_lm_major = c->obs_consistent = false;
}

vio_status vio_set_observations(vio_ctx *c, int64_t m, const int32_t *lm, const int32_t *host, const int32_t *target,
                                const double *pi, const double *pj) {
    if (!c || m < 0 || (m > 0 && (!lm || !host || !target || !pi || !pj))) return VIO_ERR_BAD_ARG;
    if (c->lm_dim == 3) return fail(c, VIO_ERR_BAD_ARG, "the context holds XYZ landmarks: use vio_set_observations_xyz");
    if (c->obs_mapped) {            // vio_set_observations ends a mapping (include/vio_backend.h): what was written in place is dropped
        c->obs_mapped = c->obs_map_stale = false;
        drop_observations(c);
    }
    const bool was_vouched = c->obs_consistent && c->h_pts_i.empty();       // the list held so far: host observations by landmark only
    enter_device(c);
    // The graph the context already holds (MargOldFrame after problemSolve, repeated solves) is recognised by comparison alone, before anything
    // waits for the device or writes a mirror: the pinned h_pts_j may still be the source of the previous activation's upload, and only a
    // list that differs has to wait for that.  The target observations first: a new frame's differ in their first bytes.
    if (m > 0 && (int64_t)c->h_olm.size() == m && (int64_t)c->h_ohost.size() == m && (int64_t)c->h_otarget.size() == m && c->h_pts_j.size() == 2 * (size_t)m &&
        std::memcmp(c->h_pts_j.p, pj, (size_t)m * 16) == 0 && std::memcmp(c->h_olm.data(), lm, (size_t)m * 4) == 0 &&
        std::memcmp(c->h_ohost.data(), host, (size_t)m * 4) == 0 && std::memcmp(c->h_otarget.data(), target, (size_t)m * 4) == 0) {
        bool same_hosts;
        if (!was_vouched) same_hosts = c->h_pts_i.size() == 2 * (size_t)m && std::memcmp(c->h_pts_i.data(), pi, (size_t)m * 16) == 0;
        else {
            same_hosts = c->h_pts_i_lm.size() == 2 * c->h_invd.size();
            const double *pl = c->h_pts_i_lm.data();
            for (int64_t e = 0; e < m && same_hosts; ++e) same_hosts = pl[2 * (size_t)lm[e]] == pi[2 * e] && pl[2 * (size_t)lm[e] + 1] == pi[2 * e + 1];     // (lm[e] is in range: it equals the mirror's, which was checked)
        }
        if (same_hosts) return VIO_OK;
    }
    if (c->arena.pending) { HIPCHK(hipEventSynchronize(c->arena.ev)); c->arena.pending = false; }      // an upload out of h_pts_j still in flight (long done)
    // the mirrors take the list piece by piece while it is scanned; pieces that hold these very edges already are left alone
    ObsPass o;
    o.N = (int64_t)c->h_invd.size(); o.m = m; o.lm = lm; o.host = host; o.target = target; o.pi = pi; o.pj = pj;
    o.comparable = (int64_t)c->h_olm.size() == m && (int64_t)c->h_ohost.size() == m && (int64_t)c->h_otarget.size() == m && c->h_pts_j.size() == 2 * (size_t)m;
    if (!o.comparable) {
        c->h_olm.resize((size_t)m); c->h_ohost.resize((size_t)m); c->h_otarget.resize((size_t)m);
        if (!c->h_pts_j.resize_uninitialized(2 * (size_t)m)) { drop_observations(c); return fail(c, VIO_ERR_HIP, "hipHostMalloc (observations)"); }
    }
    o.mlm = c->h_olm.data(); o.mhost = c->h_ohost.data(); o.mtarget = c->h_otarget.data(); o.mpj = c->h_pts_j.p;
    vio_plan::ScanResult r;
    bool same_graph = false;
    run_obs_pass(c, o, &r, &same_graph);
    if (r.bad) {
        // the pass has written into h_pts_i_lm and the mirrors, which the list held so far relied on: a refused list leaves the context without one
        drop_observations(c);
        c->raw_pts_valid = false; c->topo_dirty = true; c->dirty_inputs = true;
        return fail(c, VIO_ERR_BAD_ARG, "observation " + std::to_string(r.bad_index) + " out of range");
    }
    c->obs_lm_major = r.lm_major;
    c->obs_consistent = r.consistent;
    const bool vouched = r.consistent;
    bool same_pi;
    if (!was_vouched) same_pi = !vouched && c->h_pts_i.size() == 2 * (size_t)m && (m == 0 || std::memcmp(c->h_pts_i.data(), pi, (size_t)m * 16) == 0);
    else same_pi = !r.changed && vouched;
    if (same_pi && same_graph) return VIO_OK;               // the graph the context already holds: its plans stay
    // the per-edge copies of the host observation are kept only for lists whose landmarks' edges this pass could not vouch for (1.3 MB
    // of the 3.5 MB a frame's list is: what a vouched list needs of them is in h_pts_i_lm)
    if (vouched) c->h_pts_i.clear(); else c->h_pts_i.assign(pi, pi + 2 * m);
    c->raw_pts_valid = false;
    c->topo_dirty = true;
    c->dirty_inputs = true;
    return VIO_OK;
}

// The list written in place (include/vio_backend.h): the context's own mirrors go out to the caller, commit scans and adopts them.
vio_status vio_map_observations(vio_ctx *c, int64_t m, int32_t **lm, int32_t **host, int32_t **target, double **pi, double **pj) {
    if (!c || m < 0 || !lm || !host || !target || !pi || !pj) return VIO_ERR_BAD_ARG;
    if (c->lm_dim == 3) return fail(c, VIO_ERR_BAD_ARG, "the context holds XYZ landmarks: use vio_set_observations_xyz");
    enter_device(c);
    if (c->arena.pending) { HIPCHK(hipEventSynchronize(c->arena.ev)); c->arena.pending = false; }
    c->h_olm.resize((size_t)m); c->h_ohost.resize((size_t)m); c->h_otarget.resize((size_t)m); c->h_pts_i.resize(2 * (size_t)m);
    if (!c->h_pts_j.resize_uninitialized(2 * (size_t)m)) return fail(c, VIO_ERR_HIP, "hipHostMalloc (observations)");
    *lm = c->h_olm.data(); *host = c->h_ohost.data(); *target = c->h_otarget.data(); *pi = c->h_pts_i.data(); *pj = c->h_pts_j.p;
    c->obs_mapped = true;           // the old list is gone, the new one is not there yet
    c->obs_map_stale = false;
    c->obs_lm_major = c->obs_consistent = false;
    c->raw_pts_valid = false;
    c->topo_dirty = true;
    c->dirty_inputs = true;
    return VIO_OK;
}

vio_status vio_commit_observations(vio_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (!c->obs_mapped) return fail(c, VIO_ERR_BAD_ARG, "vio_commit_observations without vio_map_observations");
    c->obs_mapped = false;
    if (c->obs_map_stale) {
        c->obs_map_stale = false;
        c->h_olm.clear(); c->h_ohost.clear(); c->h_otarget.clear(); c->h_pts_i.clear(); c->h_pts_j.clear();
        return fail(c, VIO_ERR_BAD_ARG, "vio_commit_observations: the mapping was invalidated by vio_set_landmarks (another landmark count): map again");
    }
    const int64_t m = (int64_t)c->h_olm.size();
    vio_status st = VIO_OK;
    {
        ObsPass o;
        o.N = (int64_t)c->h_invd.size(); o.m = m; o.lm = c->h_olm.data(); o.host = c->h_ohost.data(); o.target = c->h_otarget.data(); o.pi = c->h_pts_i.data(); o.pj = nullptr;
        o.mlm = o.mhost = o.mtarget = nullptr; o.mpj = nullptr; o.comparable = false;
        vio_plan::ScanResult r;
        bool same_graph = false;
        run_obs_pass(c, o, &r, &same_graph);
        if (r.bad) st = fail(c, VIO_ERR_BAD_ARG, "observation " + std::to_string(r.bad_index) + " out of range");
        else { c->obs_lm_major = r.lm_major; c->obs_consistent = r.consistent; }
    }
    if (st != VIO_OK) drop_observations(c);
    return st;
}

static vio_status set_imu_one(vio_ctx *c, int32_t k, const vio_preint *pre) {
    double blk[PRE_STRIDE];
    double *o = blk;
    std::memcpy(blk, c->h_pre.data() + (size_t)k * PRE_STRIDE, sizeof(blk));
    if (pre) {
        o[PRE_SUMDT] = pre->sum_dt;
        for (int i = 0; i < 3; ++i) { o[PRE_DP + i] = pre->delta_p[i]; o[PRE_DV + i] = pre->delta_v[i]; o[PRE_BA + i] = pre->linearized_ba[i]; o[PRE_BG + i] = pre->linearized_bg[i]; }
        for (int i = 0; i < 4; ++i) o[PRE_DQ + i] = pre->delta_q[i];
        std::memcpy(o + PRE_JAC, pre->jacobian, 225 * 8);
        vio_host::inverse15(pre->covariance, o + PRE_INFO);     // SetInformation(covariance.inverse()), edge_imu.cc:35
    }
    if (c->imu_valid[k] == (pre != nullptr) && std::memcmp(blk, c->h_pre.data() + (size_t)k * PRE_STRIDE, sizeof(blk)) == 0) return VIO_OK;     // unchanged
    c->imu_valid[k] = pre != nullptr;
    std::memcpy(c->h_pre.data() + (size_t)k * PRE_STRIDE, blk, sizeof(blk));
    c->imu_dirty = true;
    c->dirty_inputs = true;
    return VIO_OK;
}

vio_status vio_set_imu(vio_ctx *c, int32_t k, const vio_preint *pre) {
    if (!c || k < 0 || k >= VIO_WINDOW_SIZE) return VIO_ERR_BAD_ARG;
    enter_device(c);
    return set_imu_one(c, k, pre);
}

vio_status vio_set_imu_all(vio_ctx *c, const vio_preint *const *pre) {
    if (!c || !pre) return VIO_ERR_BAD_ARG;
    enter_device(c);
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) VIOCHK(set_imu_one(c, k, pre[k]));
    return VIO_OK;
}

vio_status vio_set_prior(vio_ctx *c, int32_t dim, const double *H, const double *b, const double *err, const double *jt) {
    if (!c || (dim != 0 && dim != PRD)) return VIO_ERR_BAD_ARG;
    if (dim && (!H || !b || !err || !jt)) return VIO_ERR_BAD_ARG;
    enter_device(c);
    // The two matrices are 430 KB of upload: left alone when they are the ones the context holds (the reference hands the same
    // H_prior / Jt_prior_inv to Solve and then to Marginalize, with b_prior / err_prior as the solve updated them)
    bool same_mats = c->has_prior == (dim ? 1 : 0);
    if (same_mats && dim) {
        same_mats = std::memcmp(c->h_Jtinv.data(), jt, (size_t)PRD * PRD * 8) == 0;
        for (int i = 0; same_mats && i < PRD; ++i) same_mats = std::memcmp(c->h_Hprior.p + (size_t)i * PD, &H[(size_t)i * PRD], PRD * 8) == 0;
    }
    const bool same_vecs = same_mats && !(c->ahead & 4u) &&
                           (dim == 0 || (std::memcmp(c->h_bprior.data(), b, PRD * 8) == 0 && std::memcmp(c->h_errprior.data(), err, PRD * 8) == 0));
    if (same_vecs) return VIO_OK;
    c->ahead &= ~4u;                 // b_prior and err_prior are replaced whole
    std::fill(c->h_bprior.begin(), c->h_bprior.end(), 0.0); std::fill(c->h_errprior.begin(), c->h_errprior.end(), 0.0);
    if (!same_mats) {
        // (the pinned mirrors may still be the source of the previous activation's upload: long done in practice)
        if (c->arena.pending) { HIPCHK(hipEventSynchronize(c->arena.ev)); c->arena.pending = false; }
        // a prior that goes away leaves zeros; one that comes overwrites its 156 x 156 corner, and the 15 rows / columns of
        // ExtendHessiansPriorSize(15) (problem.cc:82-91) are zero and stay zero: nobody writes them (430 KB of memset per frame before)
        if (!dim) { std::memset(c->h_Hprior.p, 0, (size_t)PD * PD * 8); std::memset(c->h_Jtinv.p, 0, (size_t)PRD * PRD * 8); }
        c->has_prior = dim ? 1 : 0;
        c->prior_dirty = true;
    }
    if (dim) {
        for (int i = 0; i < PRD; ++i) c->h_bprior[i] = b[i];
        std::memcpy(c->h_errprior.data(), err, PRD * 8);
        if (!same_mats) {
            for (int i = 0; i < PRD; ++i) std::memcpy(c->h_Hprior.p + (size_t)i * PD, &H[(size_t)i * PRD], PRD * 8);
            std::memcpy(c->h_Jtinv.p, jt, (size_t)PRD * PRD * 8);
        }
    }
    if (!same_mats) {
        // speed-bias block f occupies rows 12 + 15 f .. 20 + 15 f
        bool ok = true;
        for (int f = 0; ok && f < NF; ++f)
            for (int g2 = 0; ok && g2 < NF; ++g2) {
                if (g2 >= f - 1 && g2 <= f + 1) continue;
                for (int a = 0; ok && a < 9; ++a)
                    for (int b2 = 0; b2 < 9; ++b2)
                        if (c->h_Hprior[(size_t)(12 + 15 * f + a) * PD + 12 + 15 * g2 + b2] != 0.0) { ok = false; break; }
            }
        c->prior_chain_ok = ok;
    }
    c->dirty_inputs = true;
    return VIO_OK;
}

vio_status vio_set_solve_order(vio_ctx *c, int32_t order) {
    if (!c || (order != VIO_ORDER_EIGEN && order != VIO_ORDER_CHAIN)) return VIO_ERR_BAD_ARG;
    if (order == c->solve_order) return VIO_OK;
    enter_device(c);
    VIOCHK(flush_decide(c));
    c->solve_order = order;
    c->linearized = false;          // the assembled system on the device is in the other order's layout
    c->natural_hs_valid = false;
    ++c->tables_gen;
    return VIO_OK;
}

vio_status vio_get_solve_order(vio_ctx *c, int32_t *requested, int32_t *effective) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (requested) *requested = c->solve_order;
    if (effective) *effective = effective_order(c);
    return VIO_OK;
}

#ifdef VIO_DEBUG_ENTRY_POINTS
// Diagnostic: (H + lambda I) x = b by the chain-order kernel alone, on a matrix the caller supplies (171 x 171 row-major, natural
// order of H_pp_schur_).  lds_dump (optional, vio_chain_lds_core_doubles() doubles): the factor as the kernel left it in LDS.
vio_status vio_debug_chain_solve(vio_ctx *c, const double *H, const double *b, double lambda, double *x, double *lds_dump) {
    if (!c || !H || !b || !x) return VIO_ERR_BAD_ARG;
    enter_device(c);
    const int n_img = vio_chain_image_doubles(), n_lds = vio_chain_lds_core_doubles();
    std::vector<double> img((size_t)n_img, 0.0);
    for (int i = 0; i < PD; ++i) {
        img[(size_t)vio_chain_y_offset() + vio_chain_dim(i)] = b[i];
        for (int j = 0; j <= i; ++j) {
            int p1, p2;
            vio_chain_entry_pos(i, j, &p1, &p2);
            const double v = H[(size_t)i * PD + j];
            if (p1 < 0) { if (v != 0.0) return fail(c, VIO_ERR_BAD_ARG, "vio_debug_chain_solve: entry outside the chain pattern"); continue; }
            img[p1] = v;
            if (p2 >= 0) img[p2] = v;
        }
    }
    DevBuf<double> d_img, d_x, d_dump;
    auto body = [&]() -> vio_status {
        HIPCHK(d_img.resize(img.size())); HIPCHK(d_x.resize(176));
        if (lds_dump) HIPCHK(d_dump.resize((size_t)n_lds));
        HIPCHK(hipMemcpyAsync(d_img.p, img.data(), img.size() * 8, hipMemcpyHostToDevice, c->stream));
        vio_launch_chain_solve_test(d_img.p, lambda, d_x.p, lds_dump ? d_dump.p : nullptr, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(x, d_x.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
        if (lds_dump) HIPCHK(hipMemcpyAsync(lds_dump, d_dump.p, (size_t)n_lds * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return VIO_OK;
    };
    const vio_status st = body();
    d_img.release(); d_x.release(); d_dump.release();       // (DevBuf has no destructor: every way out passes here)
    return st;
}
#endif

vio_status vio_linearize(vio_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(activate(c, c->solve_plan, 0));
    if (c->stepwise_updated) c->stepwise_updated = false;     // a new linearisation commits the step
    VIOCHK(enqueue_linearize(c, c->solve_plan));
    c->lin_fresh = true;
    return VIO_OK;
}

vio_status vio_prepare(vio_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c);
    bool any_imu = false;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) any_imu |= c->imu_valid[k];
    if (c->h_olm.empty() && !any_imu) return VIO_OK;        // nothing to prepare (vio_solve will say what it thinks of an empty graph)
    return activate(c, c->solve_plan, 0);
}

vio_status vio_init_lm(vio_ctx *c, double *chi2, double *lambda) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    enter_device(c);
    MAKE_TABLES(T, c, *c->active);
    VIOCHK(enqueue_init_lm(c, T, 1 << 30));
    VIOCHK(read_lm(c));
    if (chi2) *chi2 = c->h_lm.chi;
    if (lambda) *lambda = c->h_lm.lambda;
    return VIO_OK;
}

vio_status vio_solve_linear(vio_ctx *c, double lambda) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    enter_device(c);
    Plan &pl = *c->active;
    MAKE_TABLES(T, c, pl);
    vio_launch_set_lambda(c->d_lm.p, lambda, c->stream);
    vio_launch_pose_solve(T, POSE_SOLVE_LDS, c->stream);
    vio_launch_backsub(T, 0, c->stream);
    HIPCHK(hipGetLastError());
    c->stepwise_updated = false;
    return VIO_OK;
}

vio_status vio_update_states(vio_ctx *c) {
    if (!c || !c->active) return VIO_ERR_BAD_ARG;
    enter_device(c);
    if (!c->stepwise_updated) { vio_launch_flip(c->d_lm.p, c->stream); c->stepwise_updated = true; c->ahead = 7u; }
    return VIO_OK;
}

vio_status vio_rollback_states(vio_ctx *c) {
    if (!c || !c->active) return VIO_ERR_BAD_ARG;
    enter_device(c);
    if (c->stepwise_updated) { vio_launch_flip(c->d_lm.p, c->stream); c->stepwise_updated = false; c->ahead = 7u; }
    return VIO_OK;
}

vio_status vio_chi2(vio_ctx *c, double *chi2) {
    if (!c || !chi2) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(activate(c, c->solve_plan, 0));
    Plan &pl = *c->active;
    MAKE_TABLES(T, c, pl);
    // the chi2 kernels read the pair table of the current state; after a stepwise update it is the trial table
    if (!c->pairtab_valid && pl.lm_dim == 1) { vio_launch_prepare(T, c->stream); c->pairtab_valid = true; }
    vio_launch_backsub(T, 1, c->stream);
    if (sharded(c)) { vio_launch_step_sum(T, 2, c->stream); VIOCHK(run_exchange(c, 1)); vio_launch_lm_decide(T, 2, 0, c->stream); }
    else vio_launch_lm_decide(T, 2, 1, c->stream);
    VIOCHK(read_lm(c));
    *chi2 = c->h_lm.chi_try;
    return VIO_OK;
}

vio_status vio_eval_step(vio_ctx *c, int32_t *accepted, double *chi2, double *lambda) {
    if (!c || !c->active) return VIO_ERR_BAD_ARG;
    enter_device(c);
    Plan &pl = *c->active;
    MAKE_TABLES(T, c, pl);
    // the decide kernel expects the trial copy to be "the other one"
    if (c->stepwise_updated) vio_launch_flip(c->d_lm.p, c->stream);
    if (sharded(c)) { vio_launch_step_sum(T, 0, c->stream); VIOCHK(run_exchange(c, 1)); vio_launch_lm_decide(T, 0, 0, c->stream); }
    else vio_launch_lm_decide(T, 0, 1, c->stream);
    VIOCHK(read_lm(c));
    const bool ok = c->h_lm.accepted != 0;
    if (ok) {
        c->stepwise_updated = false;           // the trial copy is current now; nothing left to roll back to
    } else if (c->stepwise_updated) {
        vio_launch_flip(c->d_lm.p, c->stream);  // stay on the updated states until the caller rolls back (problem.cc:228-231)
    }
    c->ahead = 7u;
    if (accepted) *accepted = ok ? 1 : 0;
    if (chi2) *chi2 = c->h_lm.chi;
    if (lambda) *lambda = c->h_lm.lambda;
    return VIO_OK;
}

vio_status vio_solve(vio_ctx *c, int32_t iterations, vio_solve_report *rep) {
    if (!c) return VIO_ERR_BAD_ARG;
    const bool fresh = c->lin_fresh;
    enter_device(c);
    bool any_imu = false;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) any_imu |= c->imu_valid[k];
    if (c->h_olm.empty() && !any_imu) return fail(c, VIO_ERR_EMPTY, "Cannot solve problem without edges or verticies");
    const auto t0 = std::chrono::steady_clock::now();
    VIOCHK(activate(c, c->solve_plan, 0));
    Plan &pl = c->solve_plan;
    // Problem::Solve opens with MakeHessian (problem.cc:178-183).  When the caller's last call was vio_linearize on this very state and
    // graph, that system is on the device already — the same kernels on the same inputs would write the same bits again — and the solve
    // starts from it (`fresh`, read before enter_device cleared it; VIO_SOLVE_RELINEARIZE=1: always linearise, for A/B).
    static const bool always_lin = std::getenv("VIO_SOLVE_RELINEARIZE") != nullptr;
    const bool reuse = fresh && !always_lin && c->linearized && c->active == &pl;
    float first_lin_ms = c->last_lin_ms;
    if (!reuse) {
        if (!c->lin_ev[0]) { HIPCHK(hipEventCreate(&c->lin_ev[0])); HIPCHK(hipEventCreate(&c->lin_ev[1])); }
        HIPCHK(hipEventRecord(c->lin_ev[0], c->stream));
        VIOCHK(enqueue_linearize(c, pl));
        HIPCHK(hipEventRecord(c->lin_ev[1], c->stream));
    }
    {
        MAKE_TABLES(T, c, pl);
        VIOCHK(enqueue_init_lm(c, T, iterations));
    }
    // Device-driven loop: LmState lives on the device and the kernels do all of Problem::Solve's bookkeeping, so the host
    // enqueues as many slots as outer iterations are left and looks at LmState once per batch of slots; everything after the
    // stop skips itself.  When every trial is accepted - the usual case - that is one read-back per solve.
    // Sharded solves run the same loop: the exchange of a slot is enqueued unconditionally (every rank holds the identical
    // LmState, so every rank enqueues the same collectives), a dead slot all-reduces buffers nobody reads.
    static const bool classic = std::getenv("VIO_LM_CLASSIC") != nullptr;     // diagnostic: the trial / re-linearisation slots of round 1
    vio_status status = VIO_OK;
    if (classic) {
        VIOCHK(read_lm(c));
        while (status == VIO_OK && !c->h_lm.stop && c->h_lm.iter < iterations) {
            // (at most 10 slots ahead: a loop that stops early leaves the rest as empty launches)
            const int batch = std::min(iterations - c->h_lm.iter, 10);
            for (int sl = 0; sl < batch && status == VIO_OK; ++sl) {
                status = enqueue_trial(c, pl, 0, false, 2);
                if (status == VIO_OK) status = enqueue_linearize(c, pl, false, 3);
            }
            if (status == VIO_OK) status = read_lm(c);
        }
    } else {
        int done = 0;
        bool stop = iterations <= 0;
        if (!stop) status = enqueue_lm_slot(c, pl, true);
        while (status == VIO_OK && !stop) {
            const int batch = std::min(iterations - done, 10);
            for (int sl = 0; sl < batch && status == VIO_OK; ++sl) status = enqueue_lm_slot(c, pl, false);
            // host work under the device's; a failure here is vio_marginalize's to report (it tries again)
            if (status == VIO_OK) status = read_lm_begin(c);
            static const bool no_prepare = std::getenv("VIO_NO_MARG_PREPARE") != nullptr;      // diagnostic
            if (status == VIO_OK && done == 0 && !sharded(c) && !no_prepare) {
                const auto tp = std::chrono::steady_clock::now();
                (void)prepare_marg_plan(c);
                c->timing[6] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp).count();
            }
            if (status == VIO_OK) status = read_lm_end(c);
            done = c->h_lm.iter;
            stop = c->h_lm.stop || done >= iterations;
        }
        if (status == VIO_OK && iterations <= 0) status = read_lm(c);
    }
    if (status != VIO_OK) return status;
    if (!reuse && hipEventElapsedTime(&first_lin_ms, c->lin_ev[0], c->lin_ev[1]) == hipSuccess) c->last_lin_ms = first_lin_ms;
    vio_solve_report r;
    std::memset(&r, 0, sizeof(r));
    r.initial_chi2 = c->h_lm.init_chi;
    // t_hessian_cost_ of the reference's printout (problem.cc:246-248): the first linearisation is timed, the others are
    // the same kernels on the same graph
    const int n_lin = 1 + c->h_lm.naccepted - ((c->h_lm.accepted && c->h_lm.stop) ? 1 : 0);
    const double hess_ms = (double)first_lin_ms * n_lin;
    if (!classic || (c->h_lm.accepted && c->h_lm.stop)) c->linearized = false;   // the reference re-linearises here; nobody reads it
    c->natural_hs_valid = c->natural_hs_valid && c->linearized;
    r.iterations = c->h_lm.iter; r.trials = c->h_lm.trials; r.accepted = c->h_lm.naccepted;
    r.stop_reason = c->h_lm.stop_reason;
    r.final_chi2 = c->h_lm.chi; r.final_lambda = c->h_lm.lambda;
    for (int i = 0; i < 128; ++i) { r.chi2_trace[i] = c->h_lm.chi_trace[i]; r.lambda_trace[i] = c->h_lm.lambda_trace[i]; }
    r.hessian_ms = hess_ms;
    r.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep) *rep = r;
    return c->h_lm.finite ? VIO_OK : VIO_ERR_NOT_FINITE;
}

vio_status vio_gn_iteration(vio_ctx *c, double lambda) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(activate(c, c->solve_plan, 0));
    Plan &pl = c->solve_plan;
    // chi2 and the gain-ratio partial of a step reach its test through vis: one exchange per iteration
    const bool gn = true;
    if (gn && c->cur_host < 0) VIOCHK(read_lm(c));      // once: from here on the host tracks LmState.cur itself
    if (lambda != c->gn_lambda) { vio_launch_set_lambda(c->d_lm.p, lambda, c->stream); c->gn_lambda = lambda; }
    VIOCHK(enqueue_linearize(c, pl, gn));
    VIOCHK(enqueue_trial(c, pl, 1, gn));
    if (!gn) c->cur_host = -1;
    return VIO_OK;
}

vio_status vio_get_stream(vio_ctx *c, void **stream) {
    if (!c || !stream) return VIO_ERR_BAD_ARG;
    *stream = (void *)c->stream;
    return VIO_OK;
}

static bool has_duplicates(vio_ctx *const *ctxs, int count) {
    std::vector<const vio_ctx *> v(ctxs, ctxs + count);
    std::sort(v.begin(), v.end());
    return std::adjacent_find(v.begin(), v.end()) != v.end();
}

// B independent windows, one launch per kernel for all of them (grid.y = window): the regime in which the chip is full —
// one window's k_pose_solve is a single workgroup on one of 256 CUs.  Every context keeps its own plan, buffers and
// LmState; the leader (ctxs[0]) holds the device array of the members' tables.
vio_status vio_batch_gn_iteration(vio_ctx *const *ctxs, int32_t count, double lambda) {
    if (!ctxs || count < 1 || !ctxs[0]) return VIO_ERR_BAD_ARG;
    vio_ctx *c = ctxs[0];       // leader: errors are reported on it
    enter_device(c);
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        if (!m) return fail(c, VIO_ERR_BAD_ARG, "vio_batch_gn_iteration: null context");
        m->lin_fresh = false;
        if (m->cfg.device != c->cfg.device || m->stream != c->stream)
            return fail(c, VIO_ERR_BAD_ARG, "vio_batch_gn_iteration: the contexts must share one device and one stream (vio_config.stream; vio_get_stream)");
        if (sharded(m)) return fail(c, VIO_ERR_UNSUPPORTED, "vio_batch_gn_iteration: sharded contexts cannot be batched");
        if (m->lm_dim != c->lm_dim) return fail(c, VIO_ERR_UNSUPPORTED, "vio_batch_gn_iteration: the windows of a batch hold one kind of landmark");
    }
    if (has_duplicates(ctxs, count)) return fail(c, VIO_ERR_BAD_ARG, "vio_batch_gn_iteration: a context appears twice (two windows of the grid would write the same buffers)");
    // a member's failure is reported on the leader, which is where the caller looks (vio_last_error(ctxs[0]))
    auto member = [&](int i, vio_status st) { return (st == VIO_OK || ctxs[i] == c) ? st : fail(c, st, "window " + std::to_string(i) + ": " + ctxs[i]->err); };
    bool rebuild = (int)c->batch_members.size() != count;
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        const uint64_t g0 = m->tables_gen;
        vio_status st = member(i, activate(m, m->solve_plan, 0));
        if (st != VIO_OK) return st;
        if (!rebuild && (c->batch_members[i] != m || c->batch_gens[i] != m->tables_gen || g0 != m->tables_gen)) rebuild = true;
        // a step waiting for its test belongs to the batch's own sequence only if the array is current; otherwise settle it
        if (m->cur_host < 0 || (rebuild && m->decide_pending)) { st = member(i, read_lm(m)); if (st != VIO_OK) return st; }
        // The array holds every window's `cur` as it was when the array was built; the kernels flip it by the parity of the batch's
        // own iteration count.  Anything that moved a member's `cur` outside the batch (vio_gn_iteration, vio_solve, the stepwise
        // flips on that one context) leaves that bookkeeping behind: build the array again from the members as they are.
        if (!rebuild && m->cur_host != (c->batch_cur0[i] ^ (c->batch_iters & 1))) rebuild = true;
        if (!m->pairtab_valid && m->lm_dim == 1) { DeviceTables T = make_tables_raw(m, m->solve_plan); T.cur_hint = m->cur_host; vio_launch_prepare(T, m->stream); m->pairtab_valid = true; }
        if (lambda != m->gn_lambda) { vio_launch_set_lambda(m->d_lm.p, lambda, m->stream); m->gn_lambda = lambda; }
    }
    if (!rebuild)
        for (int i = 0; i < count; ++i) if (ctxs[i]->decide_pending != ctxs[0]->decide_pending) rebuild = true;
    if (rebuild) {
        for (int i = 0; i < count; ++i)
            if (ctxs[i]->decide_pending || ctxs[i]->cur_host < 0) { const vio_status st = member(i, read_lm(ctxs[i])); if (st != VIO_OK) return st; }
        std::vector<DeviceTables> tabs((size_t)count);
        c->batch_members.assign(ctxs, ctxs + count);
        c->batch_gens.resize((size_t)count);
        c->batch_cur0.resize((size_t)count);
        for (int i = 0; i < count; ++i) {
            tabs[i] = make_tables_raw(ctxs[i], ctxs[i]->solve_plan);
            tabs[i].cur_hint = ctxs[i]->cur_host;         // this window's `cur` now; the kernels flip it by the iteration parity
            c->batch_gens[i] = ctxs[i]->tables_gen;
            c->batch_cur0[i] = ctxs[i]->cur_host;
        }
        HIPCHK(c->d_batch_tabs.resize((size_t)count));
        HIPCHK(hipMemcpyAsync(c->d_batch_tabs.p, tabs.data(), (size_t)count * sizeof(DeviceTables), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));           // `tabs` goes out of scope
        c->batch_iters = 0;
    }
    int max_blocks = 1, any_prior = 0, any_ext = 0;
    int lin_threads = lin_threads_half_host();               // half-width workgroups only if every window's plan was sized for them
    size_t lds = 0;
    for (int i = 0; i < count; ++i) {
        const Plan &pl = ctxs[i]->solve_plan;
        if (pl.lin_threads != lin_threads_half_host()) lin_threads = lin_threads_host();
        any_ext |= pl.use_ext;
        max_blocks = std::max<int>(max_blocks, (int)pl.items.size() + VIO_WINDOW_SIZE);
        lds = std::max(lds, (size_t)pl.max_lds_doubles * 8);
        any_prior |= ctxs[i]->has_prior;
    }
    if (any_ext) lin_threads = -lin_threads;                 // (the launchers' sign convention: kernels that read the extrinsic block from the item)
    const int test_prev = ctxs[0]->decide_pending ? 1 : 0;
    int order = VIO_ORDER_CHAIN;                             // the chain order if every window can take it (one kernel for the whole batch)
    for (int i = 0; i < count; ++i) if (effective_order(ctxs[i]) != VIO_ORDER_CHAIN) order = VIO_ORDER_EIGEN;
    for (int i = 0; i < count; ++i) use_pg_layout(ctxs[i], order);
    // vio_profile_begin on the leader: an event pair around that kernel's batched launch (every prof_every-th iteration)
    int ev_kernel = -1;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // (a pair is taken only for a kernel this order launches: k_linearize, k_reduce, k_pose_solve in both, k_assemble in Eigen's order only;
    // a pair consumed but never recorded would make vio_profile_end read events nobody recorded, or stale ones)
    const bool prof_marks = c->prof_which == VIO_K_LINEARIZE || c->prof_which == VIO_K_REDUCE || c->prof_which == VIO_K_POSE_SOLVE ||
                            (c->prof_which == VIO_K_ASSEMBLE && order == VIO_ORDER_EIGEN);
    if (prof_marks && c->prof_seen++ % c->prof_every == 0) {
        bool ok = true;
        while (ok && c->prof_used + 2 > c->prof_events.size()) { hipEvent_t e; ok = hipEventCreate(&e) == hipSuccess; if (ok) c->prof_events.push_back(e); }
        if (ok) { ev[0] = c->prof_events[c->prof_used]; ev[1] = c->prof_events[c->prof_used + 1]; ev_kernel = c->prof_which; c->prof_used += 2; }
    }
    vio_launch_batch_gn(c->d_batch_tabs.p, count, c->lm_dim, max_blocks, lds, lin_threads, test_prev, any_prior, c->batch_iters & 1, POSE_SOLVE_LDS, order, c->stream,
                        ev_kernel, ev);
    HIPCHK(hipGetLastError());
    ++c->batch_iters;
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        m->decide_pending = true;
        m->cur_host ^= 1;
        m->ahead = 7u;
        m->linearized = order == effective_order(m);       // (a window solved in the batch's order, not its own: its image is the other layout)
        m->natural_hs_valid = false;
    }
    return VIO_OK;
}

// Problem::Solve for `count` independent windows at once: vio_solve's device-driven loop with every kernel launched once for the
// whole batch (grid.y = window).  Every window follows its own LmState — its own lambda, its own accept / reject decisions,
// its own stop — and the kernels of a window that has stopped skip themselves; the host reads the LmStates once per batch of
// slots.  The same kernel bodies on the same data as vio_solve: bit-identical results.
vio_status vio_batch_solve(vio_ctx *const *ctxs, int32_t count, int32_t iterations, vio_solve_report *reports) {
    if (!ctxs || count < 1 || !ctxs[0] || iterations < 0) return VIO_ERR_BAD_ARG;
    vio_ctx *c = ctxs[0];       // leader: errors are reported on it
    enter_device(c);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        if (!m) return fail(c, VIO_ERR_BAD_ARG, "vio_batch_solve: null context");
        m->lin_fresh = false;
        if (m->cfg.device != c->cfg.device || m->stream != c->stream)
            return fail(c, VIO_ERR_BAD_ARG, "vio_batch_solve: the contexts must share one device and one stream (vio_config.stream; vio_get_stream)");
        if (sharded(m)) return fail(c, VIO_ERR_UNSUPPORTED, "vio_batch_solve: sharded contexts cannot be batched");
        if (m->lm_dim != c->lm_dim) return fail(c, VIO_ERR_UNSUPPORTED, "vio_batch_solve: the windows of a batch hold one kind of landmark");
        bool any_imu = false;
        for (int k = 0; k < VIO_WINDOW_SIZE; ++k) any_imu |= m->imu_valid[k];
        if (m->h_olm.empty() && !any_imu) return fail(c, VIO_ERR_EMPTY, "window " + std::to_string(i) + ": Cannot solve problem without edges or verticies");
    }
    if (has_duplicates(ctxs, count)) return fail(c, VIO_ERR_BAD_ARG, "vio_batch_solve: a context appears twice (two windows of the grid would write the same buffers)");
    std::vector<DeviceTables> tabs((size_t)count);
    int max_blocks = 1, any_prior = 0, any_ext = 0;
    int lin_threads = lin_threads_half_host();
    size_t lds = 0;
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        vio_status st = activate(m, m->solve_plan, 0);
        if (st != VIO_OK) return m == c ? st : fail(c, st, "window " + std::to_string(i) + ": " + m->err);
        st = flush_decide(m);
        if (st != VIO_OK) return m == c ? st : fail(c, st, "window " + std::to_string(i) + ": " + m->err);
        m->cur_host = -1;
        tabs[i] = make_tables_raw(m, m->solve_plan);         // cur_hint = -1: the kernels take `cur` from the window's LmState
        if (!m->pairtab_valid && m->lm_dim == 1) { vio_launch_prepare(tabs[i], m->stream); m->pairtab_valid = true; }
        const Plan &pl = m->solve_plan;
        if (pl.lin_threads != lin_threads_half_host()) lin_threads = lin_threads_host();
        any_ext |= pl.use_ext;
        max_blocks = std::max<int>(max_blocks, (int)pl.items.size() + VIO_WINDOW_SIZE);
        lds = std::max(lds, (size_t)pl.max_lds_doubles * 8);
        any_prior |= m->has_prior;
    }
    c->batch_members.clear();                                // the GN batch's cached array is overwritten
    HIPCHK(c->d_batch_tabs.resize((size_t)count));
    HIPCHK(hipMemcpyAsync(c->d_batch_tabs.p, tabs.data(), (size_t)count * sizeof(DeviceTables), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the windows still running: a window that has stopped leaves the launches (its workgroups would each hold a CU for the
    // round trip that tells them so), the array on the device is rewritten with the ones that go on
    std::vector<int> live((size_t)count);
    for (int i = 0; i < count; ++i) live[i] = i;
    auto read_all = [&]() -> vio_status {
        for (int i : live) HIPCHK(hipMemcpyAsync(&ctxs[i]->h_lm, ctxs[i]->d_lm.p, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return VIO_OK;
    };
    static const bool classic = std::getenv("VIO_LM_CLASSIC") != nullptr;
    if (any_ext) lin_threads = -lin_threads;                 // (the launchers' sign convention, as in vio_batch_gn_iteration)
    int order = VIO_ORDER_CHAIN;
    for (int i = 0; i < count; ++i) if (effective_order(ctxs[i]) != VIO_ORDER_CHAIN) order = VIO_ORDER_EIGEN;
    for (int i = 0; i < count; ++i) use_pg_layout(ctxs[i], order);
    vio_launch_batch_lm(c->d_batch_tabs.p, count, c->lm_dim, max_blocks, lds, lin_threads, any_prior, POSE_SOLVE_LDS, 0, iterations, order, c->stream);
    if (!classic && iterations > 0) vio_launch_batch_lm(c->d_batch_tabs.p, count, c->lm_dim, max_blocks, lds, lin_threads, any_prior, POSE_SOLVE_LDS, 3, iterations, order, c->stream);
    HIPCHK(hipGetLastError());
    if (classic || iterations <= 0) VIOCHK(read_all());
    else for (int i = 0; i < count; ++i) { ctxs[i]->h_lm.stop = 0; ctxs[i]->h_lm.iter = 0; }
    for (;;) {
        int left = 0;           // slots the slowest window may still need
        for (int i = 0; i < count; ++i)
            if (!ctxs[i]->h_lm.stop && ctxs[i]->h_lm.iter < iterations) left = std::max(left, iterations - ctxs[i]->h_lm.iter);
        if (left == 0) break;
        {
            std::vector<int> keep;
            for (int i : live) if (!ctxs[i]->h_lm.stop && ctxs[i]->h_lm.iter < iterations) keep.push_back(i);
            if (keep.size() != live.size()) {
                std::vector<DeviceTables> sub;
                for (int i : keep) sub.push_back(tabs[i]);
                HIPCHK(hipMemcpyAsync(c->d_batch_tabs.p, sub.data(), sub.size() * sizeof(DeviceTables), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));           // `sub` goes out of scope
                live.swap(keep);
            }
        }
        const int batch = std::min(left, 10);
        for (int sl = 0; sl < batch; ++sl)
            vio_launch_batch_lm(c->d_batch_tabs.p, (int)live.size(), c->lm_dim, max_blocks, lds, lin_threads, any_prior, POSE_SOLVE_LDS, classic ? 1 : 2, iterations, order, c->stream);
        HIPCHK(hipGetLastError());
        VIOCHK(read_all());
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    bool finite = true;
    for (int i = 0; i < count; ++i) {
        vio_ctx *m = ctxs[i];
        m->ahead = 7u;
        m->decide_pending = false;
        m->linearized = classic && !(m->h_lm.accepted && m->h_lm.stop);
        m->natural_hs_valid = false;
        m->stepwise_updated = false;
        finite = finite && m->h_lm.finite;
        if (reports) {
            vio_solve_report r;
            std::memset(&r, 0, sizeof(r));
            r.initial_chi2 = m->h_lm.init_chi;
            r.iterations = m->h_lm.iter; r.trials = m->h_lm.trials; r.accepted = m->h_lm.naccepted; r.stop_reason = m->h_lm.stop_reason;
            r.final_chi2 = m->h_lm.chi; r.final_lambda = m->h_lm.lambda;
            for (int k = 0; k < 128; ++k) { r.chi2_trace[k] = m->h_lm.chi_trace[k]; r.lambda_trace[k] = m->h_lm.lambda_trace[k]; }
            r.solve_ms = ms;        // the whole batch's wall clock
            reports[i] = r;
        }
    }
    return finite ? VIO_OK : VIO_ERR_NOT_FINITE;
}

vio_status vio_synchronize(vio_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c, true);
    HIPCHK(hipStreamSynchronize(c->stream));
    return VIO_OK;
}

// The marginalisation in two halves (include/vio_backend.h).  begin: the device part — plan, kernels, the 171x171 read-back — and
// the hand-over of the dense tail (problem.cc:717-779: two eigen-decompositions, three products) to a helper thread; end: wait for
// it and copy the prior out.  Between the two the caller owns the context as usual (the helper touches host memory of its own only).
static void marg_tail_job(vio_ctx *c, int frame) {
    const auto t1 = std::chrono::steady_clock::now();
    double *Hm = c->marg_stage, *bm = c->marg_stage + (size_t)PD * PD;
    MargResult &r = c->marg_out;
    // A landmark block without an inverse (h_ll == 0; a 3x3 block of XYZ landmarks that the marginalisation graph left at rank 2 and
    // whose elimination hit an exact zero pivot) makes the reference's dense Hpm * Hmm^-1 (problem.cc:701-703) non-finite everywhere;
    // its two eigen-solvers then return NaN spectra, every `> eps` test fails (:747-769), the 1e-9 cut zeroes H_prior_ (:778) and
    // b_prior_, err_prior_, Jt_prior_inv_ are NaN.  The same outcome here, said in the status as well.
    bool finite_in = true;
    for (size_t i = 0; i < (size_t)PD * PD + PD && finite_in; ++i) finite_in = std::isfinite(Hm[i]);
    r.live_rows = 0;
    if (finite_in) {
        // The tail's row-parallel parts and the eigen-solver's pipeline can use the process's helper threads (the same bits whatever their
        // number: tests/test_host_units.py) — measured on the round's host, a two-socket EPYC 9575F, and left OFF: one thread 290 us for
        // the 75-row eigen-problem, two 250, three and more 650 .. 1 000 (profiles/r06d_host_tail_threads.txt: the rotation stream and the
        // eigenvector matrix travel between cores that share no cache).  VIO_MARG_THREADS=n turns them on.
        static const int marg_threads = std::getenv("VIO_MARG_THREADS") ? std::atoi(std::getenv("VIO_MARG_THREADS")) : 1;
        vio_plan::HostPool *hp = marg_threads > 1 ? vio_plan::shared_pool() : nullptr;
        vio_host::Par par{hp, [](void *ctx, int want, void (*fn)(void *, int, int), void *arg) { vio_plan::pool_run_n((vio_plan::HostPool *)ctx, want, fn, arg); },
                          std::min(marg_threads, vio_plan::pool_width(hp))};
        r.live_rows = vio_host::marginalize_tail(Hm, bm, frame, r.H.data(), r.b.data(), r.err.data(), r.jt.data(), hp ? &par : nullptr);
    }
    else {
        const double nan = std::nan("");
        std::fill(r.H.begin(), r.H.end(), 0.0); std::fill(r.jt.begin(), r.jt.end(), nan);
        std::fill(r.b.begin(), r.b.end(), nan); std::fill(r.err.begin(), r.err.end(), nan);
    }
    r.finite = finite_in;
    r.tail_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
}

// The tail runs on the process's ONE background worker (vio_plan::bg_submit: created with the first job, parked between jobs, joined when the
// last context goes; round 6 — until then every context had a worker thread of its own) or, for the synchronous vio_marginalize, on the
// caller's thread; either way its row-parallel parts use the process's helper threads when they are free.  Without threads everything runs on
// the caller: no exception crosses the C boundary.
static void marg_job_fn(void *arg) {
    vio_ctx *c = (vio_ctx *)arg;
    marg_tail_job(c, c->marg_job_frame);
}
static void marg_join(vio_ctx *c) { vio_plan::bg_wait(&c->marg_ticket); }
static void marg_submit(vio_ctx *c, int frame, bool inline_tail) {
    c->marg_job_frame = frame;
    if (inline_tail) { marg_tail_job(c, frame); return; }
    vio_plan::bg_submit(&c->marg_ticket, marg_job_fn, c);
}

static vio_status marginalize_begin_impl(vio_ctx *c, int32_t kind, bool inline_tail);
vio_status vio_marginalize_begin(vio_ctx *c, int32_t kind) { return marginalize_begin_impl(c, kind, false); }
static vio_status marginalize_begin_impl(vio_ctx *c, int32_t kind, bool inline_tail) {

    if (!c) return VIO_ERR_BAD_ARG;
    if (kind != VIO_MARG_OLD && kind != VIO_MARG_SECOND_NEW) return VIO_ERR_BAD_ARG;
    enter_device(c);
    marg_join(c);                     // (a marginalisation nobody asked the result of: its buffers are about to be reused)
    c->marg_pending = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!c->marg_stage) HIPCHK(hipHostMalloc((void **)&c->marg_stage, ((size_t)PD * PD + PD) * 8, hipHostMallocDefault));      // pinned: the 234 KB come back by DMA
    double *Hm = c->marg_stage, *bm = c->marg_stage + (size_t)PD * PD;
    if (kind == VIO_MARG_OLD) {
        // Straight after a solve (the frame loop) everything the marginalisation graph needs is on the device already, at the
        // current slot of the double buffers: only the plan's tables are new, and its landmarks (those hosted in frame 0) are a
        // gather out of the solve plan's.  No read-back of the window, no second upload of it.
        const bool resident = c->active == &c->solve_plan && c->solve_plan.valid && !c->dirty_inputs && !c->topo_dirty && c->lm_dim == 1 &&
                              !c->stepwise_updated;
        if (resident) {
            Plan &mp = c->marg_plan, &sp = c->solve_plan;
            VIOCHK(flush_decide(c));
            VIOCHK(prepare_marg_plan(c));             // (vio_solve has done this while the device was busy)
            if (mp.Ns) vio_launch_gather_landmarks(c->d_lm.p, sp.d_invd.p, (int)sp.Ns, mp.d_invd.p, (int)mp.Ns, c->d_gather_map.p, c->stream);
            ++c->tables_gen;
            c->active = &mp;
            c->linearized = false;
            c->pairtab_valid = false;
            c->gn_lambda = -1.0;
        } else {
            VIOCHK(activate(c, c->marg_plan, 1));
        }
        VIOCHK(enqueue_linearize(c, c->marg_plan));
        HIPCHK(hipMemcpyAsync(Hm, c->d_Hs.p, (size_t)PD * PD * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(bm, c->d_bs.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->linearized = false;
    } else {
        // MargNewFrame builds a graph without edges (estimator.cpp:830-901): H_marg is the prior alone
        VIOCHK(pull_from_device(c, 4u));
        std::memcpy(Hm, c->h_Hprior.data(), (size_t)PD * PD * 8);
        std::memcpy(bm, c->h_bprior.data(), (size_t)PD * 8);
    }
    MargResult &r = c->marg_out;
    r.H.resize((size_t)PRD * PRD); r.jt.resize((size_t)PRD * PRD); r.b.resize(PRD); r.err.resize(PRD);
    r.kind = kind;
    c->timing[3] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    c->marg_pending = true;
    marg_submit(c, kind == VIO_MARG_OLD ? 0 : VIO_WINDOW_SIZE - 1, inline_tail);
    return VIO_OK;
}

vio_status vio_marginalize_end(vio_ctx *c, double *H, double *b, double *err, double *jt) {
    if (!c || !H || !b || !err || !jt) return VIO_ERR_BAD_ARG;
    if (!c->marg_pending) return fail(c, VIO_ERR_BAD_ARG, "vio_marginalize_end without vio_marginalize_begin");
    marg_join(c);
    c->marg_pending = false;
    const MargResult &r = c->marg_out;
    std::memcpy(H, r.H.data(), (size_t)PRD * PRD * 8); std::memcpy(jt, r.jt.data(), (size_t)PRD * PRD * 8);
    std::memcpy(b, r.b.data(), PRD * 8); std::memcpy(err, r.err.data(), PRD * 8);
    c->timing[4] = r.tail_us;
    c->timing[5] = r.live_rows;
    static const bool timing = std::getenv("VIO_HOST_TIMING") != nullptr;
    if (timing) std::fprintf(stderr, "[vio host timing] vio_marginalize(kind=%d): activate + kernels + read-back %.0f us, host tail %.0f us\n", r.kind, c->timing[3], r.tail_us);
    if (!r.finite) return fail(c, VIO_ERR_NOT_FINITE, "vio_marginalize: a landmark block has no inverse; the prior is the reference's outcome for that case (H_prior 0, the rest NaN)");
    return VIO_OK;
}

vio_status vio_marginalize(vio_ctx *c, int32_t kind, double *H, double *b, double *err, double *jt) {
    if (!c || !H || !b || !err || !jt) return VIO_ERR_BAD_ARG;
    VIOCHK(marginalize_begin_impl(c, kind, true));         // (the caller waits for the prior anyway: its thread runs the tail, no hand-over)
    return vio_marginalize_end(c, H, b, err, jt);
}

vio_status vio_get_window(vio_ctx *c, double *poses, double *sb, double *ext) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(pull_from_device(c, 1u));
    if (ext) std::memcpy(ext, c->h_state + STATE_EXT, 7 * 8);
    if (poses) std::memcpy(poses, c->h_state + STATE_POSE, 77 * 8);
    if (sb) std::memcpy(sb, c->h_state + STATE_SB, 99 * 8);
    return VIO_OK;
}

vio_status vio_get_landmarks(vio_ctx *c, int64_t n, double *invd) {
    if (!c || c->lm_dim != 1 || n != (int64_t)c->h_invd.size() || (n > 0 && !invd)) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(pull_from_device(c, 2u));
    if (n) std::memcpy(invd, c->h_invd.data(), (size_t)n * 8);
    return VIO_OK;
}

vio_status vio_get_landmarks_xyz(vio_ctx *c, int64_t n, double *xyz) {
    if (!c || c->lm_dim != 3 || 3 * n != (int64_t)c->h_invd.size() || (n > 0 && !xyz)) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(pull_from_device(c, 2u));
    if (n) std::memcpy(xyz, c->h_invd.data(), (size_t)n * 24);
    return VIO_OK;
}

vio_status vio_get_prior(vio_ctx *c, double *b, double *err) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(pull_from_device(c, 4u));
    if (b) std::memcpy(b, c->h_bprior.data(), PD * 8);
    if (err) std::memcpy(err, c->h_errprior.data(), PRD * 8);
    return VIO_OK;
}

vio_status vio_get_delta(vio_ctx *c, double *dxp, int64_t n, double *dxl) {
    if (!c || !c->active) return VIO_ERR_BAD_ARG;
    enter_device(c);
    const size_t ld = (size_t)c->lm_dim;
    if (dxl && n * (int64_t)ld != (int64_t)c->h_invd.size()) return VIO_ERR_BAD_ARG;
    VIOCHK(flush_decide(c));        // a GN step's landmark update is owed until somebody asks (or the next linearisation)
    Plan &pl = *c->active;
    if (dxp) HIPCHK(hipMemcpyAsync(dxp, c->d_dx.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
    std::vector<double> tmp(ld * (size_t)std::max<int64_t>(pl.Ns, 1));
    if (dxl && pl.Ns) HIPCHK(hipMemcpyAsync(tmp.data(), pl.d_dxl.p, ld * (size_t)pl.Ns * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (dxl) {
        for (int64_t l = 0; l < n * (int64_t)ld; ++l) dxl[l] = 0.0;
        for (int64_t s = 0; s < pl.Ns; ++s)
            for (size_t k = 0; k < ld; ++k) dxl[ld * pl.sorted_to_orig[s] + k] = tmp[k * pl.Ns + s];
    }
    return VIO_OK;
}

vio_status vio_get_schur_system(vio_ctx *c, double *H, double *b) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    enter_device(c);
    if (H && !c->natural_hs_valid) {            // the solve path keeps only the permuted packed copy: assemble once more
        c->want_natural_hs = true;
        MAKE_TABLES(T, c, *c->active);
        vio_launch_assemble(T, c->stream);
        c->want_natural_hs = false;
        c->natural_hs_valid = true;
    }
    if (H) HIPCHK(hipMemcpyAsync(H, c->d_Hs.p, (size_t)PD * PD * 8, hipMemcpyDeviceToHost, c->stream));
    if (b) HIPCHK(hipMemcpyAsync(b, c->d_bs.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VIO_OK;
}

vio_status vio_get_landmark_system(vio_ctx *c, int64_t n, double *hll, double *bl) {
    if (!c || !c->linearized || n * c->lm_dim != (int64_t)c->h_invd.size()) return VIO_ERR_BAD_ARG;
    enter_device(c);
    VIOCHK(flush_decide(c));
    Plan &pl = *c->active;
    std::vector<double> lw(std::max<size_t>(pl.lw_doubles, 1));
    if (pl.lw_doubles) HIPCHK(hipMemcpyAsync(lw.data(), pl.d_lw.p, pl.lw_doubles * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (pl.lm_dim == 3) {       // H_ll as 3x3 row-major from its 6 distinct entries, b_l (3)
        for (int64_t l = 0; l < n; ++l) { if (hll) for (int k = 0; k < 9; ++k) hll[9 * l + k] = 0; if (bl) for (int k = 0; k < 3; ++k) bl[3 * l + k] = 0; }
        static const int sym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
        for (const ItemDesc &it : pl.items)
            for (int g = 0; g < it.G; ++g) {
                const int32_t l = pl.sorted_to_orig[it.lm_base + g];
                const size_t f0 = (size_t)it.lw_base + g;
                if (hll) for (int k = 0; k < 9; ++k) hll[9 * l + k] = lw[f0 + (size_t)sym[k] * it.G];
                if (bl) for (int k = 0; k < 3; ++k) bl[3 * l + k] = lw[f0 + (size_t)(6 + k) * it.G];
            }
        return VIO_OK;
    }
    for (int64_t l = 0; l < n; ++l) { if (hll) hll[l] = 0; if (bl) bl[l] = 0; }
    for (const ItemDesc &it : pl.items)
        for (int g = 0; g < it.G; ++g) {
            const int32_t l = pl.sorted_to_orig[it.lm_base + g];
            if (hll) hll[l] = lw[(size_t)it.lw_base + (size_t)(6 * it.nb) * it.G + g];
            if (bl) bl[l] = lw[(size_t)it.lw_base + (size_t)(6 * it.nb + 1) * it.G + g];
        }
    return VIO_OK;
}

vio_status vio_get_pose_gradient(vio_ctx *c, double *b, double *diag) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    enter_device(c);
    if (b) HIPCHK(hipMemcpyAsync(b, c->d_bfull.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
    if (diag) HIPCHK(hipMemcpyAsync(diag, c->d_diagfull.p, PD * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return VIO_OK;
}

vio_status vio_exchange_buffers(vio_ctx *c, void **reduced, int64_t *n_reduced, void **scalars, int64_t *n_scalars) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (reduced) *reduced = c->ext_vis ? c->ext_vis : c->d_vis.p;
    if (n_reduced) *n_reduced = VIS_SEND;       // the slots summed over the shards, then max |h_ll|
    if (scalars) *scalars = c->ext_step ? c->ext_step : c->d_step_tot.p;
    if (n_scalars) *n_scalars = 2;
    return VIO_OK;
}

int32_t vio_abi_version(void) { return VIO_ABI_VERSION; }

vio_status vio_set_exchange_hook(vio_ctx *c, vio_exchange_fn fn, void *user) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->hook = fn;
    c->hook_user = user;
    ++c->tables_gen;
    return VIO_OK;
}

vio_status vio_triangulate(vio_ctx *c, int64_t n, const int32_t *start_frame, const int64_t *obs_offset, const double *pts,
                           const double *poses, const double *ext, double init_depth, double *depth) {
    if (!c || n < 0 || (n > 0 && (!start_frame || !obs_offset || !depth)) || !poses || !ext) return VIO_ERR_BAD_ARG;
    if (n == 0) return VIO_OK;
    const int64_t m = obs_offset[n];
    if (m < 0 || (m > 0 && !pts)) return VIO_ERR_BAD_ARG;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t k = obs_offset[i + 1] - obs_offset[i];
        if (k < 0 || start_frame[i] < 0 || start_frame[i] + k > VIO_NF) return fail(c, VIO_ERR_BAD_ARG, "vio_triangulate: a track leaves the window");
    }
    enter_device(c);
    DevBuf<int32_t> d_sf; DevBuf<int64_t> d_off; DevBuf<double> d_pts, d_pose, d_depth;
    vio_status st = VIO_OK;
    auto body = [&]() -> vio_status {
        HIPCHK(d_sf.resize(n)); HIPCHK(d_off.resize(n + 1)); HIPCHK(d_pts.resize(2 * (size_t)m)); HIPCHK(d_pose.resize(7 * VIO_NF + 7));
        HIPCHK(d_depth.resize(n));
        HIPCHK(hipMemcpyAsync(d_sf.p, start_frame, n * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_off.p, obs_offset, (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (m) HIPCHK(hipMemcpyAsync(d_pts.p, pts, 2 * (size_t)m * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_pose.p, poses, 7 * VIO_NF * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_pose.p + 7 * VIO_NF, ext, 7 * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_depth.p, depth, n * 8, hipMemcpyHostToDevice, c->stream));
        TriTables Q{n, d_sf.p, d_off.p, d_pts.p, d_pose.p, d_pose.p + 7 * VIO_NF, init_depth, d_depth.p};
        vio_launch_triangulate(Q, c->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(depth, d_depth.p, n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return VIO_OK;
    };
    st = body();
    d_sf.release(); d_off.release(); d_pts.release(); d_pose.release(); d_depth.release();
    return st;
}

vio_status vio_comm_unique_id(void *id128) {
    if (!id128) return VIO_ERR_BAD_ARG;
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return VIO_ERR_HIP;
    return api->GetUniqueId((RcclId128 *)id128) == 0 ? VIO_OK : VIO_ERR_HIP;
}

vio_status vio_comm_destroy(vio_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (c->comm) {
        std::string err;
        if (RcclApi *api = rccl_api(err)) { hipStreamSynchronize(c->stream); api->CommDestroy(c->comm); }
        c->comm = nullptr;
    }
    return VIO_OK;
}

vio_status vio_comm_init(vio_ctx *c, const void *id128, int32_t rank, int32_t nranks) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return VIO_ERR_BAD_ARG;
    if (rank != c->cfg.shard_rank || nranks != c->cfg.shard_count) return fail(c, VIO_ERR_BAD_ARG, "vio_comm_init: rank/nranks differ from the context's shard_rank/shard_count");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(c, VIO_ERR_HIP, err);
    vio_comm_destroy(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    RcclId128 id;
    std::memcpy(&id, id128, sizeof(id));
    const int rc = api->CommInitRank(&c->comm, nranks, id, rank);
    if (rc != 0) { c->comm = nullptr; return fail(c, VIO_ERR_HIP, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error")); }
    c->linearized = false;
    return VIO_OK;
}

// What RCCL itself says about the communicator of the native exchange: the number of ranks it spans and this rank's place in it
// (ncclCommCount / ncclCommUserRank) — the N > 1 bench line records them, so that a scaling record shows how many ranks the collective
// really ran over.  No communicator (unsharded, or the hook exchange): 0 ranks.
vio_status vio_comm_info(vio_ctx *c, int32_t *nranks_seen, int32_t *rank_seen) {
    if (!c) return VIO_ERR_BAD_ARG;
    int n = 0, r = -1;
    if (c->comm) {
        std::string err;
        RcclApi *api = rccl_api(err);
        if (!api || !api->CommCount || !api->CommUserRank) return fail(c, VIO_ERR_HIP, "librccl.so lacks ncclCommCount / ncclCommUserRank");
        if (api->CommCount(c->comm, &n) != 0 || api->CommUserRank(c->comm, &r) != 0) return fail(c, VIO_ERR_HIP, "ncclCommCount failed");
    }
    if (nranks_seen) *nranks_seen = n;
    if (rank_seen) *rank_seen = r;
    return VIO_OK;
}

vio_status vio_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int32_t count,
                            const double *dt, const double *acc, const double *gyr, double acc_n, double gyr_n,
                            double acc_w, double gyr_w, vio_preint *out) {
    if (!acc0 || !gyr0 || !ba || !bg || !out || count < 0 || (count > 0 && (!dt || !acc || !gyr))) return VIO_ERR_BAD_ARG;
    vio_host::preintegrate(acc0, gyr0, ba, bg, count, dt, acc, gyr, acc_n, gyr_n, acc_w, gyr_w, &out->sum_dt, out->delta_p,
                           out->delta_q, out->delta_v, out->jacobian, out->covariance);
    for (int k = 0; k < 3; ++k) { out->linearized_ba[k] = ba[k]; out->linearized_bg[k] = bg[k]; }
    return VIO_OK;
}

vio_status vio_gather_buffers(vio_ctx *c, void **gathered_system, void **gathered_scalars) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (gathered_system) *gathered_system = c->ext_gath ? c->ext_gath : c->d_gath.p;
    if (gathered_scalars) *gathered_scalars = c->ext_step_gath ? c->ext_step_gath : c->d_step_gath.p;
    c->gather_known = true;
    return VIO_OK;
}

vio_status vio_bind_gather_buffers(vio_ctx *c, void *gathered_system, void *gathered_scalars) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->ext_gath = (double *)gathered_system;
    c->ext_step_gath = (double *)gathered_scalars;
    c->gather_known = true;
    ++c->tables_gen;
    c->linearized = false;
    return VIO_OK;
}

vio_status vio_bind_exchange_buffers(vio_ctx *c, void *reduced, void *scalars) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->ext_vis = (double *)reduced;
    c->ext_step = (double *)scalars;
    ++c->tables_gen;
    c->linearized = false;
    return VIO_OK;
}

#ifdef VIO_STAMPS
// diagnostic build: copy out the s_memtime stamps of the last k_linearize launch ([block][16])
__attribute__((visibility("default"))) vio_status vio_debug_stamps(vio_ctx *c, unsigned long long *out, int64_t n_blocks) {
    if (!c || !c->d_dbg.p) return VIO_ERR_BAD_ARG;
    enter_device(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_dbg.p, (size_t)n_blocks * 16 * 8, hipMemcpyDeviceToHost));
    return VIO_OK;
}
#endif

const char *vio_kernel_name(int32_t which) {
    static const char *names[VIO_K_COUNT] = {"k_linearize", "k_reduce", "k_assemble", "k_pose_solve", "k_backsub", "k_lm_decide"};
    return (which >= 0 && which < VIO_K_COUNT) ? names[which] : "";
}

vio_status vio_get_host_timing(vio_ctx *c, double *out8) {
    if (!c || !out8) return VIO_ERR_BAD_ARG;
    std::memcpy(out8, c->timing, sizeof(c->timing));
    return VIO_OK;
}

vio_status vio_profile_begin_sampled(vio_ctx *c, int32_t which, int32_t every) {
    if (!c || every < 1) return VIO_ERR_BAD_ARG;
    vio_status s = vio_profile_begin(c, which);
    c->prof_every = every;
    return s;
}

vio_status vio_profile_begin(vio_ctx *c, int32_t which) {
    if (!c || which >= VIO_K_COUNT) return VIO_ERR_BAD_ARG;
    c->prof_which = which;
    c->prof_used = 0;
    c->prof_every = 1; c->prof_seen = 0;
    return VIO_OK;
}

vio_status vio_profile_end(vio_ctx *c, double *total_ms, int64_t *launches) {
    if (!c) return VIO_ERR_BAD_ARG;
    enter_device(c, true);
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0;
    const size_t used = c->prof_used;
    c->prof_which = -1;             // (the session ends whatever the events say)
    c->prof_used = 0;
    for (size_t i = 0; i + 1 < used; i += 2) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_events[i], c->prof_events[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)(used / 2);
    return VIO_OK;
}

}  // extern "C"
