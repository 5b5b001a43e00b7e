// vio_kernels.hip — hand-written gfx950 kernels of the sliding-window VIO backend.
//
// One Levenberg-Marquardt trial of the reference (Problem::Solve, VM/src/backend/problem.cc:169-250) maps to
//
//   k_linearize   MakeHessian (problem.cc:303-389) for all reprojection edges (edge_reprojection.cc:18-109,
//                 edge.cc:48-74) + the per-landmark Schur terms of SolveLinearSystem (problem.cc:412-429),
//                 one workgroup of 1024 threads per item (48..96 landmarks sharing a (host, targets) pattern), the
//                 item's blocks formed as fp64 MFMA tiles; IMU edges (edge_imu.cc:13-156,
//                 integration_base.h:160-186) ride in the same grid, and in GN mode so does the previous step's test
//   k_reduce      fixed-order sum of the per-item partial blocks -> 72x72 reduced visual system
//   k_assemble    + IMU blocks + prior (problem.cc:365-384) -> H_pp_schur_, b_pp_schur_ (171), in Eigen's pivot
//                 order, as 16x16 tiles
//   k_pose_solve  + lambda (problem.cc:434-436), LDLT in that order (Eigen LDLT, problem.cc:439) blocked on fp64
//                 MFMA tiles with look-ahead, pose update (UpdateStates :453-480, vertex_pose.cc:7-19), prior
//                 first-order update (:473-474)
//   k_backsub     landmark back-substitution (problem.cc:445), landmark update, chi2 of the trial state
//                 (IsGoodStepInLM :549-556)
//   k_lm_decide   gain ratio + Nielsen update + accept/rollback (IsGoodStepInLM :541-573, Solve :188-245)
// and, either side of the solve:
//   k_triangulate FeatureManager::triangulate (feature_manager.cpp:203-257), one thread per track
//
// All arithmetic is fp64.  Every reduction has a fixed order (no float atomics), so results are bitwise
// reproducible run to run.  wave = 64 lanes.
#include <hip/hip_runtime.h>

#include "vio_device_math.h"
#include "vio_types.h"

// threads of a k_linearize / k_linearize_xyz workgroup.  One workgroup holds a CU (its LDS), so the register budget of a thread is
// 512 / (LIN_THREADS / 256): 128 at 1024 threads (16 waves, 4 per SIMD), 170 at 768 (12 waves, 3 per SIMD).
// (LIN_THREADS: vio_types.h)
#ifdef LIN_MIN_WAVES            // (experiments with smaller workgroups, several to a CU: waves per SIMD the register allocation must leave room for)
#define LIN_BOUNDS __launch_bounds__(LIN_THREADS, LIN_MIN_WAVES)
#else
#define LIN_BOUNDS __launch_bounds__(LIN_THREADS)
#endif

// LIN_EARLY_LW: k_linearize stores the landmarks' Schur rows right behind phase 1.5 instead of at its end (see there): 0 (default) in the
// 1024-thread kernels only, 1 everywhere, -1 nowhere
#ifndef LIN_EARLY_LW
#define LIN_EARLY_LW 0
#endif
// a workgroup barrier that orders LDS accesses only: outstanding global stores are not waited for (__syncthreads() carries a fence that is)
__device__ __forceinline__ void d_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Diagnostic builds for the per-phase attribution of the counters (tools/profile_phases.sh): -DLIN_EXIT_AFTER=p makes every item workgroup
// of k_linearize* leave behind phase p (0 head, 1 phase 1, 2 phase 1.5, 3 phase 2); the differences between consecutive builds are the phases
#ifdef LIN_EXIT_AFTER
#define LIN_EXIT(p) do { if (LIN_EXIT_AFTER == (p)) return; } while (0)
#else
#define LIN_EXIT(p) do { } while (0)
#endif
// In-kernel stamps exist only in the diagnostic build (-DVIO_STAMPS -> libvio_hip_stamps.so, never shipped or timed)
#ifdef VIO_STAMPS
// Stamps go to LDS and are flushed once at the end: a global store in front of a barrier would add its own round
// trip to the phase it is meant to time.
// Slots 8/9: s_memrealtime (the 100 MHz reference clock, one clock domain for the whole device) at the first and at the
// last stamp; slot 10: XCC_ID, slot 11: HW_ID — where and when a workgroup ran relative to the others.
__shared__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned d_hwreg_xcc() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x; }
__device__ __forceinline__ unsigned d_hwreg_hwid() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x)); return x; }
#define STAMP(T, slot) do { if (threadIdx.x == 0) { g_stamps[slot] = __builtin_amdgcn_s_memtime(); \
        if ((slot) == 0) { g_stamps[8] = __builtin_amdgcn_s_memrealtime(); g_stamps[10] = d_hwreg_xcc(); g_stamps[11] = d_hwreg_hwid(); } \
        if ((slot) == 5) g_stamps[9] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define STAMP_FLUSH(T) do { if (threadIdx.x == 0 && (T).dbg) for (int q__ = 0; q__ < 16; ++q__) (T).dbg[(size_t)blockIdx.x * 16 + q__] = g_stamps[q__]; } while (0)
#else
#define STAMP(T, slot) do { } while (0)
#define STAMP_FLUSH(T) do { } while (0)
#endif

// which copy of the double-buffered state is current: the host's hint when it tracks it (GN mode), else LmState
__device__ __forceinline__ int d_cur(const DeviceTables &T) {
    if (T.cur_hint >= 0) return T.cur_hint;
    return T.cur_hint == -2 ? (T.lm->cur ^ T.lm->pending) : T.lm->cur;
}
// vio_solve's loop keeps two sets of what a linearisation leaves for later launches (the landmark rows lw, the permuted system
// Pg / perm, b_ of the pose block): the set a launch reads — the system the pending step came from — and the one it writes
__device__ __forceinline__ int d_set_r(const DeviceTables &T) { return T.cur_hint == -2 ? T.lm->sys : 0; }
__device__ __forceinline__ int d_set_w(const DeviceTables &T) { return T.cur_hint == -2 ? (T.lm->sys ^ T.lm->pending) : 0; }
#define PS_SET_STRIDE (66 * 272 + 192)      // PS_PACKED (defined with k_pose_solve)
// bit 0 / bit 1 of gn_flags ("the previous step waits for its test / its landmark back-substitution"): in vio_solve's loop
// (cur_hint -2) the host sets them on every slot and the device knows whether there is such a step
__device__ __forceinline__ bool d_step_owed(const DeviceTables &T, int bit) {
    return (T.gn_flags & bit) && (T.cur_hint != -2 || T.lm->pending != 0);
}

// Device-driven LM loop (vio_solve): the host enqueues whole iterations ahead; a kernel whose turn has not come
// (the loop has stopped, or the last trial was rejected and there is nothing to re-linearise) returns at once.
__device__ __forceinline__ bool d_gated_off(const LmState *lm, int gate) {
    if (gate == 0) return false;
    if (lm->stop) return true;
    return (gate & 1) && !lm->need_linearize;
}

__device__ __forceinline__ int cam_to_full(int c) { return c < 6 ? c : 6 + 15 * ((c - 6) / 6) + (c - 6) % 6; }
// inverse: -1 when the full index is a speed-bias dimension
__device__ __forceinline__ int full_to_cam(int i) {
    if (i < 6) return i;
    const int f = (i - 6) / 15, o = (i - 6) % 15;
    return o < 6 ? 6 + 6 * f + o : -1;
}

// Entry idx of the reduced visual system summed over the shards: unsharded, the window's own sums (k_reduce); sharded, the
// all-gathered slabs of all ranks added in RANK ORDER — the same additions in the same order on every rank, so every rank
// holds the identical bits whatever algorithm the collective library chose ("fixed reduction order").
// The loads of the ranks do not depend on each other: one round trip, as for the single read.
__device__ __forceinline__ double d_vis(const DeviceTables &T, int idx) {
    if (T.n_shards == 0) return T.vis[idx];
    const double *g = T.gath + idx;
    double s = g[0];
#pragma unroll 8
    for (int r = 1; r < T.n_shards; ++r) s += g[(size_t)r * VIS_SEND];
    return s;
}
__device__ __forceinline__ double d_vis_maxh(const DeviceTables &T) {       // max |h_ll| over all shards
    if (T.n_shards == 0) return T.vis[VIS_MAXH];
    double m = T.gath[VIS_MAXH];
    for (int r = 1; r < T.n_shards; ++r) m = fmax(m, T.gath[(size_t)r * VIS_SEND + VIS_MAXH]);
    return m;
}
__device__ __forceinline__ double d_step_tot(const DeviceTables &T, int i) {    // the two step scalars of the stepwise path, the same way
    if (T.n_shards == 0) return T.step_tot[i];
    double s = T.step_gath[i];
    for (int r = 1; r < T.n_shards; ++r) s += T.step_gath[2 * r + i];
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// pair table: for every ordered frame pair (h,t) the composed maps of the reprojection chain
//   p_cj = C * p_ci + d   with C = ric^T Rt^T Rh ric,  d = ric^T (Rt^T (Rh tic + Ph - Pt) - tic)
// (edge_reprojection.cc:35-40 evaluated once per pair instead of once per edge)
// ---------------------------------------------------------------------------------------------------------
// The table is built in two steps: the 12 rotation matrices (extrinsic, 11 poses) into sR, then — after a barrier — one task per
// (pair, row): row r of A, B, C, El and d[r].  k_pose_solve spreads the 330 tasks over its 1024 threads after the last barrier of
// its tail (one thread per pair on two waves before: 64.24 -> 63.87 us per GN iteration, tools/ab_gn_timing.py); k_prepare loops
// over them with 128.  Every entry is the same sum whoever forms it.
__device__ __forceinline__ void d_pair_rotations(const double *st, double *sR, int tid) {
    if (tid < 12) {
        const double *q = (tid == 0) ? st + STATE_EXT + 3 : st + STATE_POSE + 7 * (tid - 1) + 3;
        d_quat_to_R(q, sR + 9 * tid);
    }
}
__device__ __forceinline__ void d_pair_rows(const double *st, double *tab, const double *sR, int tid, int nt) {
    const double *ric = sR;
    const double *tic = st + STATE_EXT;
    // one task per (part, pair, row): part 0 = row r of A, B, C and d[r], part 1 = row r of El (its 27-product R_t^T R_h is as long as
    // the rest of the row together: as one task of 363 a row kept six waves of k_pose_solve_c's tail busy for 2.9 k ticks while eight idled)
    for (int task = tid; task < 2 * 121 * 3; task += nt) {
        const int part = task >= 121 * 3;
        const int tk = task - part * 121 * 3;
        const int pr = tk / 3, r = tk - 3 * pr;
        const int h = pr / 11, t = pr % 11;
        if (h == t) continue;
        const double *Rh = sR + 9 * (1 + h), *Rt = sR + 9 * (1 + t);
        const double c0 = ric[r], c1 = ric[3 + r], c2 = ric[6 + r];          // column r of ric
        double *o = tab + pr * PAIR_STRIDE;
        if (part == 0) {
            const double *Ph = st + STATE_POSE + 7 * h, *Pt = st + STATE_POSE + 7 * t;
            double Ar[3], Br[3], Cr[3], u[3], w[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) Ar[j] = Rt[3 * j] * c0 + Rt[3 * j + 1] * c1 + Rt[3 * j + 2] * c2;      // A = ric^T Rt^T = (Rt ric)^T
#pragma unroll
            for (int j = 0; j < 3; ++j) Br[j] = Ar[0] * Rh[j] + Ar[1] * Rh[3 + j] + Ar[2] * Rh[6 + j];         // B = A Rh
#pragma unroll
            for (int j = 0; j < 3; ++j) Cr[j] = Br[0] * ric[j] + Br[1] * ric[3 + j] + Br[2] * ric[6 + j];      // C = B ric
            d_m3_vec(Rh, tic, u);
#pragma unroll
            for (int k = 0; k < 3; ++k) u[k] = u[k] + Ph[k] - Pt[k];
            d_m3_tvec(Rt, u, w);
#pragma unroll
            for (int k = 0; k < 3; ++k) w[k] -= tic[k];
            const double dr = c0 * w[0] + c1 * w[1] + c2 * w[2];                                               // d = ric^T (Rt^T (Rh tic + Ph - Pt) - tic)
#pragma unroll
            for (int j = 0; j < 3; ++j) { o[PAIR_A + 3 * r + j] = Ar[j]; o[PAIR_B + 3 * r + j] = Br[j]; o[PAIR_C + 3 * r + j] = Cr[j]; }
            o[PAIR_D + r] = dr;
        } else {
            double Elr[3], RtRh[9];
            d_m3_tmul(Rt, Rh, RtRh);
            RtRh[0] -= 1; RtRh[4] -= 1; RtRh[8] -= 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) Elr[j] = c0 * RtRh[j] + c1 * RtRh[3 + j] + c2 * RtRh[6 + j];           // El = ric^T (Rt^T Rh - I)
#pragma unroll
            for (int j = 0; j < 3; ++j) o[PAIR_EL + 3 * r + j] = Elr[j];
        }
    }
    if (tid < 9) tab[121 * PAIR_STRIDE + CAMTAB_RIC + tid] = ric[tid];
    if (tid < 3) tab[121 * PAIR_STRIDE + CAMTAB_TIC + tid] = tic[tid];
}
__device__ void d_build_pairtab(const double *st, double *tab, double *sR, int tid, int nt) {
    d_pair_rotations(st, sR, tid);
    __syncthreads();
    d_pair_rows(st, tab, sR, tid, nt);
}

__global__ __launch_bounds__(128) void k_prepare(DeviceTables T) {
    __shared__ double sR[12 * 9];
    const int cur = d_cur(T);
    d_build_pairtab(T.state + cur * STATE_STRIDE, T.pairtab + cur * PAIRTAB_STRIDE, sR, threadIdx.x, blockDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// IMU factor (one workgroup per edge, appended to the linearize grid)
// ---------------------------------------------------------------------------------------------------------
#define O_P 0
#define O_R 3
#define O_V 6
#define O_BA 9
#define O_BG 12

// The IMU factor's residual and Jacobian blocks are evaluated with NO floating-point contraction: every product and every sum is the
// IEEE operation the source states, in the order it states it (as the oracle's C is compiled).  With the compiler free to fuse a*b+c
// the result of these expressions depended on what surrounded them after inlining (a product with a second use is not fused), and two
// call sites of the same function — the IMU workgroups of d_imu_item, the chain workgroup of the GN loop — could differ in the last bit.
// The helpers of vio_device_math.h they use are restated here under the same rule.
#pragma clang fp contract(off)
__device__ __forceinline__ void nc_quat_to_R(const double *q, double *R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void nc_m3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ dquat nc_qmul(dquat a, dquat b) {       // Eigen quaternion product
    dquat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
__device__ __forceinline__ dquat nc_qinv(dquat q) {                // Eigen::QuaternionBase::inverse
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    dquat r = {0, 0, 0, 0};
    if (n2 > 0) { r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; r.w = q.w / n2; }
    return r;
}
__device__ __forceinline__ void nc_qrot(dquat q, const double *v, double *o) {   // Eigen _transformVector
    double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
struct ImuCommon {
    dquat Qi, Qj, Qi_inv, dq, cdq;
    double sum_dt;
    double dba[3], dbg[3];
};

__device__ __forceinline__ void d_skew(const double *v, double *S) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
__device__ __forceinline__ void d_qleft_br(dquat q, double *B) {      // Utility::Qleft bottom-right 3x3 (utility.h:48-56)
    double v[3] = {q.x, q.y, q.z};
    d_skew(v, B);
    B[0] += q.w; B[4] += q.w; B[8] += q.w;
}
__device__ __forceinline__ void d_qright_br(dquat q, double *B) {     // Utility::Qright bottom-right 3x3 (utility.h:58-66)
    double v[3] = {q.x, q.y, q.z}, S[9];
    d_skew(v, S);
#pragma unroll
    for (int k = 0; k < 9; ++k) B[k] = -S[k];
    B[0] += q.w; B[4] += q.w; B[8] += q.w;
}

__device__ void d_imu_common(const double *pre, const double *pi, const double *si, const double *pj, ImuCommon &c) {
    c.Qi = d_qload(pi); c.Qj = d_qload(pj);
    c.Qi_inv = nc_qinv(c.Qi);
    c.sum_dt = pre[PRE_SUMDT];
    c.dq.x = pre[PRE_DQ]; c.dq.y = pre[PRE_DQ + 1]; c.dq.z = pre[PRE_DQ + 2]; c.dq.w = pre[PRE_DQ + 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.dba[k] = si[3 + k] - pre[PRE_BA + k]; c.dbg[k] = si[6 + k] - pre[PRE_BG + k]; }
    const double *Jm = pre + PRE_JAC;
    double th[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        th[i] = Jm[15 * (O_R + i) + O_BG] * c.dbg[0] + Jm[15 * (O_R + i) + O_BG + 1] * c.dbg[1] + Jm[15 * (O_R + i) + O_BG + 2] * c.dbg[2];
    dquat dth = {th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0};      // Utility::deltaQ, not normalised
    c.cdq = nc_qmul(c.dq, dth);
}

// IntegrationBase::evaluate (integration_base.h:160-186)
__device__ void d_imu_residual(const double *pre, const double *G, const double *pi, const double *si, const double *pj,
                               const double *sj, const ImuCommon &c, double *res) {
    const double *Jm = pre + PRE_JAC;
    const double sum_dt = c.sum_dt;
    double cdp[3], cdv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a = 0, b = 0, e = 0, f = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a += Jm[15 * (O_V + i) + O_BA + j] * c.dba[j];
            b += Jm[15 * (O_V + i) + O_BG + j] * c.dbg[j];
            e += Jm[15 * (O_P + i) + O_BA + j] * c.dba[j];
            f += Jm[15 * (O_P + i) + O_BG + j] * c.dbg[j];
        }
        cdv[i] = pre[PRE_DV + i] + a + b;
        cdp[i] = pre[PRE_DP + i] + e + f;
    }
    double t[3], u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = 0.5 * G[k] * sum_dt * sum_dt + pj[k] - pi[k] - si[k] * sum_dt;
    nc_qrot(c.Qi_inv, t, u);
#pragma unroll
    for (int k = 0; k < 3; ++k) res[O_P + k] = u[k] - cdp[k];
    dquat qe = nc_qmul(nc_qinv(c.cdq), nc_qmul(c.Qi_inv, c.Qj));
    res[O_R] = 2 * qe.x; res[O_R + 1] = 2 * qe.y; res[O_R + 2] = 2 * qe.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = G[k] * sum_dt + sj[k] - si[k];
    nc_qrot(c.Qi_inv, t, u);
#pragma unroll
    for (int k = 0; k < 3; ++k) res[O_V + k] = u[k] - cdv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { res[O_BA + k] = sj[3 + k] - si[3 + k]; res[O_BG + k] = sj[6 + k] - si[6 + k]; }
}

// One 3x3 block of the 15x30 Jacobian [J_pose_i | J_sb_i | J_pose_j | J_sb_j] (edge_imu.cc:74-153).
// `blk` enumerates the 14 non-zero blocks; sJ is the 15x30 row-major LDS image (zero-initialised).
// RiT = Qi.inverse().toRotationMatrix() (d_imu_rit)
__device__ __forceinline__ void d_imu_rit(const ImuCommon &c, double *RiT) {
    double qi[4] = {c.Qi_inv.x, c.Qi_inv.y, c.Qi_inv.z, c.Qi_inv.w};
    nc_quat_to_R(qi, RiT);
}
__device__ void d_imu_jac_block(int blk, const double *pre, const double *G, const double *pi, const double *si,
                                const double *pj, const double *sj, const ImuCommon &c, const double *RiT, double *sJ) {
    const double *Jm = pre + PRE_JAC;
    const double sum_dt = c.sum_dt;
    double B[9];
    int r0 = 0, c0 = 0;
    switch (blk) {
    case 0: r0 = O_P; c0 = 0 + O_P;       // jacobian_pose_i(O_P,O_P) = -Ri^T
        for (int k = 0; k < 9; ++k) B[k] = -RiT[k];
        break;
    case 1: { r0 = O_P; c0 = 0 + O_R;     // skew(Qi^-1 (0.5 G dt^2 + Pj - Pi - Vi dt))
        double t[3], u[3];
        for (int k = 0; k < 3; ++k) t[k] = 0.5 * G[k] * sum_dt * sum_dt + pj[k] - pi[k] - si[k] * sum_dt;
        nc_qrot(c.Qi_inv, t, u); d_skew(u, B);
        break; }
    case 2: { r0 = O_R; c0 = 0 + O_R;     // -(Qleft(Qj^-1 Qi) Qright(corrected_delta_q)).bottomRight
        dquat a = nc_qmul(nc_qinv(c.Qj), c.Qi), b = c.cdq;
        double La[9], Rb[9], P[9];
        d_qleft_br(a, La); d_qright_br(b, Rb); nc_m3_mul(La, Rb, P);
        const double va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = -(va[i] * (-vb[j]) + P[3 * i + j]);
        break; }
    case 3: { r0 = O_V; c0 = 0 + O_R;     // skew(Qi^-1 (G dt + Vj - Vi))
        double t[3], u[3];
        for (int k = 0; k < 3; ++k) t[k] = G[k] * sum_dt + sj[k] - si[k];
        nc_qrot(c.Qi_inv, t, u); d_skew(u, B);
        break; }
    case 4: r0 = O_P; c0 = 6 + 0;         // speedbias_i(O_P, V) = -Ri^T dt
        for (int k = 0; k < 9; ++k) B[k] = -RiT[k] * sum_dt;
        break;
    case 5: r0 = O_P; c0 = 6 + 3;         // -dp_dba
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = -Jm[15 * (O_P + i) + O_BA + j];
        break;
    case 6: r0 = O_P; c0 = 6 + 6;         // -dp_dbg
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = -Jm[15 * (O_P + i) + O_BG + j];
        break;
    case 7: { r0 = O_R; c0 = 6 + 6;       // -Qleft(Qj^-1 Qi delta_q).bottomRight * dq_dbg  (delta_q, not corrected: edge_imu.cc:107-109)
        double L[9], nL[9], D[9];
        d_qleft_br(nc_qmul(nc_qmul(nc_qinv(c.Qj), c.Qi), c.dq), L);
        for (int k = 0; k < 9; ++k) nL[k] = -L[k];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[3 * i + j] = Jm[15 * (O_R + i) + O_BG + j];
        nc_m3_mul(nL, D, B);
        break; }
    case 8: r0 = O_V; c0 = 6 + 0;         // -Ri^T
        for (int k = 0; k < 9; ++k) B[k] = -RiT[k];
        break;
    case 9: r0 = O_V; c0 = 6 + 3;         // -dv_dba
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = -Jm[15 * (O_V + i) + O_BA + j];
        break;
    case 10: r0 = O_V; c0 = 6 + 6;        // -dv_dbg
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = -Jm[15 * (O_V + i) + O_BG + j];
        break;
    case 11: r0 = O_P; c0 = 15 + O_P;     // pose_j(O_P,O_P) = Ri^T
        for (int k = 0; k < 9; ++k) B[k] = RiT[k];
        break;
    case 12: { r0 = O_R; c0 = 15 + O_R;   // Qleft(corrected_dq^-1 Qi^-1 Qj).bottomRight
        d_qleft_br(nc_qmul(nc_qmul(nc_qinv(c.cdq), c.Qi_inv), c.Qj), B);
        break; }
    case 13: r0 = O_V; c0 = 21 + 0;       // speedbias_j(O_V,V) = Ri^T
        for (int k = 0; k < 9; ++k) B[k] = RiT[k];
        break;
    default: return;
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sJ[30 * (r0 + i) + c0 + j] = B[3 * i + j];
}

#pragma clang fp contract(fast)
// T = J^T Info J (30x30), g = J^T Info r, chi = r^T Info r  for IMU edge k
// (Round 6 tried the inputs staged in LDS and the sixteen tasks one per wave on uniform branches: bit-identical, the IMU workgroup 19-20 k ->
//  16.4 k cycles — and k_linearize at 20 000 landmarks 13.5 -> 14.2 us, the GN iteration 45.7 -> 46.4 us (tools/ab.py, profiles/r06n_imu_item_ab.txt):
//  the item workgroups, which share the kernel's register allocation, lost more than the IMU workgroups — never the long pole — gained.  Reverted.)
template <int NT> __device__ void d_imu_item(const DeviceTables &T, int k, double *smem) {
    const int tid = threadIdx.x;
    double *sJ = smem;               // 450
    double *sI = sJ + 450;           // 225 information
    double *sJtI = sI + 225;         // 30x15
    double *sr = sJtI + 450;         // 15
    double *sIr = sr + 16;           // 15
    double *out = T.imu_out + k * IMU_OUT;
    if (!T.imu_valid[k]) {
        for (int e = tid; e < IMU_OUT; e += NT) out[e] = 0.0;
        return;
    }
    const int cur = d_cur(T);
    const double *st = T.state + cur * STATE_STRIDE;
    const double *pre = T.pre + k * PRE_STRIDE;
    const double *pi = st + STATE_POSE + 7 * k, *pj = pi + 7, *si = st + STATE_SB + 9 * k, *sj = si + 9;
    for (int e = tid; e < 450; e += NT) sJ[e] = 0.0;
    for (int e = tid; e < 225; e += NT) sI[e] = pre[PRE_INFO + e];
    __syncthreads();
    if (tid < 16) {
        ImuCommon c;
        double RiT[9];
        d_imu_common(pre, pi, si, pj, c);
        d_imu_rit(c, RiT);
        if (tid < 14) d_imu_jac_block(tid, pre, T.gravity, pi, si, pj, sj, c, RiT, sJ);
        else if (tid == 14) {
            double r[15];
            d_imu_residual(pre, T.gravity, pi, si, pj, sj, c, r);
            for (int i = 0; i < 15; ++i) sr[i] = r[i];
        } else {
            for (int i = 0; i < 3; ++i) {       // identity blocks (edge_imu.cc:116-118,146-148)
                sJ[30 * (O_BA + i) + 6 + 3 + i] = -1.0; sJ[30 * (O_BG + i) + 6 + 6 + i] = -1.0;
                sJ[30 * (O_BA + i) + 21 + 3 + i] = 1.0; sJ[30 * (O_BG + i) + 21 + 6 + i] = 1.0;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 450; e += NT) {      // JtI[a][j] = sum_i J[i][a] * I[i][j]
        const int a = e / 15, j = e % 15;
        double s = 0;
        for (int i = 0; i < 15; ++i) s = fma(sJ[30 * i + a], sI[15 * i + j], s);
        sJtI[e] = s;
    }
    if (tid < 15) {
        double s = 0;
        for (int j = 0; j < 15; ++j) s = fma(sI[15 * tid + j], sr[j], s);
        sIr[tid] = s;
    }
    __syncthreads();
    for (int e = tid; e < 900; e += NT) {
        const int a = e / 30, b = e % 30;
        double s = 0;
        for (int j = 0; j < 15; ++j) s = fma(sJtI[15 * a + j], sJ[30 * j + b], s);
        out[IMU_T + e] = s;
    }
    if (tid < 30) {
        double s = 0;
        for (int i = 0; i < 15; ++i) s = fma(sJ[30 * i + tid], sIr[i], s);
        out[IMU_G + tid] = s;
    }
    if (tid == 32) {
        double s = 0;
        for (int i = 0; i < 15; ++i) s = fma(sr[i], sIr[i], s);
        out[IMU_CHI] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_linearize: one workgroup (1024 threads) per item
//   phase 1   thread per observation (k-major: a wave shares one target frame): residual, Jacobians, robust
//             weight; the whitened rows L*J go to sRows (one plane per k), the thread's own terms of the
//             per-landmark sums go to sAux (host/extrinsic blocks) or straight to the landmark record (target block)
//   phase 1.5 thread per (landmark, quantity): sums the K per-observation terms: h_ll, b_l, Schur row w = Hpl
//   phase 2   one wave per 16x16 product on the matrix cores (v_mfma_f64_16x16x4_f64): C_k = V_k^T V_k for the
//             whitened rows of every observation index k, and the Schur term - sum_g w_g w_g^T / h_g; the b vectors
//             ride in the products' padding: the pose part of b as one more column z of V_k (L z = drho Info r), its
//             Schur correction as one more row b_l of w; fixed summation order
//   combine   thread per slab element picks its direct and Schur entries out of the tiles; coalesced store to the
//             slab; w/h/b_l go to HBM last (the back-substitution reads them)
// ---------------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
typedef double ps_v4d __attribute__((ext_vector_type(4)));      // accumulator of v_mfma_f64_16x16x4_f64

// (lin_rrow .. lin_lds_doubles, the LDS layout of an item: vio_types.h — the host-only planner sizes items with the same formulas)
// b_prior'[i] = b_prior[i] - (H_prior dx)[i]  (problem.cc:473), one wave per row; the same sum whoever calls it
__device__ __forceinline__ double d_bprior_dot(double h0, double h1, double h2, double x0, double x1, double x2, double b) {
    double s = 0;
    s += h0 * x0;
    s += h1 * x1;
    s += h2 * x2;
    s = d_wave_sum_to_lane63(s);
    return b - s;
}
// rows first, first + step, ... of the update above, from HBM (the GN loop's deferred form: k_linearize, k_backsub)
__device__ __forceinline__ void d_bprior_rows(const DeviceTables &T, int from, int to, int first, int step, int lane) {
    const bool in2 = lane + 128 < VIO_PD;
    const double x0 = T.dx[lane], x1 = T.dx[lane + 64], x2 = in2 ? T.dx[lane + 128] : 0.0;
    for (int i = first; i < VIO_PD; i += step) {
        const double *h = T.Hprior + (size_t)i * VIO_PD;
        const double v = d_bprior_dot(h[lane], h[lane + 64], in2 ? h[lane + 128] : 0.0, x0, x1, x2, T.bprior[from * 176 + i]);
        if (lane == 63) T.bprior[to * 176 + i] = v;
    }
}

// the chain workgroup of the GN loop's grid (vio_pose_solve_chain.h)
__device__ __forceinline__ void d_chain_pre_item(const DeviceTables &T);
// NT: threads of the workgroup.  UE = 0: the plan has no extrinsic block (vio_config.ext_fixed, the reference's ESTIMATE_EXTRINSIC = 0) —
// row-record sizes and operand-stream strides are compile-time constants (phase 2: 7.3 k -> 6.8 k cycles); UE = 1: read from the item.
// CH = 1 (the 1024-thread single-window kernels): with gn_flags bit 4 the grid's first workgroup is the chain workgroup, the items follow.
template <int NT, int UE, int CH = 0> __device__ __forceinline__ void d_linearize_body(const DeviceTables &T) {
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    if (CH && (T.gn_flags & 16)) {
        if (b == 0) { d_chain_pre_item(T); return; }
        b -= 1;
    }
    // the item's descriptor is requested before the gate looks at LmState (vio_solve's loop): one round trip for both, not two
    int32_t desc_word = 0;
    if (b < T.n_items && tid < (int)(sizeof(ItemDesc) / 4)) desc_word = ((const int32_t *)(T.items + b))[tid];
    if (d_gated_off(T.lm, T.lm_gate)) return;
    // GN mode (gn_flags bit 1) with a prior: the previous step also owes b_prior' = b_prior - H_prior dx (problem.cc:473).
    // Row r belongs to workgroup r mod grid, to its last wave: idle in phase 1 of a landmark item, first thing in an IMU item.
    const bool owe = d_step_owed(T, 2);
    const bool owe_prior = owe && T.has_prior;
    if (b >= T.n_items) {
        STAMP(T, 0);
        if (owe_prior && (tid >> 6) == NT / 64 - 1) { const int c = d_cur(T); d_bprior_rows(T, c ^ 1, c, b, T.n_step_blocks, tid & 63); }
#ifndef LIN_DIAG_NO_IMU
        d_imu_item<NT>(T, b - T.n_items, dyn_smem);
#endif
        STAMP(T, 5);
        STAMP_FLUSH(T);
        return;
    }
    __shared__ ItemDesc sIt;        // kept in LDS: its small arrays are indexed at run time
    const int cur = d_cur(T);      // requested together with the descriptor: both are cold after the kernel boundary
    const int64_t lw_r = d_set_r(T) * T.lw_set, lw_w = d_set_w(T) * T.lw_set;
    if (tid < (int)(sizeof(ItemDesc) / 4)) ((int32_t *)&sIt)[tid] = desc_word;
    __syncthreads();
    const ItemDesc &it = sIt;
    STAMP(T, 0);
    // (the descriptor comes out of LDS into vector registers; these are the same in every lane: scalar registers, scalar loop
    // control and address arithmetic from here on)
    const int G = __builtin_amdgcn_readfirstlane(it.G), K = __builtin_amdgcn_readfirstlane(it.K), nb = __builtin_amdgcn_readfirstlane(it.nb),
              use_ext = UE == 0 ? 0 : __builtin_amdgcn_readfirstlane(it.use_ext);
    const int RROW = lin_rrow(use_ext), RAUX = lin_raux(use_ext), PLANE = lin_plane(G, use_ext), LREC = lin_lrec(nb);
    const int offH = 0, offT = 12, offE = 24, offZ = RROW - 2;      // inside a row record; (z0, z1) at its end
    // per-observation partials of the landmark quantities (sAux record): host w (6), h, b_l, [ext w (6)]
    const int pkWH = 0, pkH = 6, pkBL = 7, pkWE = 8;
    // inside a landmark record
    const int lHinv = 6 * nb, lBl = 6 * nb + 1, lH = 6 * nb + 2, lLam = 6 * nb + 3, lSc = 6 * nb + 4, lDl = 6 * nb + 5;

    double *sPair = dyn_smem;                         // VIO_MAXK * PAIR_STRIDE
    double *sCam = sPair + VIO_MAXK * PAIR_STRIDE;    // ric, tic
    double *sZero = sCam + 12;                        // a zero for the padding lanes of phase 2 (sCam holds 12 values)
    double *sRed = sCam + 16;                         // 3 * waves
    double *sRows = sRed + 3 * (LIN_THREADS / 64);    // K * PLANE
    double *sL = sRows + K * PLANE;                   // G * LREC
    double *sAux = sL + G * LREC;                     // G*K*RAUX, dead after phase 1.5 ...
    double *sTile = sAux;                             // ... then the 16x16 tiles of phase 2 and the vector partials

    const double *invd = T.invd + (size_t)cur * T.Ns + it.lm_base;
    const double *pts_i = T.pts_i + 2 * (size_t)it.lm_base;
    const double *pts_j = T.pts_j + 2 * (size_t)it.obs_base;
    // the first observation's inputs and the table words of the later phases are requested before the pair table's
    // barrier: every global latency of the workgroup overlaps this one
    double pf_lam = 1.0, pf_x = 0.0, pf_y = 0.0, pf_u = 0.0, pf_v = 0.0;
    if (tid < G * K) {
        const int k0 = tid / G, g0 = tid - k0 * G;
        if (!owe) pf_lam = invd[g0];
        pf_x = pts_i[2 * g0]; pf_y = pts_i[2 * g0 + 1]; pf_u = pts_j[2 * tid]; pf_v = pts_j[2 * tid + 1];
    }
    // the pair table entries of this item, requested now, stored after the landmark update below has used their place
    const double *ptab = T.pairtab + cur * PAIRTAB_STRIDE;
    double pv = 0.0, cv = 0.0;
    if (tid < K * PAIR_STRIDE) {
        const int k = tid / PAIR_STRIDE, o = tid % PAIR_STRIDE;
        pv = ptab[(it.host * 11 + it.target[k]) * PAIR_STRIDE + o];
    }
    if (tid < 12) cv = ptab[121 * PAIR_STRIDE + tid];
    // GN mode (gn_flags bit 1): the landmarks still owe the back-substitution of the PREVIOUS step (problem.cc:445):
    // delta_lambda = (b_l - w . dx_pose) / h from the rows this item's workgroup stored at the end of the previous
    // linearisation.  The new inverse depth goes to the current copy of invd and, through the landmark record, to
    // phase 1; the landmark part of the gain-ratio denominator goes to the record for the combine phase.
    // The item's rows ((6 nb + 2) x G doubles) and the pose update come in as one coalesced copy by the whole workgroup —
    // one round trip instead of one per pattern block — into LDS that phase 1 only needs later (sAux, sPair).
    if (owe) {
        const double *lw = T.lw + lw_r + it.lw_base;
        const int nlw = (6 * nb + 2) * G;                  // <= 7 * NT: nb <= 12, G <= 86
        double *sStage = sAux, *sDxS = sPair;
        double stv[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) { const int e = tid + q * NT; stv[q] = e < nlw ? lw[e] : 0.0; }
        const double dxv = tid < VIO_PD ? T.dx[tid] : 0.0;
        // (what the update itself reads from HBM is requested here too, not behind the barrier)
        const double inv_prev = tid < G ? T.invd[(size_t)(cur ^ 1) * T.Ns + it.lm_base + tid] : 0.0;
        const double lambda_lm = T.lm->lambda;
#pragma unroll
        for (int q = 0; q < 7; ++q) { const int e = tid + q * NT; if (e < nlw) sStage[e] = stv[q]; }
        if (tid < 176) sDxS[tid] = dxv;
        __syncthreads();
        if (tid < G) {
            const int g = tid;
            double t = 0.0;
            for (int p = 0; p < nb; ++p) {
                const int cb = it.cam_block[p];
                const int base = cb == 0 ? 0 : 6 + 15 * (cb - 1);
#pragma unroll
                for (int i = 0; i < 6; ++i) t += sStage[(6 * p + i) * G + g] * sDxS[base + i];
            }
            const double h = sStage[(6 * nb) * G + g], bl = sStage[(6 * nb + 1) * G + g];
            const double dl = (1.0 / h) * (bl - t);
            const double lam = inv_prev + dl;
            // dxl and the new inverse depth go out to HBM at the end of the kernel: a global store in front of a barrier
            // costs the store's whole round trip
            double *L = sL + (size_t)g * LREC;
            L[lLam] = lam;
            L[lSc] = dl * (lambda_lm * dl + bl);
            L[lDl] = dl;
        }
        __syncthreads();
    }
    if (tid < K * PAIR_STRIDE) sPair[tid] = pv;
    if (tid < 12) sCam[tid] = cv;
    if (tid >= 12 && tid < 16) sCam[tid] = 0.0;
    __syncthreads();

    const double s_info = T.sqrt_info, info = s_info * s_info;
    const double *ric = sCam, *tic = sCam + CAMTAB_TIC;

    // ---------------- phase 1 ----------------
    LIN_EXIT(0);
    STAMP(T, 1);
    if (owe_prior && (tid >> 6) == NT / 64 - 1) d_bprior_rows(T, cur ^ 1, cur, b, T.n_step_blocks, tid & 63);
    double chi_acc = 0.0;
    for (int o = tid; o < G * K; o += NT) {
        const int k = o / G, g = o - k * G;
        const double *PA = sPair + k * PAIR_STRIDE;
        const bool first = o == tid;
        const double lam = owe ? sL[(size_t)g * LREC + lLam] : (first ? pf_lam : invd[g]);
        const double il = 1.0 / lam;
        const double x = first ? pf_x : pts_i[2 * g], y = first ? pf_y : pts_i[2 * g + 1];
        const double u = first ? pf_u : pts_j[2 * o], v = first ? pf_v : pts_j[2 * o + 1];
        const double pci[3] = {x * il, y * il, il};
        double Cp[3], pcj[3];
        d_m3_vec(PA + PAIR_C, pci, Cp);
#pragma unroll
        for (int m = 0; m < 3; ++m) pcj[m] = Cp[m] + PA[PAIR_D + m];
        const double iz = 1.0 / pcj[2];
        const double r0 = pcj[0] * iz - u, r1 = pcj[1] * iz - v;
        const double ra = -pcj[0] * (iz * iz), rb = -pcj[1] * (iz * iz);     // reduce = [iz 0 ra; 0 iz rb]
        // J_lambda = reduce * C*pts_i * (-1/lam^2) = reduce * (C*pc_i) * (-1/lam)
        const double Jl0 = (iz * Cp[0] + ra * Cp[2]) * (-il), Jl1 = (iz * Cp[1] + rb * Cp[2]) * (-il);

        // robust weight: Edge::RobustInfo (edge.cc:48-74) with information = s^2 I
        const double e2 = r0 * (info * r0) + r1 * (info * r1);
        double rho0, rho1, rho2;
        d_loss(T.loss_type, T.loss_delta, e2, rho0, rho1, rho2);
        chi_acc += (T.loss_type == 0) ? e2 : rho0;
        double lam2 = rho1;
        if (T.loss_type != 0 && rho1 + 2 * rho2 * e2 > 0.) lam2 = rho1 + 2 * rho2 * e2;
        // symmetric square root L of W = s^2 (rho1 I + 2 rho2 (s r)(s r)^T): L = s (al I + (be-al) rhat rhat^T)
        const double al = sqrt(rho1), be = sqrt(fmax(lam2, 0.0));
        const double rn2 = r0 * r0 + r1 * r1;
        const double irn2 = rn2 > 0 ? 1.0 / rn2 : 0.0;
        const double gm = (be - al) * irn2;
        const double L00 = s_info * (al + gm * r0 * r0), L01 = s_info * (gm * r0 * r1), L11 = s_info * (al + gm * r1 * r1);
        const double c0 = rho1 * (info * r0), c1 = rho1 * (info * r1);         // drho * Information * residual
        const double a0 = L00 * Jl0 + L01 * Jl1, a1 = L01 * Jl0 + L11 * Jl1;      // whitened d r / d lambda

        // The landmark quantities of phase 1.5 are sums over the landmark's observations; this thread has its own
        // term of each in registers.  Target-block terms have one contributor: they go straight to the landmark
        // record; host (and extrinsic) terms go to the per-observation partials and are summed in phase 1.5.
        // The pose part of b, -drho J^T Info r (problem.cc:357), is not formed per block at all: with z = L^-1 (drho Info r) it is
        // -(L J)^T z, i.e. one more column of the product V_k^T V_k phase 2 forms anyway (the tile has 16 columns, V_k 12 or 18).
        // drho Info r is parallel to r, an eigenvector of L = s (al I + (be - al) r r^T / |r|^2) with eigenvalue s be:
        // z = drho Info r / (s be)   (be = 0 only where drho = 0: no contribution).
        double *rec = sRows + k * PLANE + g * RROW;
        double *pk = sAux + (size_t)o * RAUX;
        double *Lg = sL + (size_t)g * LREC;
        const int pT = it.tslot[k];
        {
            const double zs = be > 0.0 ? 1.0 / (s_info * be) : 0.0;
            rec[offZ] = c0 * zs; rec[offZ + 1] = c1 * zs;
        }
        pk[pkH] = a0 * a0 + a1 * a1;
        pk[pkBL] = Jl0 * c0 + Jl1 * c1;
        // The two 2x6 Jacobians one after the other, each whitened and stored before the next is formed: all four rows alive at once,
        // with what they are made of, is what pushed this loop over its 128 registers (65 spilled; now none on this path).
        // reduce*A: first three columns of J_pose_i, negated those of J_pose_j
        double RA0[3], RA1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            RA0[c] = iz * PA[PAIR_A + c] + ra * PA[PAIR_A + 6 + c];
            RA1[c] = iz * PA[PAIR_A + 3 + c] + rb * PA[PAIR_A + 6 + c];
        }
        {   // J_pose_i = [reduce*A | pb_i x (reduce*B)rows]   (row * hat(v) = row x v; -row x v = v x row)
            double pbi[3], RB0[3], RB1[3], Jh0[6], Jh1[6];
            d_m3_vec(ric, pci, pbi);
#pragma unroll
            for (int m = 0; m < 3; ++m) pbi[m] += tic[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                RB0[c] = iz * PA[PAIR_B + c] + ra * PA[PAIR_B + 6 + c];
                RB1[c] = iz * PA[PAIR_B + 3 + c] + rb * PA[PAIR_B + 6 + c];
                Jh0[c] = RA0[c]; Jh1[c] = RA1[c];
            }
            Jh0[3] = pbi[1] * RB0[2] - pbi[2] * RB0[1]; Jh0[4] = pbi[2] * RB0[0] - pbi[0] * RB0[2]; Jh0[5] = pbi[0] * RB0[1] - pbi[1] * RB0[0];
            Jh1[3] = pbi[1] * RB1[2] - pbi[2] * RB1[1]; Jh1[4] = pbi[2] * RB1[0] - pbi[0] * RB1[2]; Jh1[5] = pbi[0] * RB1[1] - pbi[1] * RB1[0];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double lh0 = L00 * Jh0[c] + L01 * Jh1[c], lh1 = L01 * Jh0[c] + L11 * Jh1[c];
                rec[offH + c] = lh0; rec[offH + 6 + c] = lh1;
                pk[pkWH + c] = lh0 * a0 + lh1 * a1;                 // Hpm column of this landmark, host block
            }
        }
        {   // J_pose_j = [-reduce*A | (reduce*ric^T)rows x pb_j]
            double pbj[3], RR0[3], RR1[3], Jt0[6], Jt1[6];
            d_m3_vec(ric, pcj, pbj);
#pragma unroll
            for (int m = 0; m < 3; ++m) pbj[m] += tic[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                RR0[c] = iz * ric[3 * c] + ra * ric[3 * c + 2];
                RR1[c] = iz * ric[3 * c + 1] + rb * ric[3 * c + 2];
                Jt0[c] = -RA0[c]; Jt1[c] = -RA1[c];
            }
            Jt0[3] = RR0[1] * pbj[2] - RR0[2] * pbj[1]; Jt0[4] = RR0[2] * pbj[0] - RR0[0] * pbj[2]; Jt0[5] = RR0[0] * pbj[1] - RR0[1] * pbj[0];
            Jt1[3] = RR1[1] * pbj[2] - RR1[2] * pbj[1]; Jt1[4] = RR1[2] * pbj[0] - RR1[0] * pbj[2]; Jt1[5] = RR1[0] * pbj[1] - RR1[1] * pbj[0];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double lt0 = L00 * Jt0[c] + L01 * Jt1[c], lt1 = L01 * Jt0[c] + L11 * Jt1[c];
                rec[offT + c] = lt0; rec[offT + 6 + c] = lt1;
                Lg[6 * pT + c] = lt0 * a0 + lt1 * a1;
            }
        }
        if (use_ext) {
            // J_ext = reduce * [El | -C hat(pc_i) + hat(C pc_i) + hat(d)] = [reduce*El | pc_i x (reduce*C)rows + red_rows x pc_j]
            double Je0[6], Je1[6], RC0[3], RC1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Je0[c] = iz * PA[PAIR_EL + c] + ra * PA[PAIR_EL + 6 + c];
                Je1[c] = iz * PA[PAIR_EL + 3 + c] + rb * PA[PAIR_EL + 6 + c];
                RC0[c] = iz * PA[PAIR_C + c] + ra * PA[PAIR_C + 6 + c];
                RC1[c] = iz * PA[PAIR_C + 3 + c] + rb * PA[PAIR_C + 6 + c];
            }
            // red0 = (iz,0,ra), red1 = (0,iz,rb);  red x pcj
            const double x0[3] = {0 * pcj[2] - ra * pcj[1], ra * pcj[0] - iz * pcj[2], iz * pcj[1] - 0 * pcj[0]};
            const double x1[3] = {iz * pcj[2] - rb * pcj[1], rb * pcj[0] - 0 * pcj[2], 0 * pcj[1] - iz * pcj[0]};
            Je0[3] = pci[1] * RC0[2] - pci[2] * RC0[1] + x0[0]; Je0[4] = pci[2] * RC0[0] - pci[0] * RC0[2] + x0[1]; Je0[5] = pci[0] * RC0[1] - pci[1] * RC0[0] + x0[2];
            Je1[3] = pci[1] * RC1[2] - pci[2] * RC1[1] + x1[0]; Je1[4] = pci[2] * RC1[0] - pci[0] * RC1[2] + x1[1]; Je1[5] = pci[0] * RC1[1] - pci[1] * RC1[0] + x1[2];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double le0 = L00 * Je0[c] + L01 * Je1[c], le1 = L01 * Je0[c] + L11 * Je1[c];
                rec[offE + c] = le0; rec[offE + 6 + c] = le1;
                pk[pkWE + c] = le0 * a0 + le1 * a1;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 1.5: thread per (landmark, quantity): sum the K partials ----------------
    LIN_EXIT(1);
    STAMP(T, 2);
    double maxh = 0.0;
    {
        // quantity q = tid / 128 (+ 8 per pass), landmark g = tid % 128 (G <= 128): a wave works on ONE quantity — the branch below
        // is uniform, only the two waves of h_ll pay for the division — and no index needs an integer division
        const int Q = use_ext ? 14 : 8;
        const int hs = __builtin_amdgcn_readfirstlane(it.host_slot);
        const int g = tid & 127;
        for (int q = __builtin_amdgcn_readfirstlane(tid >> 7); q < Q; q += NT / 128) {
            if (g >= G) continue;
            // (one pointer stepped by a scalar stride: the compiler's own 8-fold unrolling of the indexed form spent more on its eight
            // 16-cycle address multiplications than the phase spends on the sums)
            double sum = 0.0;
            const double *pq = sAux + (size_t)g * RAUX + q;
            const int stp = G * RAUX;
#pragma unroll 2
            for (int k = 0; k < K; ++k) { sum += *pq; pq += stp; }
            double *L = sL + (size_t)g * LREC;
            if (q < pkH) L[6 * hs + q] = sum;
            else if (q == pkH) { L[lHinv] = 1.0 / sum; L[lH] = sum; maxh = fmax(maxh, fabs(sum)); }   // Hmm_inv (problem.cc:419-425)
            else if (q == pkBL) L[lBl] = -sum;
            else L[q - pkWE] = sum;                                 // the extrinsic is pattern-local block 0
        }
    }
    // The item's chi2, max |h_ll| and (GN) the previous step's gain-ratio partial: the wave partials go to LDS now, while the
    // per-thread terms are still in registers (kept until the combine phase they come back from scratch, a memory round
    // trip in front of it), and are added after this barrier by the thread that stores them.  The partial of the GN step:
    // thread g holds landmark g's term, so the sum is the one k_backsub forms — DPP inside waves 0 and 1, then wave 0 +
    // wave 1 — bit for bit.
    {
        // (only the waves that hold terms run the DPP trees — 20 instructions each, three of them on all 16 waves were 1 k cycles of
        // this phase —: chi2 lives in the observation threads, the step partial in the first G, max h_ll in the waves of quantity h)
        const int wv64 = (tid >> 6) * 64;
        double sc = (owe && tid < G) ? sL[(size_t)tid * LREC + lSc] : 0.0;
        double ws = 0.0, wsc = 0.0, wm = 0.0;
        if (wv64 < G * K) ws = d_wave_sum_to_lane63(chi_acc);
        if (owe && wv64 < G) wsc = d_wave_sum_to_lane63(sc);
        if ((tid >> 7) == pkH % (NT / 128)) wm = d_wave_max_to_lane63(maxh);      // (the waves that handled quantity h_ll: q = tid / 128 + pass * NT / 128)
        if ((tid & 63) == 63) { sRed[tid >> 6] = ws; sRed[NT / 64 + (tid >> 6)] = wsc; sRed[2 * (NT / 64) + (tid >> 6)] = wm; }
    }
    __syncthreads();

    // The 1024-thread kernels (one window alone, one round of workgroups): w, h, b_l of the item's landmarks for the back-substitution (the
    // next head, k_backsub) leave for HBM NOW — they are final, and nothing below writes the landmark records — so that their 256 bytes
    // per landmark drain under phase 2 and the combine phase instead of behind the kernel's last instruction; the barrier that follows
    // is LDS-only (d_lds_barrier: __syncthreads() would wait for these stores).  49.75 -> 49.41 us per GN iteration; under the batched
    // loop, where the device's HBM pipe is busy all the time, it costs 1.4 % (10.81 -> 10.96 us per window-iteration,
    // profiles/r05f_*): the half-width kernels keep the stores at the end.
    constexpr bool EARLY_LW = LIN_EARLY_LW >= 0 ? (LIN_EARLY_LW > 0 || NT == 1024) : false;
    if (EARLY_LW) {
        double *lw = T.lw + lw_w + it.lw_base;
        for (int e = tid; e < (6 * nb + 2) * G; e += NT) {
            const int r = e / G, g = e - r * G;
            const double *L = sL + (size_t)g * LREC;
            lw[e] = (r < 6 * nb) ? L[r] : (r == 6 * nb ? L[lH] : L[lBl]);
        }
    }
    // ---------------- phase 2: the item's contribution as 16x16 products on the matrix cores ----------------
    // Streaming the LDS rows through VALU strips was bound by LDS bandwidth (9 bytes per FMA).  v_mfma_f64_16x16x4
    // takes one operand element per lane, so the same sums cost two LDS reads per 1024 FMAs:
    //   direct  for every observation index k: C_k = V_k^T V_k, V_k = the 2G whitened rows of the item's k-th
    //           observations with columns [host 6 | target 6 | extrinsic 6]  (one tile; three with the extrinsic)
    //   Schur   S = - sum_g w_g w_g^T / h_g over the 6nb pattern columns (lower tiles)
    // One wave per product; the b vectors are plain sums over the landmarks (LIN_VS partials each).
    LIN_EXIT(2);
    STAMP(T, 3);
    const int D = 6 * nb;
    const int ntd = use_ext ? 3 : 1;
    const int TS = (D + 16) >> 4, nts = TS * (TS + 1) / 2;      // (row D of the Schur tiles carries the correction of b)
    {
        const int wave = tid >> 6, lane = tid & 63, cl = lane & 15, rg = lane >> 4;
        const int nwork = K * ntd + nts;
        for (int wk = wave; wk < nwork; wk += NT / 64) {
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (wk < K * ntd) {
                const int k = wk / ntd, t = wk - k * ntd;          // t: 0 -> tile (0,0), 1 -> (1,0), 2 -> (1,1)
                const int Dk = use_ext ? 18 : 12;
                const int ca = 16 * (t == 0 ? 0 : 1) + cl, cb = 16 * (t == 2 ? 1 : 0) + cl;
                const int cac = min(ca, Dk - 1), cbc = min(cb, Dk - 1);
                // column -> offset inside a row record (host 0..5, target 6..11, extrinsic 12..17), + which of the 2 rows
                const int oa = (cac < 6 ? offH + cac : (cac < 12 ? offT + cac - 6 : offE + cac - 12)) + (rg & 1) * 6;
                const int ob = (cbc < 6 ? offH + cbc : (cbc < 12 ? offT + cbc - 6 : offE + cbc - 12)) + (rg & 1) * 6;
                // Per-lane operand stream: element (landmark g, this lane's row of the pair, this lane's column) sits at
                // pa + g * sa.  A padding column (>= Dk) streams a zero with stride 0 instead of being masked, so the
                // steady state is loads and MFMAs only; the loads of chunk ch+1 are issued before the MFMAs of chunk ch
                // and nothing touches them until the next iteration (no s_waitcnt in front of the matrix core).
                // Column Dk — the first padding column — is z (the row record's last two doubles): row Dk of the product is then
                // sum_g (L J)^T z = sum_g drho J^T Info r, the pose part of b per block, at no extra instruction.
                const double *plane = sRows + k * PLANE + (rg >> 1) * RROW;
                // (A column past Dk streams its clamped neighbour instead of zeros: what lands in the tile's unused rows / columns is never
                // read, and ONE stride for all lanes — a compile-time constant without an extrinsic block — lets the loads carry
                // immediate offsets: no address arithmetic between the matrix-core instructions.)
                const double *pa = ca == Dk ? plane + offZ + (rg & 1) : plane + oa;
                const double *pb = cb == Dk ? plane + offZ + (rg & 1) : plane + ob;
                const int sa = 2 * RROW, sb = 2 * RROW;             // 2 landmarks per MFMA step
                // (on a diagonal tile B is A: both streams are read all the same — a branch-free loop of loads and MFMAs
                // is worth more than the four reads it would save)
                const int steps_full = G >> 1;                       // steps whose two landmarks both exist
                const int chunks_full = steps_full >> 2;
                double va[4], vb[4], xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = pb[u * sb]; }      // (G >= 1: in range even if unused)
                // lgkmcnt(0) here, by hand: left to itself the compiler puts a partial wait on the loop header, i.e. in front
                // of every chunk's MFMAs and on the loads just issued for the next one (an LDS latency per chunk)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                // two chunks per trip, the operand registers taking turns (no copies); a load that would run past the last
                // full chunk re-reads the chunk before it instead of branching
                int ch = 0;
                // (the scheduling barriers keep the loads of the next chunk IN FRONT of this chunk's products: left alone the compiler
                // sinks every load to its use — fewer live registers — and each product then waits for an LDS latency)
                for (; ch + 2 <= chunks_full; ch += 2) {
                    pa += 4 * sa; pb += 4 * sb;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xa[u] = pa[u * sa]; xb[u] = pb[u * sb]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], vb[u], acc, 0, 0, 0);
                    const int nx = ch + 2 < chunks_full ? 4 : 0;
                    pa += nx * sa; pb += nx * sb;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = pb[u * sb]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], xb[u], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ch < chunks_full) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], vb[u], acc, 0, 0, 0);
                }
                STAMP(T, 14);
                // the remaining steps (fewer than 4 full ones, plus the half step of an odd G), landmark by landmark masked
                for (int st = 4 * chunks_full; 2 * st < G; ++st) {
                    const int g = 2 * st + (rg >> 1);
                    const double m = g < G ? 1.0 : 0.0;
                    const int gc = min(g, G - 1) - (rg >> 1);        // pa already points at landmark (rg >> 1)
                    const double *rec0 = sRows + k * PLANE + (rg >> 1) * RROW + (size_t)gc * RROW;
                    const double *qa = ca == Dk ? rec0 + offZ + (rg & 1) : rec0 + oa;
                    const double *qb = cb == Dk ? rec0 + offZ + (rg & 1) : rec0 + ob;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[0] * m, qb[0], acc, 0, 0, 0);
                }
            } else {
                const int ts = wk - K * ntd;
                int ta = 0;
                while ((ta + 1) * (ta + 2) / 2 <= ts) ++ta;
                const int tb = ts - ta * (ta + 1) / 2;
                const int a = 16 * ta + cl, bq = 16 * tb + cl;
                const int ac = min(a, D - 1), bc = min(bq, D - 1);
                // same streaming as above: element (landmark g, column) at pa + g * LREC, 4 landmarks per MFMA step;
                // B carries the -1/h_g of the landmark
                // Row / column D — the first padding index — is b_l: row D of the product is then - sum_g b_l w_g / h_g, the Schur
                // correction of b (problem.cc:429), at no extra instruction.
                // (columns past D: the clamped neighbour, one stride for all lanes, as above)
                const double *pa = sL + (size_t)rg * LREC + (a == D ? lBl : ac);
                const double *pb = sL + (size_t)rg * LREC + (bq == D ? lBl : bc);
                const double *ph = sL + (size_t)rg * LREC + lHinv;
                const int sa = 4 * LREC, sb = 4 * LREC, sh = 4 * LREC;
                const int chunks_full = (G >> 2) >> 2;               // chunks whose 16 landmarks all exist
                double va[4], vb[4], vh[4], xa[4], xb[4], xh[4];
#ifdef VIO_STAMPS
                if (tid == 256) g_stamps[12] = __builtin_amdgcn_s_memtime();
#endif
                if (chunks_full > 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = pb[u * sb]; vh[u] = ph[u * sh]; }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                int ch = 0;
                for (; ch + 2 <= chunks_full; ch += 2) {             // as above: two chunks per trip, registers taking turns
                    pa += 4 * sa; pb += 4 * sb; ph += 4 * sh;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { xa[u] = pa[u * sa]; xb[u] = pb[u * sb]; xh[u] = ph[u * sh]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], -(vb[u] * vh[u]), acc, 0, 0, 0);
                    const int nx = ch + 2 < chunks_full ? 4 : 0;
                    pa += nx * sa; pb += nx * sb; ph += nx * sh;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { va[u] = pa[u * sa]; vb[u] = pb[u * sb]; vh[u] = ph[u * sh]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], -(xb[u] * xh[u]), acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ch < chunks_full) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], -(vb[u] * vh[u]), acc, 0, 0, 0);
                }
#ifdef VIO_STAMPS
                if (tid == 256) g_stamps[13] = __builtin_amdgcn_s_memtime();
#endif
                for (int st = 16 * chunks_full; st < G; st += 4) {   // the remaining landmarks, masked
                    const int g = st + rg, gc = min(g, G - 1);
                    const double m = g < G ? 1.0 : 0.0;
                    const double *Lg = sL + (size_t)gc * LREC;
                    const double wa = a == D ? Lg[lBl] : Lg[ac], wb = bq == D ? Lg[lBl] : Lg[bc];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa * m, -(wb * Lg[lHinv]), acc, 0, 0, 0);
                }
            }
            STAMP(T, 15);
            double *tl = sTile + (size_t)wk * 256 + rg * 16 + cl;   // C/D image: row rg + 4v, column cl
#pragma unroll
            for (int v = 0; v < 4; ++v) tl[64 * v] = acc[v];
        }
        STAMP(T, 6);
        // (the b vectors need no sums of their own: row Dk of the direct products, row D of the Schur term)
        STAMP(T, 7);
    }
    if (EARLY_LW) d_lds_barrier();
    else __syncthreads();

    // ---------------- combine: thread per slab element ----------------
    LIN_EXIT(3);
    STAMP(T, 4);
    {
        double *out = T.slab + it.out_base;
        const int n_out = it.n_rows * 6;
        const int n_pair = (nb * (nb + 1) / 2) * 36;
        // the block totals from the wave partials of phase 1.5 (fixed order: wave 0 first)
        // (by two threads of the last wave, which has no slab element to form: on wave 0 these 48 dependent LDS reads and adds
        // sat in front of its share of the elements)
        const int tA = NT - 64, tB = NT - 32;
        double chi = 0.0, sc = 0.0, mh = 0.0;
        if (tid == tA || tid == tB) {
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                chi += sRed[w]; sc += sRed[NT / 64 + w]; mh = fmax(mh, sRed[2 * (NT / 64) + w]);
            }
        }
        STAMP(T, 12);
        // entry (x, y) of the k-th direct product; x, y: host 0..5, target 6..11, extrinsic 12..17
        auto cdir = [&](int k, int x, int y) -> double {
            const int hi = max(x, y), lo = min(x, y);
            const int t = hi < 16 ? 0 : (lo < 16 ? 1 : 2);
            return sTile[(size_t)(k * ntd + t) * 256 + (hi & 15) * 16 + (lo & 15)];
        };
        auto colof = [&](int ty, int i) { return ty == 1 ? i : (ty == 2 ? 6 + i : 12 + i); };
        for (int e = tid; e < n_out; e += NT) {
            double v = 0.0;
            if (e < n_pair) {
                const int pi = e / 36, rem = e - 36 * pi, i = rem / 6, j = rem - 6 * i;
                int p = 0, left = pi;
                while (left >= nb - p) { left -= nb - p; ++p; }
                const int q = p + left;
                const int a = 6 * p + i, bq = 6 * q + j;
                const int hi = (a >> 4) >= (bq >> 4) ? a : bq, lo = (a >> 4) >= (bq >> 4) ? bq : a;
                const int ta = hi >> 4, tb = lo >> 4;
                v = sTile[(size_t)(K * ntd + ta * (ta + 1) / 2 + tb) * 256 + (hi & 15) * 16 + (lo & 15)];
                const int tp = it.btype[p], tq = it.btype[q];
                if (tp != 2 && tq != 2) {                       // both touched by every observation
                    for (int k = 0; k < K; ++k) v += cdir(k, colof(tp, i), colof(tq, j));
                } else if (tp == 2 && tq == 2) {
                    if (p == q) v += cdir(it.bk[p], 6 + i, 6 + j);
                } else {
                    v += cdir(tp == 2 ? it.bk[p] : it.bk[q], colof(tp, i), colof(tq, j));
                }
            } else {
                const int ve = e - n_pair, which = ve / D, a = ve - which * D;
                const int p = a / 6, i = a - 6 * p, ty = it.btype[p];
                const int Dk = use_ext ? 18 : 12;
                if (which == 0) {                               // pose part of b = - sum drho J^T Info r (problem.cc:357): row Dk of the direct products
                    if (ty != 2) { for (int k = 0; k < K; ++k) v -= cdir(k, Dk, colof(ty, i)); }
                    else v = -cdir(it.bk[p], Dk, 6 + i);
                } else if (which == 1) {                        // its Schur correction sum_g (b_l / h)_g w_g: minus row D of the Schur term
                    const int ta = D >> 4, tb = a >> 4;
                    v = -sTile[(size_t)(K * ntd + ta * (ta + 1) / 2 + tb) * 256 + (D & 15) * 16 + (a & 15)];
                } else {                                        // direct diagonal (diag(Hessian_) before the Schur complement)
                    if (ty != 2) { for (int k = 0; k < K; ++k) v += cdir(k, colof(ty, i), colof(ty, i)); }
                    else v = cdir(it.bk[p], 6 + i, 6 + i);
                }
            }
            out[e] = v;
        }
        STAMP(T, 13);
        if (tid == tA) { out[n_out] = chi; out[n_out + 1] = mh; }
        if (owe && tid == tB) { T.step_part[2 * b + STEP_SCALE] = sc; T.step_part[2 * b + STEP_CHI] = 0.0; }
        if (owe && tid < G) {               // the landmark update of the head, out to HBM now
            const size_t li = (size_t)it.lm_base + tid;
            const double *L = sL + (size_t)tid * LREC;
            T.dxl[li] = L[lDl];
            T.invd[(size_t)cur * T.Ns + li] = L[lLam];
        }
        if (!EARLY_LW) {
            // w, h, b_l of the item's landmarks for the back-substitution (k_backsub), from the LDS records
            double *lw = T.lw + lw_w + it.lw_base;
            for (int e = tid; e < (6 * nb + 2) * G; e += NT) {
                const int r = e / G, g = e - r * G;
                const double *L = sL + (size_t)g * LREC;
                lw[e] = (r < 6 * nb) ? L[r] : (r == 6 * nb ? L[lH] : L[lBl]);
            }
        }
    }
    STAMP(T, 5);
    STAMP_FLUSH(T);
}
__global__ LIN_BOUNDS void k_linearize(DeviceTables T) { d_linearize_body<LIN_THREADS, 0, LIN_THREADS == 1024>(T); }
__global__ LIN_BOUNDS void k_linearize_g(DeviceTables T) { d_linearize_body<LIN_THREADS, 1, LIN_THREADS == 1024>(T); }      // plans with an extrinsic block (free extrinsic, marginalisation)
// The same kernel with half the threads and two workgroups to a CU (items of at most half the LDS): what the throughput policy's
// plans run on (vio_config.item_policy = VIO_ITEMS_THROUGHPUT).  When every CU has workgroup after workgroup to run, a CU sits
// idle for 1.7 us between two of them and a workgroup's head is two dependent round trips that overlap nothing
// (tools/diag_batch_stamps.py, profiles/r03h_*); two co-resident workgroups fill each other's gaps.  For one window in one
// round of workgroups it is slower (twice the fixed part of a workgroup): the latency policy keeps the 1024-thread kernel.
#define LIN_THREADS_H 512
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_h(DeviceTables T) { d_linearize_body<LIN_THREADS_H, 0>(T); }
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_gh(DeviceTables T) { d_linearize_body<LIN_THREADS_H, 1>(T); }

// Batched launches (vio_batch_gn_iteration): B independent windows in one launch, blockIdx.y = window.  The windows'
// tables sit in a device array built once per batch; what changes from iteration to iteration travels as kernel
// arguments: the GN flags, and the parity of the iteration count, which every window's own starting `cur` is flipped by.
struct BatchArgs {
    const DeviceTables *tabs;
    int32_t gn_flags;
    int32_t parity;              // GN loop: 0 / 1, flips the window's cur_hint; -1 (batched LM solve): `cur` comes from the window's LmState
    int32_t gate;                // d_gated_off's gate (batched LM solve: every window follows its own LmState)
};
__device__ __forceinline__ DeviceTables d_batch_tables(const BatchArgs &a) {
    DeviceTables T = a.tabs[blockIdx.y];
    T.gn_flags = a.gn_flags;
    if (a.parity < 0) T.cur_hint = a.parity; else T.cur_hint ^= a.parity;
    T.lm_gate = a.gate;
    return T;
}
__global__ LIN_BOUNDS void k_linearize_b(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;       // the grid is the widest window's
    d_linearize_body<LIN_THREADS, 0>(T);
}
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_hb(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;
    d_linearize_body<LIN_THREADS_H, 0>(T);
}
__global__ LIN_BOUNDS void k_linearize_gb(BatchArgs a) {            // (a batch with an extrinsic block in some window's plan)
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;
    d_linearize_body<LIN_THREADS, 1>(T);
}
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_ghb(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;
    d_linearize_body<LIN_THREADS_H, 1>(T);
}

// ---------------------------------------------------------------------------------------------------------
// k_reduce: fixed-order sum of the item slabs through inverted lists built at upload time.
//   list b < 78           camera block pair (P,Q): entries = slab offset of that 6x6 block
//   list 78 + P           camera block P vectors: entries = (offset of b_dir block, 6*nb)
//   list 90               chi / max h: entries = offset
// ---------------------------------------------------------------------------------------------------------
struct ReduceTables {
    const int32_t *list_off;     // [92]
    const int32_t *list;         // offsets (pairs: 1 int per entry; vectors: 2 ints per entry)
    const double *slab;
    double *vis;
    const double *step_part;     // GN mode: per-item partials of the previous step (k_backsub), or null
    int32_t n_step;              // items
    int32_t gate;                // see d_gated_off
    const LmState *lm;
    const double *jtinv;         // GN mode with a prior: err_prior of the previous step is formed here (blocks >= 91), or null
    const double *bprior;        // b_prior after that step
    double *errprior;
    int32_t lm_loop;             // 1 (vio_solve's loop): step_part / jtinv count only while lm->pending; bprior / errprior are the bases of
                                 // the two copies and the step's copy is lm->cur ^ 1
};

#define RED_THREADS 1024
#define RED_ERR_BLOCKS ((VIO_PRD + RED_THREADS / 64 - 1) / (RED_THREADS / 64))

// err_prior[i] = -(Jt_prior_inv b_prior.head(156))[i]  (problem.cc:474), one wave per row; the same sum whoever calls it
__device__ __forceinline__ double d_errprior_dot(double j0, double j1, double j2, double y0, double y1, double y2) {
    double s = 0;
    s += -j0 * y0;
    s += -j1 * y1;
    s += -j2 * y2;
    return d_wave_sum_to_lane63(s);
}
__device__ __forceinline__ void d_errprior_row(const double *jt, const double *b, double *err, int i, int lane) {
    const bool in2 = lane + 128 < VIO_PRD;
    const double j0 = jt[i * VIO_PRD + lane], j1 = jt[i * VIO_PRD + lane + 64], j2 = in2 ? jt[i * VIO_PRD + lane + 128] : 0.0;
    const double y0 = b[lane], y1 = b[lane + 64], y2 = in2 ? b[lane + 128] : 0.0;
    const double s = d_errprior_dot(j0, j1, j2, y0, y1, y2);
    if (lane == 63) err[i] = s;
}

// FUSED (the three-launch path of the chain order, vio_pose_solve_chain.h): the workgroup that has summed a block also adds the IMU and
// prior terms and writes the entries straight into the image of the pose system — k_assemble_c's work without its launch; 20 more
// workgroups write the 99 rows of the speed-bias variables (no visual part).  The hooks are defined with the chain layout.
// (the *_pre hooks fetch what the entry takes besides the sum — IMU and prior terms, which depend on the indices only — at the top of the
//  workgroup, with the list bounds: behind the sum they were two more dependent round trips at the kernel's end)
__device__ double d_fused_pair_pre(const DeviceTables &T, int b, int tid);
__device__ void d_fused_vec_pre(const DeviceTables &T, int P, int tid, double &extra, double &dgrest);
__device__ void d_fused_pair(const DeviceTables &T, int b, int tid, double tot, double rest);
__device__ void d_fused_vec(const DeviceTables &T, int P, int tid, double bd, double bc, double dg, double extra, double dgrest);
__device__ void d_fused_sb_row(const DeviceTables &T, int r, int tid);
#define RED_SB_BLOCKS 20          // five rows of speed-bias variables per workgroup (5 x 171 threads)
template <bool FUSED>
__device__ __forceinline__ void d_reduce_body(const ReduceTables &R, const DeviceTables *Tp = nullptr) {
    // A list is cut into interleaved slots (entry e belongs to slot e mod nslots); a group of 36 (18) threads owns a
    // slot and sums its entries with 8 gathers in flight, then the slots are added in slot order: the summation order
    // is fixed by the list, not by timing.  Three dependent round trips (offsets, list, slab) whatever the list length.
    __shared__ double sV[56 * 18 + 8];
    const int b = blockIdx.x, tid = threadIdx.x;
    // (the list bounds are requested before the gate looks at LmState: one round trip for both)
    const int lo_pre = b < VIO_NPAIR + VIO_NCB + 1 ? R.list_off[b] : 0, hi_pre = b < VIO_NPAIR + VIO_NCB + 1 ? R.list_off[b + 1] : 0;
    double pre0 = 0.0, pre1 = 0.0;
    if (FUSED) {
        if (b < VIO_NPAIR) { if (tid < 36) pre0 = d_fused_pair_pre(*Tp, b, tid); }
        else if (b < VIO_NPAIR + VIO_NCB) { if (tid < 6) d_fused_vec_pre(*Tp, b - VIO_NPAIR, tid, pre0, pre1); }
    }
    if (d_gated_off(R.lm, R.gate)) return;
    const bool step_owed = !R.lm_loop || R.lm->pending != 0;
    if (FUSED && b >= VIO_NPAIR + VIO_NCB + 1 && b < VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS) {
        d_fused_sb_row(*Tp, b - (VIO_NPAIR + VIO_NCB + 1), tid);
        return;
    }
    if (b >= VIO_NPAIR + VIO_NCB + 1) {          // GN mode: err_prior of the step k_pose_solve took (it left b_prior' only)
        const int row = (b - (VIO_NPAIR + VIO_NCB + 1) - (FUSED ? RED_SB_BLOCKS : 0)) * (RED_THREADS / 64) + (tid >> 6);
        const int copy = R.lm_loop ? (R.lm->cur ^ 1) : 0;
        if (row < VIO_PRD && step_owed) d_errprior_row(R.jtinv, R.bprior + copy * 176, R.errprior + copy * 160, row, tid & 63);
        return;
    }
    const int lo = lo_pre, hi = hi_pre;
    if (b < VIO_NPAIR) {
        constexpr int W = 36, NS = RED_THREADS / W;           // 28 slots
        const int s = tid / W, t = tid - s * W;
        const int n = hi - lo;
        if (s < NS) {
            double acc = 0.0;
            int e = s;
            for (; e + 7 * NS < n; e += 8 * NS) {
                int off[8];
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) off[u] = R.list[lo + e + u * NS];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = R.slab[(size_t)off[u] + t];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            {   // tail: up to 7 entries, still all in flight together
                int off[7];
                double v[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) off[u] = (e + u * NS < n) ? R.list[lo + e + u * NS] : -1;
#pragma unroll
                for (int u = 0; u < 7; ++u) v[u] = off[u] >= 0 ? R.slab[(size_t)off[u] + t] : 0.0;
#pragma unroll
                for (int u = 0; u < 7; ++u) if (off[u] >= 0) acc += v[u];
            }
            sV[s * W + t] = acc;
        }
        __syncthreads();
        if (tid < W) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < NS; ++q) tot += sV[q * W + tid];
            R.vis[VIS_H + b * 36 + tid] = tot;         // block b = VIS_PAIR(P, Q), entry (i, j) = tid: the mirror image is not stored
            if (FUSED) d_fused_pair(*Tp, b, tid, tot, pre0);
        }
    } else if (b < VIO_NPAIR + VIO_NCB) {
        constexpr int W = 18, NS = RED_THREADS / W;           // 56 slots
        const int P = b - VIO_NPAIR;
        const int s = tid / W, t = tid - s * W;
        const int kind = t / 6, i = t % 6;
        const int n = (hi - lo) / 2;
        if (s < NS) {
            double acc = 0.0;
            for (int e = s; e < n; e += 4 * NS) {
                int off[4], str[4];
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = e + u * NS < n;
                    off[u] = ok ? R.list[lo + 2 * (e + u * NS)] : -1;
                    str[u] = ok ? R.list[lo + 2 * (e + u * NS) + 1] : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = off[u] >= 0 ? R.slab[(size_t)off[u] + kind * str[u] + i] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (off[u] >= 0) acc += v[u];
            }
            sV[s * W + t] = acc;
        }
        __syncthreads();
        if (tid < 6) {
            double bd = 0.0, bc = 0.0, dg = 0.0;
#pragma unroll
            for (int q = 0; q < NS; ++q) { bd += sV[q * W + tid]; bc += sV[q * W + 6 + tid]; dg += sV[q * W + 12 + tid]; }
            R.vis[VIS_BDIR + 6 * P + tid] = bd;
            R.vis[VIS_BRED + 6 * P + tid] = bd - bc;     // bpp - (Hpm*Hmm^-1)*bmm (problem.cc:429)
            R.vis[VIS_DIAG + 6 * P + tid] = dg;
            if (FUSED) d_fused_vec(*Tp, P, tid, bd, bc, dg, pre0, pre1);
        }
    } else {
        // chi2 and max|h_ll| over the items, and the landmark part of the previous step's gain-ratio denominator (summed here so that it is
        // in the exchange buffer when the shards gather it): threads stride the lists, then the waves' sums (DPP, a fixed tree) and the
        // sixteen wave partials in wave order.  (Until round 5 two ten-level trees through LDS, twenty barriers: this workgroup was the
        // kernel's long pole.)
        __shared__ double sC[3 * (RED_THREADS / 64)];
        const bool with_step = R.step_part && step_owed;
        double chi = 0.0, mh = 0.0, sc = 0.0;
        if (with_step) for (int e = tid; e < R.n_step; e += RED_THREADS) sc += R.step_part[2 * e + STEP_SCALE];
        for (int e = lo + tid; e < hi; e += RED_THREADS) {
            const size_t o = (size_t)R.list[e];
            chi += R.slab[o];
            mh = fmax(mh, R.slab[o + 1]);
        }
        const double wc = d_wave_sum_to_lane63(chi), wm = d_wave_max_to_lane63(mh), ws = d_wave_sum_to_lane63(sc);
        if ((tid & 63) == 63) { sC[tid >> 6] = wc; sC[RED_THREADS / 64 + (tid >> 6)] = wm; sC[2 * (RED_THREADS / 64) + (tid >> 6)] = ws; }
        __syncthreads();
        if (tid == 0) {
            double c = 0.0, m = 0.0, t = 0.0;
#pragma unroll
            for (int w = 0; w < RED_THREADS / 64; ++w) { c += sC[w]; m = fmax(m, sC[RED_THREADS / 64 + w]); t += sC[2 * (RED_THREADS / 64) + w]; }
            R.vis[VIS_CHI] = c; R.vis[VIS_MAXH] = m; R.vis[VIS_STEP] = 0.0;
            if (!R.step_part) R.vis[VIS_STEP + 1] = 0.0;
            else if (with_step) R.vis[VIS_STEP + 1] = t;
        }
    }
}
__global__ __launch_bounds__(RED_THREADS) void k_reduce(ReduceTables R) { d_reduce_body<false>(R); }
// batched: the tables of k_reduce are made of the window's DeviceTables; bit 0 of gn_flags here = "a step is waiting for its test"
__global__ __launch_bounds__(RED_THREADS) void k_reduce_b(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    const bool test_prev = (a.gn_flags & 1) != 0, err_prev = test_prev && T.has_prior;
    if ((int)blockIdx.x >= VIO_NPAIR + VIO_NCB + 1 && !err_prev) return;
    const int loop = T.cur_hint == -2, cur = loop ? 0 : T.cur_hint;
    ReduceTables R{T.list_off, T.list, T.slab, T.vis, test_prev ? T.step_part : nullptr, T.n_items, a.gate, T.lm,
                   err_prev ? T.Jtinv : nullptr, err_prev ? T.bprior + cur * 176 : nullptr, err_prev ? T.errprior + cur * 160 : nullptr, loop};
    d_reduce_body<false>(R);
}

// ---------------------------------------------------------------------------------------------------------
// k_assemble: row i of H_pp_schur_ (without lambda) = reduced visual (72 -> 171) + IMU blocks + prior
//   (problem.cc:365-384: prior rows/cols of a fixed extrinsic are zeroed; Marginalize adds it unmasked)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int imu_vblock(int a) { return a < 6 ? 0 : (a < 15 ? 1 : (a < 21 ? 2 : 3)); }

#define PS_N VIO_PD
#define PS_NP 176       // 171 padded to 11 block rows of 16 with identity pivots
// The permuted lower triangle is kept as 16x16 tiles (block row I >= block column J), each tile row-major with a row
// stride of 17 doubles: with that stride both the C/D image of v_mfma_f64_16x16x4_f64 (lane -> row (l>>4)+4v,
// column l&15) and its A/B image (lane -> row l&15, k = (l>>4)+4s) read and write LDS without bank conflicts.
// The right-hand side is a 177th row kept apart (192 doubles after the tiles).
#define PS_NB 16
#define PS_NT (PS_NP / PS_NB)
#define PS_TROW 17
#define PS_TS (PS_NB * PS_TROW)
#define PS_TILES (PS_NT * (PS_NT + 1) / 2)
#define PS_YOFF (PS_TILES * PS_TS)
#define PS_PACKED (PS_YOFF + 192)
static_assert(PS_SET_STRIDE == PS_PACKED, "stride between the two sets of Pg");
__device__ __forceinline__ int tix(int I, int J) { return (I * (I + 1) / 2 + J) * PS_TS; }
// element (r, c) with c's block <= r's block (for a diagonal tile both halves exist)
__device__ __forceinline__ int telem(int r, int c) { return tix(r >> 4, c >> 4) + (r & 15) * PS_TROW + (c & 15); }

// entry (i,j), i >= j, of H_pp_schur_ without lambda: the lower triangle as Eigen's LDLT reads it
// valid: bit k set = IMU edge k exists (a kernel argument: an entry costs one round trip, not two)
__device__ __forceinline__ int d_imu_mask(const DeviceTables &T) { return T.imu_mask; }
// key of the pivot rank sort: |d|, with NaN mapped to +inf — the ranks must be a permutation whatever the matrix holds
// (a landmark without information makes 1/h_ll infinite in the reference too; its NaNs must not become wild indices here)
__device__ __forceinline__ double d_rank_key(double d) { const double a = fabs(d); return a == a ? a : __builtin_huge_val(); }
// the IMU + prior part of entry (i,j), i >= j
__device__ __forceinline__ double d_hs_rest(const DeviceTables &T, int valid, int i, int j) {
    double vr = 0.0;
    if (i >= 6 && j >= 6) {
        const int fi = (i - 6) / 15;
        for (int k = fi - 1; k <= fi; ++k) {
            if (k < 0 || k >= 10 || !((valid >> k) & 1)) continue;
            if (T.marg_mode && k != 0) continue;
            const int a = i - (6 + 15 * k), bb = j - (6 + 15 * k);
            if (bb < 0 || bb >= 30) continue;
            const double *Tk = T.imu_out + k * IMU_OUT + IMU_T;
            // upper vertex blocks are computed, lower ones mirrored (problem.cc:347-355)
            vr += (imu_vblock(a) <= imu_vblock(bb)) ? Tk[a * 30 + bb] : Tk[bb * 30 + a];
        }
    }
    if (T.has_prior) {
        const bool mask = T.ext_fixed && !T.marg_mode && (i < 6 || j < 6);
        if (!mask) vr += T.Hprior[i * VIO_PD + j];
    }
    return vr;
}
__device__ __forceinline__ void d_hs_entry(const DeviceTables &T, int valid, int i, int j, double &vv, double &vr) {
    vv = 0.0;                           // reduced visual part; vr: IMU + prior part
    const int ci = full_to_cam(i), cj = full_to_cam(j);
    if (ci >= 0 && cj >= 0) {
        int P = ci / 6, a = ci - 6 * P, Q = cj / 6, bq = cj - 6 * Q;
        if (P > Q) { const int t0 = P; P = Q; Q = t0; const int t1 = a; a = bq; bq = t1; }
        vv = d_vis(T, VIS_H + VIS_PAIR(P, Q) * 36 + a * 6 + bq);
    }
    vr = d_hs_rest(T, valid, i, j);
}

// Row i of the right-hand sides: b_pp_schur_ (returned and stored in T.bs), the pose part of b_ (T.bfull, for the
// gain ratio) and diag(Hessian_) before the Schur complement (T.diagfull, for ComputeLambdaInitLM, problem.cc:511-516)
__device__ __forceinline__ double d_rhs_rest(const DeviceTables &T, int valid, int i, int cur) {     // the IMU + prior part of row i of b
    const bool mask_i = T.ext_fixed && !T.marg_mode && i < 6;
    double extra = 0.0;
    if (i >= 6) {
        const int fi = (i - 6) / 15;
        for (int k = fi - 1; k <= fi; ++k) {
            if (k < 0 || k >= 10 || !((valid >> k) & 1)) continue;
            if (T.marg_mode && k != 0) continue;
            const int a = i - (6 + 15 * k);
            if (a < 0 || a >= 30) continue;
            extra -= T.imu_out[k * IMU_OUT + IMU_G + a];
        }
    }
    if (T.has_prior && !mask_i) extra += T.bprior[cur * 176 + i];
    return extra;
}
__device__ __forceinline__ double d_rhs_entries(const DeviceTables &T, int valid, int i, int cur, int wset) {
    const int ci = full_to_cam(i);
    double bred = 0.0, bdir = 0.0, dv, dr;
    if (ci >= 0) { bred = d_vis(T, VIS_BRED + ci); bdir = d_vis(T, VIS_BDIR + ci); }
    const double extra = d_rhs_rest(T, valid, i, cur);
    T.bs[i] = bred + extra;
    T.bfull[wset * 176 + i] = bdir + extra;
    d_hs_entry(T, valid, i, i, dv, dr);
    T.diagfull[i] = ((ci >= 0) ? d_vis(T, VIS_DIAG + ci) : 0.0) + dr;
    return bred + extra;
}

// Workgroup b < 171: row b of H_pp_schur_ in natural order (getters, marginalisation) and, for the solve, row b of
// the PERMUTED, tiled lower triangle Pg: the pivot order of Eigen's LDLT is the order of |diag + lambda|, which for
// lambda >= 0 and a non-negative diagonal does not depend on lambda, so it is fixed here once per linearisation
// (every workgroup recomputes the 171 ranks, 5 threads per entry: cheaper than one more launch).  Workgroups 171..175: identity padding.
// Workgroup 176: right-hand side row, b_pp_schur_, pose part of b_, diag(Hessian_).
#define ASM_THREADS 896        // 5 x 171 threads rank-sort, then 171 write the row
// the pivot ranks of the 171 diagonal entries: rank_i = #{j : d_j > d_i or (d_j == d_i and j < i)}, 5 threads per entry.
// On return (after its barriers) sCnt holds the five partial counts of every entry; needs >= 5 * 171 threads.
__device__ __forceinline__ void d_rank_sort(const double *sDg, int *sCnt, int t) {
    if (t < 5 * VIO_PD) {
        const int i = t % VIO_PD, part = t / VIO_PD;
        const int j0 = part * 35, j1 = min(VIO_PD, j0 + 35);
        const double di = sDg[i];
        int rank = 0;
        for (int j = j0; j < j1; ++j) {
            const double dj = sDg[j];
            rank += (dj > di || (dj == di && j < i)) ? 1 : 0;
        }
        sCnt[part * 176 + i] = rank;
    }
    __syncthreads();
}
__device__ __forceinline__ int d_rank_of(const int *sCnt, int i) { return sCnt[i] + sCnt[176 + i] + sCnt[352 + i] + sCnt[528 + i] + sCnt[704 + i]; }

// RANKS_GIVEN (batched launches): the ranks come from k_rank_b (T.rank, T.perm), the workgroup has 192 threads and skips the
// sort every workgroup of the single-window kernel repeats (one launch more is nothing once it is shared by the batch)
template <bool RANKS_GIVEN>
__device__ __forceinline__ void d_assemble_body(const DeviceTables &T) {
    __shared__ double sDg[176];
    __shared__ int sPerm[176];
    __shared__ int sCnt[RANKS_GIVEN ? 8 : 5 * 176];
    const int b = blockIdx.x, t = threadIdx.x;
    if (d_gated_off(T.lm, T.lm_gate)) return;
    const int cur = d_cur(T);
    const int valid = d_imu_mask(T);
    // Workgroup b < 171 owns NATURAL row b: its entries are requested together with the diagonal (one round trip for both)
    // and scattered once the ranks are known: natural row b is row rank(b) of the permuted matrix.
    double row_e = 0.0;
    int my_rank = 0, row_rank = 0;
    // The last workgroup (right-hand sides + the GN step test) is the kernel's long pole: four dependent round trips if its
    // loads are issued where they are used.  Everything it reads that does not depend on the ranks is requested here, before
    // the diagonal: the previous step's dx / b_ / err_prior, the scalars of the test, the terms of the right-hand sides.
    const bool last_wg = b == PS_NP;
    const bool test_prev = last_wg && d_step_owed(T, 1);
    const int rset = d_set_r(T), wset = d_set_w(T);
    double *Pg = T.Pg + wset * PS_SET_STRIDE;
    int32_t *perm_w = T.perm + wset * 176;
    double p_dx = 0.0, p_bf = 0.0, p_er = 0.0, p_lambda = 0.0, p_chi = 0.0, p_step = 0.0, p_lmchi = 0.0, p_imu[10], rhs_v = 0.0;
    if (test_prev) {
        p_lambda = T.lm->lambda;
        if (t < VIO_PD) { p_dx = T.dx[t]; p_bf = T.bfull[rset * 176 + t]; }          // b_ of the previous linearisation: read before d_rhs_entries replaces it
        if (T.has_prior && t < VIO_PRD) p_er = T.errprior[cur * 160 + t];
        if (t == 0) {
            p_chi = d_vis(T, VIS_CHI); p_step = d_vis(T, VIS_STEP + 1); p_lmchi = T.lm->chi;
#pragma unroll
            for (int k = 0; k < 10; ++k) p_imu[k] = ((valid >> k) & 1) ? T.imu_out[k * IMU_OUT + IMU_CHI] : 0.0;
        }
    }
    if (RANKS_GIVEN) {
        if (t < VIO_PD) {
            double wv = 0.0, wr = 0.0;
            if (b < VIO_PD) d_hs_entry(T, valid, max(b, t), min(b, t), wv, wr);
            row_e = wv + wr;
            my_rank = T.rank[t];
            sPerm[t] = perm_w[t];
            if (last_wg) rhs_v = d_rhs_entries(T, valid, t, cur, wset);
        }
        if (b < VIO_PD) row_rank = T.rank[b];
        __syncthreads();
    } else {
        if (t < VIO_PD) {
            double vv, vr;
            d_hs_entry(T, valid, t, t, vv, vr);
            double wv = 0.0, wr = 0.0;
            if (b < VIO_PD) d_hs_entry(T, valid, max(b, t), min(b, t), wv, wr);
            sDg[t] = d_rank_key(vv + vr);
            row_e = wv + wr;
            if (last_wg) rhs_v = d_rhs_entries(T, valid, t, cur, wset);
        }
        __syncthreads();
        d_rank_sort(sDg, sCnt, t);
        if (t < VIO_PD) { my_rank = d_rank_of(sCnt, t); sPerm[my_rank] = t; }
        if (b < VIO_PD) row_rank = d_rank_of(sCnt, b);
    }
    if (b < VIO_PD) {
        if (t < VIO_PD) {
            if (T.natural_hs) T.Hs[b * VIO_PD + t] = row_e;      // natural-order H_pp_schur_: only the getters and Marginalize read it
            const int i = row_rank, j = my_rank;
            if (j <= i) {                       // entry (i, j) of the permuted, tiled triangle
                Pg[telem(i, j)] = row_e;
                if (j < i && (j >> 4) == (i >> 4)) Pg[telem(j, i)] = row_e;      // upper half of a diagonal tile
            }
        }
        return;
    }
    __syncthreads();
    if (t >= 192) return;
    if (b < PS_NP) {
        if (t <= b) {
            Pg[telem(b, t)] = (t == b) ? 1.0 : 0.0;
            if (t < b && (t >> 4) == (b >> 4)) Pg[telem(t, b)] = 0.0;
        }
    } else {
        // GN mode: the step test of the PREVIOUS iteration (IsGoodStepInLM's bookkeeping, always accepted).  The chi2 of
        // the state that step produced is the one this linearisation has just computed and k_reduce summed (all-reduced
        // with the rest of vis when sharded), so nobody evaluates it twice.  b_ of the previous linearisation is read
        // before d_rhs_entries replaces it.
        double sp = 0.0, e2 = 0.0;
        if (test_prev) {
            if (t < VIO_PD) sp = p_dx * (p_lambda * p_dx + p_bf);
            if (T.has_prior && t < VIO_PRD) e2 = p_er * p_er;
        }
        if (t < VIO_PD) {
            sDg[t] = rhs_v;
            if (!RANKS_GIVEN) perm_w[t] = sPerm[t];
        }
        if (test_prev) {
            __shared__ double sSum[8];
            d_block_sum2<192>(sp, e2, sSum, t);
            if (t == 0) {
                LmState *lm = T.lm;
                double chi_imu = 0.0;
#pragma unroll
                for (int k = 0; k < 10; ++k) if ((valid >> k) & 1) chi_imu += p_imu[k];
                double total = p_chi + chi_imu;
                if (T.has_prior) total += sqrt(e2);             // err_prior_.norm(), not squared (problem.cc:554-556)
                const double tempChi = 0.5 * total;
                const double scale = 0.5 * (p_step + sp) + 1e-6;
                lm->chi_try = tempChi;
                lm->scale = scale;
                // vio_solve's loop: IsGoodStepInLM's verdict on these two numbers is k_pose_solve's first act (other workgroups
                // of this kernel are still reading lm->cur)
                if (T.cur_hint != -2) {
                    lm->rho = (p_lmchi - tempChi) / scale;
                    lm->trials += 1;
                    lm->chi = tempChi;
                    lm->cur = cur;
                    lm->accepted = 1;
                    lm->naccepted += 1;
                    lm->need_linearize = 1;
                    lm->false_cnt = 0;
                    if (!isfinite(tempChi)) lm->finite = 0;
                }
            }
        }
        __syncthreads();                        // the 192 remaining threads, all of them
        if (t < PS_NP) Pg[PS_YOFF + t] = (t < VIO_PD) ? sDg[sPerm[t]] : 0.0;
    }
}
__global__ __launch_bounds__(ASM_THREADS) void k_assemble(DeviceTables T) { d_assemble_body<false>(T); }
// batched: the ranks once per window (k_rank_b), then 177 light workgroups of 192 threads per window
__global__ __launch_bounds__(ASM_THREADS) void k_rank_b(BatchArgs a) {
    __shared__ double sDg[176];
    __shared__ int sCnt[5 * 176];
    const DeviceTables T = d_batch_tables(a);
    const int t = threadIdx.x;
    if (d_gated_off(T.lm, T.lm_gate)) return;
    if (t < VIO_PD) { double vv, vr; d_hs_entry(T, d_imu_mask(T), t, t, vv, vr); sDg[t] = d_rank_key(vv + vr); }
    __syncthreads();
    d_rank_sort(sDg, sCnt, t);
    if (t < VIO_PD) { const int r = d_rank_of(sCnt, t); T.rank[t] = r; T.perm[d_set_w(T) * 176 + r] = t; }
}
__global__ __launch_bounds__(192) void k_assemble_b(BatchArgs a) { const DeviceTables T = d_batch_tables(a); d_assemble_body<true>(T); }

// ---------------------------------------------------------------------------------------------------------
// k_pose_solve: single workgroup, 1024 threads.  (H_pp_schur_ + lambda I) dx = b_pp_schur_ (problem.cc:434-439).
//
// Pivoting: Eigen's LDLT picks, at step k, the largest |diagonal| among the NOT YET UPDATED trailing diagonal
// (it is a left-looking algorithm, Cholesky/LDLT.h:317-320), so the whole pivot order is a sort of |diag(A)| and is
// known before the factorisation starts: k_assemble rank-sorts the diagonal and writes the permuted lower triangle
// as 16x16 tiles (row stride 17); this kernel copies them into LDS (143 KB of the CU's 160 KB) and runs an unpivoted
// blocked right-looking LDL^T, 11 block columns of 16, as a task graph with look-ahead (see the loop below):
//   F(K)      factor the diagonal tile: one wave, lane = row, the identity's rows riding along give M_K = L_KK^-T
//   S(I,K)    U_IK = A_IK M_K, one MFMA product per tile below the diagonal (U = L D: columns stay unscaled)
//   U(I,J,K)  A_IJ -= U_IK D_K^-1 U_JK^T, four chained v_mfma_f64_16x16x4_f64 per tile, operands straight from the tiles
// The right-hand side rides along as row 176, so the forward substitution is free; the back-substitution with L^T
// runs block by block from the bottom on all waves: x_K = M_K (z_K - D_K^-1 sum_{J>K} U_JK^T x_J), one barrier per block.
// Then: trial pose states (UpdateStates :453-480, vertex_pose.cc:7-19) and their pair table.  The first-order prior
// update (:473-474) is formed here only on the stepwise path; the solve paths leave it to the kernels that follow.
// ---------------------------------------------------------------------------------------------------------
#define PS_THREADS 1024
#define PS_PRIOR_ROWS 13   // rows of H_prior per wave (14 waves x 13 >= 171)
#define PS_JT_ROWS 12      // rows of Jt_prior_inv per wave (14 waves x 12 >= 156)
#define PS_TY (PS_THREADS / 32)


__device__ __forceinline__ double d_readlane(double x, int lane) {      // lane is wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double d_fast_rcp(double d) {                // 1/d to within an ulp; 0 for d == 0
    if (d == 0.0) return 0.0;
    // v_rcp_f64 is good to 4.6e-8; one cubic step x (1 + e + e^2), e = 1 - d x, brings it to 1.1e-16 (the same as two Newton
    // steps, measured over 1M values) with three dependent FMAs instead of four: this sits on the chain of every pivot
    double x = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, x, 1.0);
    const double t = fma(e, e, e);
    return fma(x, t, x);
}

// F(K) of k_pose_solve: factor the 16x16 diagonal tile `tile` (row stride 17) in place, one wave.
// Lanes 16..31 hold the rows of the identity (read from sI) and end up with M = the tile's column operations
// applied to I (sM, 16 x 17); every other lane holds row (lane & 15) of the tile.
// A lone wave issues one instruction of any kind every 4 cycles and dependent fp64 FMAs do not stall it, so the cost
// is the instruction count plus whatever an s_waitcnt really waits.  Hence:
//   * the pivot column is not broadcast with v_readlane (two per value plus a hazard nop) but published to LDS —
//     lane c stores U(c,j) as soon as it is final — and read back as uniform loads, two values per instruction;
//   * it is published INTO the tile, transposed (U(c,j) at [17 j + c]): that is the factored tile's final place,
//     no copy-back (the pivots on its diagonal are what the S phase takes the reciprocals from, before M_K replaces it);
//   * the sched_barriers pin the order publish -> (4 updates, 2 refills) groups, which gives every load a full
//     reciprocal chain of slack; the reciprocals themselves are recomputed from the stored pivots by another
//     wave (sDinv), off this chain.
// Out of line: inlined into the kernel it pushes wave 0 over the 128 registers a 1024-thread workgroup allows.
typedef __attribute__((address_space(3))) double lds_double;   // generic pointers would become flat_load here
__device__ __noinline__ void ps_factor_diag(lds_double *tile, lds_double *sI, lds_double *sM, int lane) {
    asm volatile("" : "+v"(tile));       // one base register + immediate offsets for the uniform loads below
    const bool ident = (lane >> 4) == 1;
    lds_double *p0 = (ident ? sI : tile) + (lane & 15) * PS_TROW;
    double a0[PS_NB], u[PS_NB];
#pragma unroll
    for (int j = 0; j < PS_NB; ++j) a0[j] = p0[j];
    // where a lane publishes its a0[j]: tile rows at tile[17 j + row] (all copies of a row store the same value),
    // identity rows straight into their final place M[row][j]
    lds_double *wp = ident ? sM + (lane & 15) * PS_TROW : tile + (lane & 15);
    const int ws = ident ? 1 : PS_TROW;
    __builtin_amdgcn_sched_barrier(0);   // every row is in registers before the first publish overwrites the tile
    wp[0] = a0[0];
    double d = d_readlane(a0[0], 0);
#pragma unroll
    for (int c = 1; c < PS_NB; ++c) u[c] = tile[c];
#pragma unroll
    for (int j = 0; j < PS_NB; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const double l0 = a0[j] * d_fast_rcp(d);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): one wait per column instead of one per update; by now
        if (j + 1 < PS_NB) {                    // the previous column's loads have had the whole reciprocal to land
            a0[j + 1] = fma(-l0, u[j + 1], a0[j + 1]);
            wp += ws;
            wp[0] = a0[j + 1];
            d = d_readlane(a0[j + 1], j + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = j + 2; c < PS_NB; ++c) {
            a0[c] = fma(-l0, u[c], a0[c]);
            if (((c - j - 2) & 3) == 3 || c + 1 == PS_NB) {     // after every 4th update: refill what was just used
#pragma unroll
                for (int e = c - ((c - j - 2) & 3); e <= c; ++e) u[e] = tile[(j + 1) * PS_TROW + e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}


// The scalars of LmState a step test touches, as a value: k_pose_solve runs the test while its other waves may still be reading
// LmState (the gate, the set to copy), so it works on this copy and stores it once they are all past that.
struct LmRegs {
    double lambda, chi, ni, last_chi, rho, scale;
    int32_t cur, accepted, stop, iter, false_cnt, trials, naccepted, need_linearize, finite, max_iter, stop_reason;
};
__device__ __forceinline__ void d_lm_load(const LmState *lm, LmRegs &r) {
    r.lambda = lm->lambda; r.chi = lm->chi; r.ni = lm->ni; r.last_chi = lm->last_chi; r.rho = lm->rho; r.scale = lm->scale;
    r.cur = lm->cur; r.accepted = lm->accepted; r.stop = lm->stop; r.iter = lm->iter; r.false_cnt = lm->false_cnt; r.trials = lm->trials;
    r.naccepted = lm->naccepted; r.need_linearize = lm->need_linearize; r.finite = lm->finite; r.max_iter = lm->max_iter; r.stop_reason = lm->stop_reason;
}
__device__ __forceinline__ void d_lm_store(LmState *lm, const LmRegs &r) {
    lm->lambda = r.lambda; lm->chi = r.chi; lm->ni = r.ni; lm->last_chi = r.last_chi; lm->rho = r.rho; lm->scale = r.scale;
    lm->cur = r.cur; lm->accepted = r.accepted; lm->stop = r.stop; lm->iter = r.iter; lm->false_cnt = r.false_cnt; lm->trials = r.trials;
    lm->naccepted = r.naccepted; lm->need_linearize = r.need_linearize; lm->finite = r.finite; lm->stop_reason = r.stop_reason;
}
// IsGoodStepInLM's verdict on a trial state's chi2 (problem.cc:541-573) and Problem::Solve's bookkeeping around it (:169-250).
// `cur` is the accepted copy the step started from.  mode 0: LM; 1: always accept, no damping update (a flushed GN step).
// (the traces go straight to `trace`: nobody but the host reads them)
__device__ void d_lm_verdict(LmRegs &lm, LmState *trace, int mode, double tempChi, double scale, int cur) {
    const double rho = (lm.chi - tempChi) / scale;
    lm.rho = rho; lm.scale = scale;
    lm.trials += 1;
    const bool finite = isfinite(tempChi);
    if (mode == 1 || (rho > 0 && finite)) {
        if (mode == 0) {
            // (problem.cc:561: 1 - pow(2 rho - 1, 3).  The cube as two products: within an ulp of the power function's value, like the device's own
            //  pow — neither is the host library's —, and 200 instructions less on the one lane the whole LM slot waits for)
            const double t2r = 2 * rho - 1;
            double alpha = 1. - (t2r * t2r) * t2r;
            alpha = fmin(alpha, 2. / 3.);
            const double scaleFactor = fmax(1. / 3., alpha);
            lm.lambda *= scaleFactor;
            lm.ni = 2;
        }
        lm.chi = tempChi;
        lm.cur = cur ^ 1;
        lm.accepted = 1;
        lm.naccepted += 1;
        lm.need_linearize = 1;
        lm.false_cnt = 0;
    } else {
        lm.lambda *= lm.ni;
        lm.ni *= 2;
        lm.accepted = 0;
        lm.need_linearize = 0;
        lm.false_cnt += 1;
    }
    if (!finite) lm.finite = 0;
    if (mode == 0 && (lm.accepted || lm.false_cnt >= 10)) {     // the inner while of Problem::Solve ends
        lm.iter += 1;
        lm.false_cnt = 0;
        if (lm.last_chi - lm.chi < 1e-5) { lm.stop = 1; lm.stop_reason = 1; }
        lm.last_chi = lm.chi;
        if (lm.iter >= lm.max_iter) lm.stop = 1;
        if (!lm.stop && lm.iter < 128) { trace->chi_trace[lm.iter] = lm.chi; trace->lambda_trace[lm.iter] = lm.lambda; }
    }
}

__device__ __forceinline__ void d_pose_solve_body(const DeviceTables &T) {
    double *P = dyn_smem;                          // 66 tiles of 16x17, then the rhs row (192)
    double *sY = P + PS_YOFF;
    double *sDinv = P + PS_PACKED;                 // 176
    double *sM = sDinv + 176;                      // 16 x 17: the panel's column operations applied to the identity
    double *sI = sM + PS_TS;                       // 16 x 17 identity
    double *sX = sI + PS_TS;                       // 192 solution in pivot order
    double *sDx = sX + 192;                        // 176 solution in natural order
    double *sR = sDx + 176;                        // 108 rotations
    double *sB = sR + 112;                         // 176 trial b_prior
    int *sPerm = (int *)(sB + 176);                // 176
    double *sState = sB + 176 + 88;                // 184: the current pose / speed-bias states
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    LmState *lm = T.lm;
    __shared__ LmRegs sLm;
    if (d_gated_off(lm, T.lm_gate)) return;
    // vio_solve's loop (cur_hint -2): this kernel opens with the verdict on the step the previous one took — k_assemble has
    // just put the chi2 of its trial state and the gain ratio's denominator into LmState — by thread 0, under the copy below.
    const bool lm_loop = T.cur_hint == -2;
    int cur = lm_loop ? 0 : d_cur(T);
    double lambda = lm_loop ? 0.0 : lm->lambda;
    const int n = PS_N, NP = PS_NP;
#ifdef VIO_STAMPS
    unsigned long long t_panel = 0, t_trail = 0, t_mark = 0, t_start = __builtin_amdgcn_s_memtime();
    if (tid < 64 && T.dbg) { T.dbg[16 + tid] = 0; }
    __syncthreads();
#define PS_MARK() (t_mark = __builtin_amdgcn_s_memtime())
#define PS_ADD(acc) do { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); acc += now__ - t_mark; t_mark = now__; } while (0)
#define PS_OUT(slot) do { if (tid == 0 && T.dbg) T.dbg[slot] = __builtin_amdgcn_s_memtime() - t_start; } while (0)
#else
#define PS_MARK() do { } while (0)
#define PS_ADD(acc) do { } while (0)
#define PS_OUT(slot) do { } while (0)
#endif

    // Fast path: the permuted tiles k_assemble wrote (pivot order fixed with lambda = 0) are valid as long as
    // |diag + lambda| is still non-increasing along the diagonal — always the case for lambda >= 0 on a
    // non-negative diagonal.  Coalesced 16-byte loads; lambda goes on the 171 real pivots.
    // In vio_solve's loop the copy starts from the set k_assemble has just written (the system at the trial state: the one to
    // solve if the step is accepted); a rejected step solves the other set — the system it came from — again, copied in then.
    int set = lm_loop ? (lm->sys ^ lm->pending) : 0;
    bool rejected = false;
    for (int pass = 0; pass < 2; ++pass) {
        const double2 *src = reinterpret_cast<const double2 *>(T.Pg + set * PS_SET_STRIDE);
        double2 *dst = reinterpret_cast<double2 *>(P);
        // 9 loads per thread, all in flight before the first store (named scalars: a local array ends up in scratch)
        static_assert((PS_PACKED / 2 + PS_THREADS - 1) / PS_THREADS == 9, "copy below is written for 9 rounds");
#define PS_LD(q) const double2 v##q = src[min(tid + q * PS_THREADS, PS_PACKED / 2 - 1)];
#define PS_ST(q) dst[min(tid + q * PS_THREADS, PS_PACKED / 2 - 1)] = v##q;          /* clamped lanes rewrite the last pair */
        PS_LD(0) PS_LD(1) PS_LD(2) PS_LD(3) PS_LD(4) PS_LD(5) PS_LD(6) PS_LD(7) PS_LD(8)
        const int pm = (tid < n) ? T.perm[set * 176 + tid] : 0;
        if (pass == 0) {
            // the states, for the update at the end (which copy is current is known after the verdict: both are requested)
            double stv = (tid < STATE_STRIDE) ? T.state[cur * STATE_STRIDE + tid] : 0.0;
            const double stv1 = (lm_loop && tid < STATE_STRIDE) ? T.state[STATE_STRIDE + tid] : 0.0;
            if (lm_loop && tid == 0) {
                int go = 1, rej = 0, sys = lm->sys;
                d_lm_load(lm, sLm);
                if (lm->pending) {
                    d_lm_verdict(sLm, lm, 0, lm->chi_try, lm->scale, sLm.cur);
                    if (sLm.accepted) sys ^= 1; else rej = 1;
                    go = !sLm.stop;
                }
                sX[0] = go ? 1.0 : 0.0; sX[1] = (double)sLm.cur; sX[2] = sLm.lambda; sX[3] = (double)sys; sX[4] = (double)rej;
            }
            PS_ST(0) PS_ST(1) PS_ST(2) PS_ST(3) PS_ST(4) PS_ST(5) PS_ST(6) PS_ST(7) PS_ST(8)
            if (tid < n) sPerm[tid] = pm;
            if (tid < PS_TS) sI[tid] = (tid / PS_TROW == tid % PS_TROW) ? 1.0 : 0.0;
            __syncthreads();
            int set_now = 0;
            if (lm_loop) {
                // every wave is past the gate and has its copy's addresses: LmState may change now
                if (tid == 0) { d_lm_store(lm, sLm); lm->sys = (int)sX[3]; lm->pending = sX[0] != 0.0 ? 1 : 0; }
                if (sX[0] == 0.0) return;
                cur = (int)sX[1]; lambda = sX[2]; set_now = (int)sX[3]; rejected = sX[4] != 0.0;
                if (cur) stv = stv1;
            }
            if (tid < STATE_STRIDE) sState[tid] = stv;
            if (set_now == set) break;
            set = set_now;
            __syncthreads();        // (sX is read by everybody before the second copy's barrier lets anyone go on)
        } else {
            PS_ST(0) PS_ST(1) PS_ST(2) PS_ST(3) PS_ST(4) PS_ST(5) PS_ST(6) PS_ST(7) PS_ST(8)
            if (tid < n) sPerm[tid] = pm;
            __syncthreads();
        }
#undef PS_LD
#undef PS_ST
    }
    const int trial = cur ^ 1;
    int same = 1;
    for (int k = tid; k + 1 < n; k += PS_THREADS)
        same &= (fabs(P[telem(k, k)] + lambda) >= fabs(P[telem(k + 1, k + 1)] + lambda)) ? 1 : 0;
    same = __syncthreads_and(same);
    // (the gather below reads the reduced system of the LAST linearisation: after a rejected step of vio_solve's loop that is the
    // trial state's, not this one's.  No step then: the next slot linearises at the kept state again and solves from there.)
    if (lm_loop && rejected && !same) { if (tid == 0) lm->pending = 0; return; }
    if (same) {
        for (int k = tid; k < n; k += PS_THREADS) P[telem(k, k)] += lambda;
    } else {
        // slow path (e.g. a negative lambda): rank-sort |diag + lambda| here and gather entry by entry
        double *sDg = sX;
        const int valid = d_imu_mask(T);
        for (int i = tid; i < n; i += PS_THREADS) { double vv, vr; d_hs_entry(T, valid, i, i, vv, vr); sDg[i] = vv + vr + lambda; }
        __syncthreads();
        for (int i = tid; i < n; i += PS_THREADS) {
            const double di = d_rank_key(sDg[i]);
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const double dj = d_rank_key(sDg[j]);
                rank += (dj > di || (dj == di && j < i)) ? 1 : 0;
            }
            sPerm[rank] = i;
        }
        __syncthreads();
        for (int e = tid; e < PS_PACKED; e += PS_THREADS) {
            double v = 0.0;
            if (e < PS_YOFF) {
                const int ti = e / PS_TS, rem = e - ti * PS_TS, rr = rem / PS_TROW, cc = rem - rr * PS_TROW;
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= ti) ++I;
                const int J = ti - I * (I + 1) / 2;
                const int r0 = 16 * I + rr, c0 = 16 * J + cc;
                const int r = max(r0, c0), c = min(r0, c0);          // upper half of a diagonal tile: mirrored
                if (cc < 16) {
                    if (r < n) {
                        const int i = sPerm[r], j = sPerm[c];
                        double vv, vr;
                        d_hs_entry(T, d_imu_mask(T), max(i, j), min(i, j), vv, vr);
                        v = vv + vr + ((r == c) ? lambda : 0.0);
                    } else {
                        v = (r == c) ? 1.0 : 0.0;
                    }
                }
            } else {
                const int c = e - PS_YOFF;
                v = (c < n) ? T.bs[sPerm[c]] : 0.0;
            }
            P[e] = v;
        }
    }
    __syncthreads();
    PS_OUT(0);

    // Task graph of the blocked factorisation, with look-ahead:
    //   F(K)      factor the diagonal tile (K,K): the serial chain of 16 pivots, wave 0 only
    //   S(I,K)    U_IK = A_IK M_K for the tiles below it, one MFMA product per tile; M_K = the column operations of
    //             F(K) applied to the identity (the rows of I ride along in lanes 16..31 of F's wave)
    //   U(I,J,K)  A_IJ -= U_IK D_K^-1 U_JK^T
    // F(K+1) only needs U(K+1,K+1,K), so wave 0 does that one update and goes straight on to F(K+1) while the
    // other 15 waves do the rest of step K's updates.  Two barriers per step.
    const int lofs = (lane & 15) * PS_TROW + (lane >> 4);      // A image: row l&15, k = (l>>4) + 4s (+4 per s)
    const int cofs = (lane >> 4) * PS_TROW + (lane & 15);      // C/D and B image: row (l>>4) + 4v (+68 per v), col l&15
    auto update_tile = [&](int I, int J, int K, const double *nd) {
        const double *ta = P + tix(I, K) + lofs, *tb = P + tix(J, K) + lofs;
        double *tc = P + tix(I, J) + cofs;
        double av[4], bv[4];
        ps_v4d acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = ta[4 * q]; bv[q] = tb[4 * q]; acc[q] = tc[4 * PS_TROW * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q] * nd[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) tc[4 * PS_TROW * q] = acc[q];
    };

    PS_MARK();
    if (uwave == 0) ps_factor_diag((lds_double *)(P + tix(0, 0)), (lds_double *)sI, (lds_double *)sM, lane);
    __syncthreads();
    PS_ADD(t_panel);
    for (int K = 0; K < PS_NT; ++K) {
        const int k0 = K * PS_NB, k1 = k0 + PS_NB;
        const int nk = PS_NT - 1 - K;                           // block rows below the panel
        // ---- S: one tile per wave; the rhs row's 16 entries by wave 15 ----
        if (uwave == 0 && nk > 0) {
            // Wave 0 is the critical path (F(K) -> U_{K+1,K} -> A_{K+1,K+1} -> F(K+1)): it forms its S product TRANSPOSED,
            // (A M)^T = M^T A^T — the same loads with the operand roles swapped, the same products in the same order — so that
            // the accumulator IS the A/B operand image of U_{K+1,K} and the update of the next diagonal tile follows from
            // registers, inside this phase instead of after the barrier.  The pivots are still on the factored tile's
            // diagonal (M_K is parked around them: its own diagonal is 1).
            double *tt = P + tix(K + 1, K);
            double *td = P + tix(K + 1, K + 1) + cofs;
            const double *pd = P + tix(K, K) + (lane >> 4) * (PS_TROW + 1);
            double av[4], bv[4], pv[4];
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = tt[lofs + 4 * q]; bv[q] = sM[cofs + 4 * PS_TROW * q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { pv[q] = pd[4 * q * (PS_TROW + 1)]; acc2[q] = td[4 * PS_TROW * q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[q], av[q], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) tt[lofs + 4 * q] = acc[q];          // U_{K+1,K}[l & 15][(l >> 4) + 4 q]
#pragma unroll
            for (int q = 0; q < 4; ++q) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q], acc[q] * -d_fast_rcp(pv[q]), acc2, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) td[4 * PS_TROW * q] = acc2[q];
        } else if (uwave < nk) {
            double *tt = P + tix(K + 1 + uwave, K);
            double av[4], bv[4];
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = tt[lofs + 4 * q]; bv[q] = sM[cofs + 4 * PS_TROW * q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) tt[cofs + 4 * PS_TROW * q] = acc[q];
        } else if (uwave == PS_THREADS / 64 - 2) {
            // reciprocals of the pivots; then M_K takes the factored tile's place: nobody reads that tile again but the
            // back-substitution, which multiplies by M_K = L_KK^-T instead of running a 16-step chain through L_KK
            double dd = 0.0, mk[4];
            if (lane < PS_NB) dd = d_fast_rcp(P[tix(K, K) + lane * (PS_TROW + 1)]);
#pragma unroll
            for (int q = 0; q < 4; ++q) mk[q] = sM[((lane >> 4) + 4 * q) * PS_TROW + (lane & 15)];
            __builtin_amdgcn_sched_barrier(0);
            if (lane < PS_NB) sDinv[k0 + lane] = dd;
#pragma unroll
            for (int q = 0; q < 4; ++q)         // (not the diagonal: M_K's is 1, and the pivots stay readable)
                if ((lane >> 4) + 4 * q != (lane & 15)) P[tix(K, K) + ((lane >> 4) + 4 * q) * PS_TROW + (lane & 15)] = mk[q];
        } else if (uwave == PS_THREADS / 64 - 1) {
            double y = 0.0;
            if (lane < PS_NB) {
#pragma unroll
                for (int j = 0; j < PS_NB; ++j) y = fma(sY[k0 + j], sM[j * PS_TROW + lane], y);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < PS_NB) sY[k0 + lane] = y;
        }
        __syncthreads();
        PS_ADD(t_trail);
        if (nk == 0) break;
        // ---- U (+ F(K+1) on wave 0).  Items of the other waves: tiles 1 .. ntile-1 in row-major order of the
        //      trailing triangle (tile 0 = (K+1,K+1) is wave 0's), then the rhs row in chunks of 64 columns ----
        {
            const int ntile = nk * (nk + 1) / 2;
            const int nitem = ntile + ((nk * PS_NB + 63) >> 6);
            double nd[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nd[q] = -sDinv[k0 + (lane >> 4) + 4 * q];
            if (uwave == 0) {
#ifdef VIO_STAMPS
                const unsigned long long f0 = __builtin_amdgcn_s_memtime();
#endif
                // (its update of tile (K+1,K+1) happened in the S phase, from registers)
#ifdef VIO_STAMPS
                const unsigned long long f1 = __builtin_amdgcn_s_memtime();
#endif
                ps_factor_diag((lds_double *)(P + tix(K + 1, K + 1)), (lds_double *)sI, (lds_double *)sM, lane);
#ifdef VIO_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0 && T.dbg) { T.dbg[32 + K] = f1 - f0; T.dbg[48 + K] = __builtin_amdgcn_s_memtime() - f1; }
#endif
            } else {
#ifdef VIO_STAMPS
                const unsigned long long f0 = __builtin_amdgcn_s_memtime();
#endif
                // Waves 4, 8 and 12 share wave 0's SIMD and slow F down by taking its issue slots: they sit the
                // update out whenever 12 waves need no more rounds than 15 would.
                const int rounds15 = (nitem - 1 + 14) / 15, rounds12 = (nitem - 1 + 11) / 12;
                const bool spare = rounds12 == rounds15;
                const int nw = spare ? 12 : 15;
                const int slot = spare ? ((uwave & 3) ? uwave - 1 - (uwave >> 2) : -1) : uwave - 1;   // 0 .. nw-1
                for (int t = 1 + slot; slot >= 0 && t < nitem; t += nw) {
                    if (t < ntile) {
                        int ii = 0;
                        while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
                        const int jj = t - ii * (ii + 1) / 2;
                        update_tile(K + 1 + ii, K + 1 + jj, K, nd);
                    } else {
                        const int c = k1 + 64 * (t - ntile) + lane;
                        if (c < NP) {
                            const double *u = P + tix(c >> 4, K) + (c & 15) * PS_TROW;
                            double y = sY[c];
#pragma unroll
                            for (int kk = 0; kk < PS_NB; ++kk) y = fma(-(sY[k0 + kk] * sDinv[k0 + kk]), u[kk], y);
                            sY[c] = y;
                        }
                    }
                }
#ifdef VIO_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0 && T.dbg && uwave == 1) T.dbg[16 + K] = __builtin_amdgcn_s_memtime() - f0;
#endif
            }
        }
        __syncthreads();
        PS_ADD(t_panel);
    }
    PS_OUT(1);
#ifdef VIO_STAMPS
    if (tid == 0 && T.dbg) { T.dbg[8] = t_panel; T.dbg[9] = t_trail; }
#endif

    // ---- z = D^+ y, then x = L^-T z (solve of Cholesky/LDLT.h:558-600), block by block from the bottom:
    //      x_K = M_K (z_K - D_K^-1 sum_{J>K} U_JK^T x_J),  M_K = L_KK^-T parked in the diagonal tile by the S phase.
    //      Wave K owns block column K; only its lanes 0..15 work (lane = column): LDS bandwidth is what a round costs
    //      — a 64-lane broadcast read returns 1 KB however few distinct addresses it has (tools/microbench/barrier_cost.hip:
    //      a barrier is ~25 cycles, 16 waves x 8 such reads ~700) — so nothing is read by lanes that do not need it.
    //      Round J (x_J is out, one barrier per round): every wave K < J adds U_JK^T x_J to its running sum; wave J-1 then
    //      has its whole sum and finishes x_{J-1} with one 16-term product per lane, the v_j taken from lane j of the row by
    //      DPP (no LDS round trip).
    {
        const int i16 = lane & 15, K = uwave;
        const bool work = lane < PS_NB;
        double acc = 0.0, mw[PS_NB], ucol[PS_NB], own_dinv = 0.0, own_z = 0.0;
#pragma unroll
        for (int j = 0; j < PS_NB; ++j) { mw[j] = 0.0; ucol[j] = 0.0; }
        // PS_DOT16: out = init + sum_j arr[j] v_j with v_j taken from lane j of the 16-lane row (DPP row_newbcast): each lane
        // holds ONE element of the vector, so a vector costs one LDS read instruction, not eight broadcast reads (a 64-lane
        // read returns 1 KB whatever its addresses: ~20 cycles each on the critical path).  Two chains; s_nop: DPP reads a
        // VGPR the VALU has just written.
#define PS_DOT16(out, init, vin, arr) do { double x__ = (init), x2__ = 0.0; const double v__ = (vin); \
            asm volatile("s_nop 1\n\t" \
                         "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %14 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %15 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf" \
                         : "+v"(x__), "+v"(x2__) \
                         : "v"(v__), "v"(arr[0]), "v"(arr[1]), "v"(arr[2]), "v"(arr[3]), "v"(arr[4]), "v"(arr[5]), "v"(arr[6]), "v"(arr[7]), \
                           "v"(arr[8]), "v"(arr[9]), "v"(arr[10]), "v"(arr[11]), "v"(arr[12]), "v"(arr[13]), "v"(arr[14]), "v"(arr[15])); \
            (out) = x__ + x2__; } while (0)
        // column i16 of U_{I,K2}
#define PS_LOAD_COL(I_, K_) do { const double *src__ = P + tix((I_), (K_)) + i16; _Pragma("unroll") for (int r = 0; r < PS_NB; ++r) ucol[r] = src__[r * PS_TROW]; } while (0)
        if (K < PS_NT && work) {
            const double *src = P + tix(K, K) + i16 * PS_TROW;                       // row i16 of M_K
#pragma unroll
            for (int j = 0; j < PS_NB; ++j) mw[j] = (j == i16) ? 1.0 : src[j];     // (the tile's diagonal still holds the pivots)
            own_dinv = sDinv[K * PS_NB + i16];
            own_z = sY[K * PS_NB + i16] * own_dinv;                                  // sY holds y = L^-1 b (unscaled)
            if (K == PS_NT - 1) { double x; PS_DOT16(x, 0.0, own_z, mw); sX[K * PS_NB + i16] = x; }
            else PS_LOAD_COL(PS_NT - 1, K);
        }
        __syncthreads();                                                             // x_10 is out
        int J = PS_NT - 1;
        for (; J >= K + 1; --J) {
            if (work) {
                PS_DOT16(acc, acc, sX[J * PS_NB + i16], ucol);
                if (J > K + 1) PS_LOAD_COL(J - 1, K);       // does not depend on x: in flight across the barrier
                else { double x; PS_DOT16(x, 0.0, fma(-own_dinv, acc, own_z), mw); sX[K * PS_NB + i16] = x; }
            }
            __syncthreads();
        }
        for (; J >= 1; --J) __syncthreads();
    }
    __syncthreads();
    PS_OUT(2);
    // (dx and the trial states go out to HBM at the very end, from their LDS copies: a global store in front of a barrier
    // costs the store's whole round trip, and there are three barriers to come)
    for (int r = tid; r < n; r += PS_THREADS) sDx[sPerm[r]] = sX[r];
    __syncthreads();

    // prior: b' = b - H_prior*dx ; err' = -Jt_prior_inv * b'.head(156)   (problem.cc:466-475).  In the GN loop
    // (gn_flags bit 2) both are left to the next iteration, where they cost nothing: the rows of b' to the head of
    // k_linearize, err' to k_reduce (flush_decide does them when anybody else asks first).  An LM trial needs them
    // before its k_lm_decide: here, by waves 2..15, every load of a pass requested before the first is used.
    const bool prior_here = T.has_prior && !(T.gn_flags & 4);
    // The waves split by role (wave-uniform branches; both sides pass the same two barriers):
    //   waves 0,1: trial states = current (+) dx (UpdateStates, problem.cc:456-463) in place in the LDS copy of the states
    //              and out to the trial slot, then the pair table of the trial states, read from that LDS copy;
    //   waves 2..15: b' before the first barrier, the rows of Jt_prior_inv requested between the two, err' after.
    if (uwave < 2) {
        if (tid < 12) {
            double *p = (tid == 0) ? sState + STATE_EXT : sState + STATE_POSE + 7 * (tid - 1);
            const double *d = (tid == 0) ? sDx : sDx + 6 + 15 * (tid - 1);
            double tmp[7];
            d_pose_plus(p, d, tmp);
            for (int k = 0; k < 7; ++k) p[k] = tmp[k];
        } else if (tid >= 16 && tid < 16 + 99) {
            const int e = tid - 16, f = e / 9, k = e % 9;
            sState[STATE_SB + e] = sState[STATE_SB + e] + sDx[12 + 15 * f + k];
        }
        __syncthreads();
        // (XYZ windows have no host frames: k_linearize_xyz composes its camera maps from the states)
        // the pair table of the trial states: the rotation matrices here, its rows after this barrier, by all the waves
        // (XYZ windows have no host frames: k_linearize_xyz composes its camera maps from the states)
        if (T.lm_dim != 3) d_pair_rotations(sState, sR, tid);
        __syncthreads();
    } else {
        if (prior_here) {
            double hp[PS_PRIOR_ROWS][3], bp[PS_PRIOR_ROWS];
#pragma unroll
            for (int r = 0; r < PS_PRIOR_ROWS; ++r) {
                const int i = (uwave - 2) + 14 * r;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int j = lane + 64 * q;
                    hp[r][q] = (i < n && j < n) ? T.Hprior[i * n + j] : 0.0;
                }
                bp[r] = (i < n) ? T.bprior[cur * 176 + i] : 0.0;
            }
            const double x0 = sDx[lane], x1 = sDx[lane + 64], x2 = (lane + 128 < n) ? sDx[lane + 128] : 0.0;
#pragma unroll
            for (int r = 0; r < PS_PRIOR_ROWS; ++r) {
                const int i = (uwave - 2) + 14 * r;
                const double v = d_bprior_dot(hp[r][0], hp[r][1], hp[r][2], x0, x1, x2, bp[r]);
                if (lane == 63 && i < n) { sB[i] = v; T.bprior[trial * 176 + i] = v; }
            }
        }
        __syncthreads();
        double jp[PS_JT_ROWS][3];
        if (prior_here) {
#pragma unroll
            for (int r = 0; r < PS_JT_ROWS; ++r) {
                const int i = (uwave - 2) + 14 * r;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int j = lane + 64 * q;
                    jp[r][q] = (i < VIO_PRD && j < VIO_PRD) ? T.Jtinv[i * VIO_PRD + j] : 0.0;
                }
            }
        }
        __syncthreads();
        if (prior_here) {
            const double y0 = sB[lane], y1 = sB[lane + 64], y2 = (lane + 128 < VIO_PRD) ? sB[lane + 128] : 0.0;
#pragma unroll
            for (int r = 0; r < PS_JT_ROWS; ++r) {
                const int i = (uwave - 2) + 14 * r;
                const double s = d_errprior_dot(jp[r][0], jp[r][1], jp[r][2], y0, y1, y2);
                if (lane == 63 && i < VIO_PRD) T.errprior[trial * 160 + i] = s;
            }
        }
    }
    if (T.lm_dim != 3) d_pair_rows(sState, T.pairtab + trial * PAIRTAB_STRIDE, sR, tid, PS_THREADS);
    // dx and the trial states, from LDS (both final since the barriers above; the waves that did not write them read them
    // after those barriers)
    if (tid >= 192 && tid < 192 + n) T.dx[tid - 192] = sDx[tid - 192];
    if (tid >= 384 && tid < 384 + STATE_STRIDE) T.state[trial * STATE_STRIDE + (tid - 384)] = sState[tid - 384];
    PS_OUT(3);
}
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve(DeviceTables T) { d_pose_solve_body(T); }
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_b(BatchArgs a) { const DeviceTables T = d_batch_tables(a); d_pose_solve_body(T); }

// ---------------------------------------------------------------------------------------------------------
// k_backsub: one wave per item.  delta_lambda = Hmm^-1 (bmm - Hmp dx_p) (problem.cc:445), trial inverse depth,
// chi2 of the trial state, landmark part of the gain-ratio denominator.  mode 1: chi2 of the CURRENT state only.
// Blocks >= n_items evaluate the IMU chi2.
// ---------------------------------------------------------------------------------------------------------
#define BS_THREADS 128      // one thread per landmark of the item (G <= 128)
// workgroup b >= n_items of a back-substitution grid: r^T Info r of IMU edge b - n_items at state copy `which`
__device__ void d_backsub_imu_block(const DeviceTables &T, int mode, int which, int b, int lane) {
    const int k = b - T.n_items;
    if (lane != 0) return;
    double chi = 0.0;
    if (T.imu_valid[k]) {
        const double *st = T.state + which * STATE_STRIDE;
        const double *pre = T.pre + k * PRE_STRIDE;
        const double *pi = st + STATE_POSE + 7 * k, *pj = pi + 7, *si = st + STATE_SB + 9 * k, *sj = si + 9;
        ImuCommon c;
        d_imu_common(pre, pi, si, pj, c);
        double r[15];
        d_imu_residual(pre, T.gravity, pi, si, pj, sj, c, r);
        for (int i = 0; i < 15; ++i) {
            double t = 0;
            for (int j = 0; j < 15; ++j) t += pre[PRE_INFO + 15 * i + j] * r[j];
            chi += r[i] * t;
        }
    }
    double *part = (mode == 1) ? T.chi_part : T.step_part;
    part[2 * b + STEP_CHI] = chi;
    part[2 * b + STEP_SCALE] = 0.0;
}

__device__ __forceinline__ void d_backsub_body(const DeviceTables &T, int mode) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const LmState *lm = T.lm;
    if (d_gated_off(lm, T.lm_gate)) return;
    const int cur = d_cur(T);
    const int which = (mode == 1) ? cur : (cur ^ 1);
    // flush of a GN step (gn_flags bit 3): its b_prior' rows, which the next k_linearize would have formed
    if ((T.gn_flags & 8) && T.has_prior && (lane >> 6) == 1) d_bprior_rows(T, cur, cur ^ 1, b, T.n_step_blocks, lane & 63);
    if (b >= T.n_items) { d_backsub_imu_block(T, mode, which, b, lane); return; }
    __shared__ double sPairCD[VIO_MAXK * 12];
    __shared__ double sDxp[176];
    __shared__ ItemDesc sIt;
    // the pose update is the same for every item: requested together with the descriptor (both cold after the boundary)
    if (mode == 0) { sDxp[lane] = T.dx[lane]; if (lane + BS_THREADS < 176) sDxp[lane + BS_THREADS] = T.dx[lane + BS_THREADS]; }
    if (lane < (int)(sizeof(ItemDesc) / 4)) ((int32_t *)&sIt)[lane] = ((const int32_t *)(T.items + b))[lane];
    __syncthreads();
    const ItemDesc &it = sIt;
    const int G = it.G, K = it.K, nb = it.nb;
    const double *ptab = T.pairtab + which * PAIRTAB_STRIDE;
    for (int e = lane; e < K * 12; e += BS_THREADS) {
        const int k = e / 12, o = e % 12;
        sPairCD[e] = ptab[(it.host * 11 + it.target[k]) * PAIR_STRIDE + PAIR_C + o];     // C (9) then d (3) are adjacent
    }
    __syncthreads();
    double chi = 0.0, scale = 0.0;
    if (lane < G) {
        const int g = lane;
        const size_t li = (size_t)it.lm_base + g;
        double lam = T.invd[(size_t)cur * T.Ns + li];
        if (mode == 0) {
            const double *lw = T.lw + it.lw_base;
            double t = 0.0;
            for (int p = 0; p < nb; ++p) {
                const int cb = it.cam_block[p];
                const int base = cb == 0 ? 0 : 6 + 15 * (cb - 1);
#pragma unroll
                for (int i = 0; i < 6; ++i) t += lw[(size_t)(6 * p + i) * G + g] * sDxp[base + i];
            }
            const double h = lw[(size_t)(6 * nb) * G + g], bl = lw[(size_t)(6 * nb + 1) * G + g];
            const double dl = (1.0 / h) * (bl - t);
            T.dxl[li] = dl;
            lam = lam + dl;
            T.invd[(size_t)(cur ^ 1) * T.Ns + li] = lam;
            scale = dl * (lm->lambda * dl + bl);
        }
        const double il = 1.0 / lam;
        const double x = T.pts_i[2 * li], y = T.pts_i[2 * li + 1];
        const double pci[3] = {x * il, y * il, il};
        const double s_info = T.sqrt_info, info = s_info * s_info;
        for (int k = 0; k < K; ++k) {
            const double *C = sPairCD + 12 * k;
            const size_t o = (size_t)it.obs_base + (size_t)k * G + g;
            double pcj[3];
            d_m3_vec(C, pci, pcj);
#pragma unroll
            for (int m = 0; m < 3; ++m) pcj[m] += C[9 + m];
            const double iz = 1.0 / pcj[2];
            const double r0 = pcj[0] * iz - T.pts_j[2 * o], r1 = pcj[1] * iz - T.pts_j[2 * o + 1];
            const double e2 = r0 * (info * r0) + r1 * (info * r1);
            double rho0, rho1, rho2;
            d_loss(T.loss_type, T.loss_delta, e2, rho0, rho1, rho2);
            chi += (T.loss_type == 0) ? e2 : rho0;
        }
    }
    // fixed order: DPP sum inside each wave, then wave 0 + wave 1
    __shared__ double sSum[2 * (BS_THREADS / 64)];
    chi = d_wave_sum_to_lane63(chi);
    scale = d_wave_sum_to_lane63(scale);
    if ((lane & 63) == 63) { sSum[2 * (lane >> 6)] = chi; sSum[2 * (lane >> 6) + 1] = scale; }
    __syncthreads();
    if (lane == 0) {
        double c = 0.0, sc = 0.0;
#pragma unroll
        for (int w = 0; w < BS_THREADS / 64; ++w) { c += sSum[2 * w]; sc += sSum[2 * w + 1]; }
        // vio_chi2 (mode 1) has partials of its own so that it never disturbs a pending step test
        double *part = (mode == 1) ? T.chi_part : T.step_part;
        part[2 * b + STEP_CHI] = c; part[2 * b + STEP_SCALE] = sc;
    }
}
__global__ __launch_bounds__(BS_THREADS) void k_backsub(DeviceTables T, int mode) { d_backsub_body(T, mode); }
__global__ __launch_bounds__(BS_THREADS) void k_backsub_b(BatchArgs a, int mode) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;       // the grid is the widest window's
    d_backsub_body(T, mode);
}

// ---------------------------------------------------------------------------------------------------------
// k_step_sum: fixed-order sum of the visual per-item partials -> step_tot[0..1] (the second exchange buffer)
// k_lm_decide: IsGoodStepInLM (problem.cc:541-573) + the loop bookkeeping of Problem::Solve (:188-245)
//   mode 0: LM trial   mode 1: fixed-lambda GN step (always accept)   mode 2: chi2 only (no state change)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_step_sum(DeviceTables T, int mode) {
    __shared__ double s0[256];
    const int tid = threadIdx.x;
    const double *part = (mode == 2) ? T.chi_part : T.step_part;
    double c = 0, s = 0;
    for (int e = tid; e < T.n_items; e += 256) { c += part[2 * e + STEP_CHI]; s += part[2 * e + STEP_SCALE]; }
    d_block_sum2<256>(c, s, s0, tid);        // the same reduction as k_lm_decide's: sharded and unsharded runs agree bit for bit
    if (tid == 0) {
        T.step_tot[0] = c; T.step_tot[1] = s;
    }
}

// The body of k_lm_decide for a workgroup of NT threads (256: the kernel; 1024: the extra workgroup of k_linearize that
// runs the previous GN step's test).  Only the first 256 threads carry data and the wave partials are added in wave
// order, so both give the same bits.
template <int NT>
__device__ void d_lm_decide(const DeviceTables &T, int mode, int sum_local, double *s0, double *sImu, int tid) {
    LmState *lm = T.lm;
    const double *part = (mode == 2) ? T.chi_part : T.step_part;
    // everything this kernel reads is requested before anything is waited for: one round trip, not five.  The prior
    // error is read from both state copies because which one counts depends on lm->cur, itself still in flight.
    const int cur = lm->cur;
    const double lambda = lm->lambda;
    const bool act = tid < 256;
    double c = 0, s = 0, e0 = 0.0, e1 = 0.0, sp = 0.0;
    if (sum_local && act)   // unsharded: fold k_step_sum in (same fixed order)
        for (int e = tid; e < T.n_items; e += 256) { c += part[2 * e + STEP_CHI]; s += part[2 * e + STEP_SCALE]; }
    if (T.has_prior && act)
        for (int i = tid; i < VIO_PRD; i += 256) { const double v0 = T.errprior[i], v1 = T.errprior[160 + i]; e0 += v0 * v0; e1 += v1 * v1; }
    if (mode != 2 && act)
        for (int i = tid; i < VIO_PD; i += 256) { const double d = T.dx[i]; sp += d * (lambda * d + T.bfull[i]); }
    if (tid < T.n_imu_items) sImu[tid] = part[2 * (T.n_items + tid) + STEP_CHI];
    const int which = (mode == 2) ? cur : (cur ^ 1);
    double e = which ? e1 : e0;
    if (sum_local) {
        d_block_sum2<NT>(c, s, s0, tid);
        if (tid == 0) {
            T.step_tot[0] = c; T.step_tot[1] = s;
        }
    }
    d_block_sum2<NT>(e, sp, s0, tid);
    const double en2 = e, scale_p = sp;
    if (tid != 0) return;
    double chi_imu = 0.0;
    for (int k = 0; k < T.n_imu_items; ++k) chi_imu += sImu[k];
    double total = (sum_local ? c : d_step_tot(T, 0)) + chi_imu;
    if (T.has_prior) total += sqrt(en2);            // err_prior_.norm(), not squared (problem.cc:554-556)
    const double tempChi = 0.5 * total;
    lm->chi_try = tempChi;
    if (mode == 2) return;
    double scale = 0.5 * ((sum_local ? s : d_step_tot(T, 1)) + scale_p);
    scale += 1e-6;
    LmRegs regs;
    d_lm_load(lm, regs);
    d_lm_verdict(regs, lm, mode, tempChi, scale, cur);
    d_lm_store(lm, regs);
}

__global__ __launch_bounds__(256) void k_lm_decide(DeviceTables T, int mode, int sum_local) {
    __shared__ double s0[8];
    __shared__ double sImu[16];
    if (d_gated_off(T.lm, T.lm_gate)) return;
    d_lm_decide<256>(T, mode, sum_local, s0, sImu, threadIdx.x);
}

__global__ __launch_bounds__(256) void k_lm_decide_b(BatchArgs a, int mode) {
    __shared__ double s0[8];
    __shared__ double sImu[16];
    const DeviceTables T = d_batch_tables(a);
    if (d_gated_off(T.lm, T.lm_gate)) return;
    d_lm_decide<256>(T, mode, 1, s0, sImu, threadIdx.x);
}

// ComputeLambdaInitLM (problem.cc:497-522)
__device__ __forceinline__ void d_init_lm_body(const DeviceTables &T, int max_iter) {
    __shared__ double s0[256];
    const int tid = threadIdx.x;
    LmState *lm = T.lm;
    // (everything thread 0 adds up at the end is requested here, with LmState: the ten IMU edges' chi2 as ONE load, lane = edge — one after the other
    //  behind their valid flags they were twenty dependent round trips at the kernel's end, half of its 7.7 us; round 6.  The same sums in the same order.)
    const int valid = d_imu_mask(T);
    const double ichi = (tid < 10) ? T.imu_out[tid * IMU_OUT + IMU_CHI] : 0.0;
    const double vchi = (tid == 0) ? d_vis(T, VIS_CHI) : 0.0, vmaxh = (tid == 0) ? d_vis_maxh(T) : 0.0;
    const int cur = lm->cur;
    double e = 0.0, md = 0.0;
    if (T.has_prior)
        for (int i = tid; i < VIO_PRD; i += 256) { const double v = T.errprior[cur * 160 + i]; e += v * v; }
    for (int i = tid; i < VIO_PD; i += 256) md = fmax(md, fabs(T.diagfull[i]));
    const double en2 = d_block_sum<256>(e, s0, tid);
    const double maxd = d_block_max<256>(md, s0, tid);
    if (tid >= 64) return;
    double total = vchi;
#pragma unroll
    for (int k = 0; k < 10; ++k) { const double c = d_readlane(ichi, k); if ((valid >> k) & 1) total += c; }
    if (tid != 0) return;
    if (T.has_prior) total += sqrt(en2);
    const double chi = 0.5 * total;
    double maxDiagonal = fmax(maxd, vmaxh);             // max |h_ll| over all shards
    maxDiagonal = fmin(5e10, maxDiagonal);
    lm->ni = 2.;
    lm->chi = chi;
    lm->init_chi = chi;
    lm->lambda = 1e-5 * maxDiagonal;
    lm->last_chi = 1e20;
    lm->iter = 0; lm->false_cnt = 0; lm->trials = 0; lm->naccepted = 0; lm->stop = 0; lm->stop_reason = 0;
    lm->finite = 1; lm->max_iter = max_iter; lm->accepted = 0; lm->need_linearize = 0; lm->pending = 0; lm->sys = 0;
    lm->chi_trace[0] = chi; lm->lambda_trace[0] = lm->lambda;
}
__global__ __launch_bounds__(256) void k_init_lm(DeviceTables T, int max_iter) { d_init_lm_body(T, max_iter); }
__global__ __launch_bounds__(256) void k_init_lm_b(BatchArgs a, int max_iter) {
    const DeviceTables T = d_batch_tables(a);
    d_init_lm_body(T, max_iter);
}

// ---------------------------------------------------------------------------------------------------------
// k_triangulate: FeatureManager::triangulate (feature_manager.cpp:203-257), one thread per track.
//   The reference takes the last column of V of a JacobiSVD of the 2K x 4 matrix A (:243); that is the eigenvector of
//   the smallest eigenvalue of A^T A, found here with cyclic Jacobi rotations on the 4x4 (10 accumulated entries
//   instead of a 2K x 4 matrix per thread).  Camera poses R_f ric, P_f + R_f tic are staged once per workgroup.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_triangulate(TriTables Q) {
    __shared__ double sRc[VIO_NF * 9], sTc[VIO_NF * 3];
    const int tid = threadIdx.x;
    if (tid < VIO_NF) {
        double Rf[9], ric[9], tic[3], t[3];
        d_quat_to_R(Q.poses + 7 * tid + 3, Rf);
        d_quat_to_R(Q.ext + 3, ric);
        for (int k = 0; k < 3; ++k) tic[k] = Q.ext[k];
        d_m3_mul(Rf, ric, sRc + 9 * tid);                    // R1 = Rs[j] * ric   (:225)
        d_m3_vec(Rf, tic, t);
        for (int k = 0; k < 3; ++k) sTc[3 * tid + k] = Q.poses[7 * tid + k] + t[k];    // t1 = Ps[j] + Rs[j] * tic   (:224)
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + tid;
    if (i >= Q.n) return;
    const int sf = Q.start_frame[i];
    const int64_t e0 = Q.obs_offset[i];
    const int K = (int)(Q.obs_offset[i + 1] - e0);
    if (!(K >= 2 && sf < VIO_NF - 1 - 2)) return;            // used_num >= 2 && start_frame < WINDOW_SIZE - 2   (:207)
    if (Q.depth[i] > 0) return;                              // :210
    const double *R0 = sRc + 9 * sf, *t0 = sTc + 3 * sf;
    // every index below is a compile-time constant after unrolling: M and V live in registers, not in scratch
    double M[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) M[a][b] = 0.0;
    for (int j = 0; j < K; ++j) {
        const int f = min(sf + j, VIO_NF - 1);
        const double *R1 = sRc + 9 * f, *t1 = sTc + 3 * f;
        double dt[3] = {t1[0] - t0[0], t1[1] - t0[1], t1[2] - t0[2]}, t[3], R[9];
        d_m3_tvec(R0, dt, t);                                // t = R0^T (t1 - t0)
        d_m3_tmul(R0, R1, R);                                // R = R0^T R1
        double P[3][4];                                      // P = [R^T | -R^T t]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            P[r][0] = R[r]; P[r][1] = R[3 + r]; P[r][2] = R[6 + r];
            P[r][3] = -(R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
        }
        const double x = Q.pts[2 * (e0 + j)], y = Q.pts[2 * (e0 + j) + 1];
        const double nn = sqrt(x * x + y * y + 1.0);
        const double f0 = x / nn, f1 = y / nn, f2 = 1.0 / nn;
        double ra[4], rb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { ra[c] = f0 * P[2][c] - f2 * P[0][c]; rb[c] = f1 * P[2][c] - f2 * P[1][c]; }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) M[a][b] += ra[a] * ra[b] + rb[a] * rb[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) M[a][b] = M[b][a];
    // cyclic Jacobi: M -> diagonal, V accumulates the rotations (columns = eigenvectors)
    double V[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) V[a][b] = a == b ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = M[0][1] * M[0][1] + M[0][2] * M[0][2] + M[0][3] * M[0][3] + M[1][2] * M[1][2] + M[1][3] * M[1][3] + M[2][3] * M[2][3];
        const double dg = M[0][0] * M[0][0] + M[1][1] * M[1][1] + M[2][2] * M[2][2] + M[3][3] * M[3][3];
        if (off <= 1e-36 * dg) break;                          // off-diagonal mass below rounding of the diagonal: converged
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const double apq = M[p][q];
                if (apq != 0.0) {
                    const double theta = (M[q][q] - M[p][p]) / (2.0 * apq);
                    const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double mkp = M[k][p], mkq = M[k][q]; M[k][p] = c * mkp - sn * mkq; M[k][q] = sn * mkp + c * mkq; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double mpk = M[p][k], mqk = M[q][k]; M[p][k] = c * mpk - sn * mqk; M[q][k] = sn * mpk + c * mqk; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
                }
            }
    }
    // eigenvector of the smallest eigenvalue, picked with selects (no run-time index into V)
    double best = M[0][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int a = 1; a < 4; ++a) if (M[a][a] < best) { best = M[a][a]; v2 = V[2][a]; v3 = V[3][a]; }
    double dep = v2 / v3;                                    // svd_V[2] / svd_V[3]   (:245)
    if (dep < 0.1) dep = Q.init_depth;                       // :252
    Q.depth[i] = dep;
}

#include "vio_pose_solve_chain.h"
#include "vio_kernels_xyz.h"

void vio_launch_triangulate(const TriTables &Q, hipStream_t s) {
    hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((Q.n + 255) / 256)), dim3(256), 0, s, Q);
}

// ---------------------------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage inside the library)
// ---------------------------------------------------------------------------------------------------------
void vio_launch_prepare(const DeviceTables &T, hipStream_t s) { hipLaunchKernelGGL(k_prepare, dim3(1), dim3(128), 0, s, T); }
// threads: the plan's workgroup width (lin_threads_host(): one workgroup per CU; lin_threads_half_host(): two, inverse-depth plans of the throughput policy)
void vio_launch_linearize(const DeviceTables &T, int n_blocks, size_t lds_bytes, int threads, int use_ext, hipStream_t s) {
    if (T.lm_dim == 3 && threads == LIN_THREADS_H) hipLaunchKernelGGL(k_linearize_xyz_h, dim3(n_blocks), dim3(LIN_THREADS_H), lds_bytes, s, T);
    else if (T.lm_dim == 3) hipLaunchKernelGGL(k_linearize_xyz, dim3(n_blocks), dim3(LIN_THREADS), lds_bytes, s, T);
    else if (threads == LIN_THREADS_H && !use_ext) hipLaunchKernelGGL(k_linearize_h, dim3(n_blocks), dim3(LIN_THREADS_H), lds_bytes, s, T);
    else if (threads == LIN_THREADS_H) hipLaunchKernelGGL(k_linearize_gh, dim3(n_blocks), dim3(LIN_THREADS_H), lds_bytes, s, T);
    else if (!use_ext) hipLaunchKernelGGL(k_linearize, dim3(n_blocks), dim3(LIN_THREADS), lds_bytes, s, T);
    else hipLaunchKernelGGL(k_linearize_g, dim3(n_blocks), dim3(LIN_THREADS), lds_bytes, s, T);
}
// threads < 0: some window's plan has an extrinsic block (the kernels that read it from the item)
static void launch_linearize_b(const BatchArgs &a, int lm_dim, int max_blocks, int B, size_t lin_lds, int threads, hipStream_t s) {
    const bool ext = threads < 0;
    if (ext) threads = -threads;
    if (lm_dim == 3 && threads == LIN_THREADS_H) hipLaunchKernelGGL(k_linearize_xyz_hb, dim3(max_blocks, B), dim3(LIN_THREADS_H), lin_lds, s, a);
    else if (lm_dim == 3) hipLaunchKernelGGL(k_linearize_xyz_b, dim3(max_blocks, B), dim3(LIN_THREADS), lin_lds, s, a);
    else if (threads == LIN_THREADS_H && !ext) hipLaunchKernelGGL(k_linearize_hb, dim3(max_blocks, B), dim3(LIN_THREADS_H), lin_lds, s, a);
    else if (threads == LIN_THREADS_H) hipLaunchKernelGGL(k_linearize_ghb, dim3(max_blocks, B), dim3(LIN_THREADS_H), lin_lds, s, a);
    else if (!ext) hipLaunchKernelGGL(k_linearize_b, dim3(max_blocks, B), dim3(LIN_THREADS), lin_lds, s, a);
    else hipLaunchKernelGGL(k_linearize_gb, dim3(max_blocks, B), dim3(LIN_THREADS), lin_lds, s, a);
}
// order 1: the chain order for every window of the batch (no rank kernel: every entry's place is static)
static void launch_assemble_b(const BatchArgs &a, int B, int order, hipStream_t s) {
    if (order == 1) { hipLaunchKernelGGL(k_assemble_cb, dim3(ASMC_BLOCKS, B), dim3(ASMC_THREADS), 0, s, a); return; }
    hipLaunchKernelGGL(k_rank_b, dim3(1, B), dim3(ASM_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_assemble_b, dim3(PS_NP + 1, B), dim3(192), 0, s, a);
}
static void launch_pose_solve_b(const BatchArgs &a, int B, size_t ps_lds, int order, hipStream_t s) {
    if (order == 1) hipLaunchKernelGGL(k_pose_solve_cb, dim3(1, B), dim3(PS_THREADS), CH_LDS_DOUBLES * sizeof(double), s, a);
    else hipLaunchKernelGGL(k_pose_solve_b, dim3(1, B), dim3(PS_THREADS), ps_lds, s, a);
}
// batched GN iteration (windows of one landmark kind): grid.y = window
// ev_kernel (VIO_K_* of vio_profile_begin, -1: none): an event pair around that kernel's launch (bench.py's batched roofline)
void vio_launch_batch_gn(const DeviceTables *tabs, int B, int lm_dim, int max_blocks, size_t lin_lds, int lin_threads, int test_prev, int any_prior,
                         int parity, size_t ps_lds, int order, hipStream_t s, int ev_kernel, hipEvent_t *ev) {
    BatchArgs a{tabs, test_prev ? 2 : 0, parity, 0};
    auto mark = [&](int k, int which) { if (ev_kernel == k) (void)hipEventRecord(ev[which], s); };
    mark(0, 0);
    launch_linearize_b(a, lm_dim, max_blocks, B, lin_lds, lin_threads, s);
    mark(0, 1);
    a.gn_flags = test_prev ? 1 : 0;
    if (order == 1) {
        // three launches: the sums, the assembly of the chain image and (k_pose_solve_cb, bit 0) the previous step's test
        mark(1, 0);
        hipLaunchKernelGGL(k_reduce_cb, dim3(VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS + ((test_prev && any_prior) ? RED_ERR_BLOCKS : 0), B), dim3(RED_THREADS), 0, s, a);
        mark(1, 1);
        a.gn_flags = 4 | (test_prev ? 1 : 0);
        mark(3, 0);
        launch_pose_solve_b(a, B, ps_lds, order, s);
        mark(3, 1);
        return;
    }
    mark(1, 0);
    hipLaunchKernelGGL(k_reduce_b, dim3(VIO_NPAIR + VIO_NCB + 1 + ((test_prev && any_prior) ? RED_ERR_BLOCKS : 0), B), dim3(RED_THREADS), 0, s, a);
    mark(1, 1);
    mark(2, 0);
    launch_assemble_b(a, B, order, s);
    mark(2, 1);
    a.gn_flags = 4;
    mark(3, 0);
    launch_pose_solve_b(a, B, ps_lds, order, s);
    mark(3, 1);
}
__global__ __launch_bounds__(RED_THREADS) void k_errprior_b(BatchArgs a);
// Batched LM solve (vio_batch_solve): the kernels of vio_solve's device-driven loop with grid.y = window;
// every window follows its own LmState (parity -1: `cur` from LmState; gate as in the single-window loop).
//   what 0: first linearisation + ComputeLambdaInitLM      what 1: one slot = trial (gate 2) + re-linearisation (gate 3)
void vio_launch_batch_lm(const DeviceTables *tabs, int B, int lm_dim, int max_blocks, size_t lin_lds, int lin_threads, int any_prior, size_t ps_lds,
                         int what, int max_iter, int order, hipStream_t s) {
    auto linearize = [&](int gate) {
        BatchArgs a{tabs, 0, -1, gate};
        launch_linearize_b(a, lm_dim, max_blocks, B, lin_lds, lin_threads, s);
        if (order == 1) { hipLaunchKernelGGL(k_reduce_cb, dim3(VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS, B), dim3(RED_THREADS), 0, s, a); return; }
        hipLaunchKernelGGL(k_reduce_b, dim3(VIO_NPAIR + VIO_NCB + 1, B), dim3(RED_THREADS), 0, s, a);
        launch_assemble_b(a, B, order, s);
    };
    if (what == 0) {
        linearize(0);
        BatchArgs a{tabs, 0, -1, 0};
        hipLaunchKernelGGL(k_init_lm_b, dim3(1, B), dim3(256), 0, s, a, max_iter);
        return;
    }
    if (what >= 2) {
        // vio_solve's loop, batched: what 3 = the first step (k_pose_solve alone); what 2 = one slot: linearise at the step's trial state
        // (its landmark back-substitution first), sum, assemble (chi2 and gain-ratio terms of the step), then k_pose_solve: verdict, next step
        BatchArgs a{tabs, 2, -2, 2};
        if (what == 2) {
            launch_linearize_b(a, lm_dim, max_blocks, B, lin_lds, lin_threads, s);
            a.gn_flags = 1;
            if (order == 1) {
                hipLaunchKernelGGL(k_reduce_cb, dim3(VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS + (any_prior ? RED_ERR_BLOCKS : 0), B), dim3(RED_THREADS), 0, s, a);
                a.gn_flags = 4 | 1;
                launch_pose_solve_b(a, B, ps_lds, order, s);
                return;
            }
            hipLaunchKernelGGL(k_reduce_b, dim3(VIO_NPAIR + VIO_NCB + 1 + (any_prior ? RED_ERR_BLOCKS : 0), B), dim3(RED_THREADS), 0, s, a);
            launch_assemble_b(a, B, order, s);
        }
        a.gn_flags = 4;
        launch_pose_solve_b(a, B, ps_lds, order, s);
        return;
    }
    BatchArgs a{tabs, 4, -1, 2};
    launch_pose_solve_b(a, B, ps_lds, order, s);
    a.gn_flags = 8;
    if (lm_dim == 3) hipLaunchKernelGGL(k_backsub_xyz_b, dim3(max_blocks, B), dim3(BS_THREADS), 0, s, a, 0);
    else hipLaunchKernelGGL(k_backsub_b, dim3(max_blocks, B), dim3(BS_THREADS), 0, s, a, 0);
    a.gn_flags = 0;
    if (any_prior) hipLaunchKernelGGL(k_errprior_b, dim3(RED_ERR_BLOCKS, B), dim3(RED_THREADS), 0, s, a);
    hipLaunchKernelGGL(k_lm_decide_b, dim3(1, B), dim3(256), 0, s, a, 0);
    linearize(3);
}
void vio_launch_reduce(const ReduceTables &R, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3(VIO_NPAIR + VIO_NCB + 1 + (R.errprior ? RED_ERR_BLOCKS : 0)), dim3(RED_THREADS), 0, s, R);
}
// the three-launch path (chain order, unsharded): k_reduce + k_assemble_c in one launch
void vio_launch_reduce_assemble(const ReduceTables &R, const DeviceTables &T, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_c, dim3(VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS + (R.errprior ? RED_ERR_BLOCKS : 0)), dim3(RED_THREADS), 0, s, R, T);
}
// err_prior of a trial step (an LM trial, or a GN step whose test is flushed the classic way): the trial slot, before
// k_lm_decide reads it
__global__ __launch_bounds__(RED_THREADS) void k_errprior(DeviceTables T) {
    if (d_gated_off(T.lm, T.lm_gate)) return;
    const int trial = 1 - d_cur(T);
    const int row = blockIdx.x * (RED_THREADS / 64) + (threadIdx.x >> 6);
    if (row < VIO_PRD) d_errprior_row(T.Jtinv, T.bprior + trial * 176, T.errprior + trial * 160, row, threadIdx.x & 63);
}
__global__ __launch_bounds__(RED_THREADS) void k_errprior_b(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if (!T.has_prior || d_gated_off(T.lm, T.lm_gate)) return;
    const int trial = 1 - d_cur(T);
    const int row = blockIdx.x * (RED_THREADS / 64) + (threadIdx.x >> 6);
    if (row < VIO_PRD) d_errprior_row(T.Jtinv, T.bprior + trial * 176, T.errprior + trial * 160, row, threadIdx.x & 63);
}
void vio_launch_errprior(const DeviceTables &T, hipStream_t s) { hipLaunchKernelGGL(k_errprior, dim3(RED_ERR_BLOCKS), dim3(RED_THREADS), 0, s, T); }
// the landmarks of one plan out of another's, at the current slot of both double buffers (MargOldFrame straight after a solve:
// dst = the landmarks hosted in frame 0, map = their places in the solve plan's order)
// The target observations of a window from the caller's list (raw, (x, y) pairs) into item order (pts_j: obs_base + k G + g, the
// layout k_linearize reads coalesced): one workgroup per item.  first[s]: where sorted landmark s's observations start — in the
// list itself when it is landmark-major (what estimator.cpp:975-1016 emits), else in obs_idx, the list's CSR by landmark.
__global__ __launch_bounds__(256) void k_gather_obs(const ItemDesc *items, const int32_t *first, const int32_t *obs_idx, const double2 *raw, double2 *out) {
    const ItemDesc *it = items + blockIdx.x;
    const int G = it->G, n = G * it->K, s = it->lm_base;
    double2 *o = out + it->obs_base;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int k = t / G, g = t - k * G;
        int src = first[s + g] + k;
        if (obs_idx) src = obs_idx[src];
        o[t] = raw[src];
    }
}
void vio_launch_gather_obs(const ItemDesc *items, int n_items, const int32_t *first, const int32_t *obs_idx, const double *raw, double *out, hipStream_t s) {
    if (n_items > 0) hipLaunchKernelGGL(k_gather_obs, dim3(n_items), dim3(256), 0, s, items, first, obs_idx, (const double2 *)raw, (double2 *)out);
}

__global__ __launch_bounds__(256) void k_gather_landmarks(const LmState *lm, const double *src, int ns_src, double *dst, int ns_dst, const int32_t *map) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= ns_dst) return;
    const int cur = lm->cur;
    dst[(size_t)cur * ns_dst + s] = src[(size_t)cur * ns_src + map[s]];
}
void vio_launch_gather_landmarks(const LmState *lm, const double *src, int ns_src, double *dst, int ns_dst, const int32_t *map, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_landmarks, dim3((ns_dst + 255) / 256), dim3(256), 0, s, lm, src, ns_src, dst, ns_dst, map);
}
void vio_launch_assemble(const DeviceTables &T, hipStream_t s) {
    if (T.solve_order == 1) hipLaunchKernelGGL(k_assemble_c, dim3(ASMC_BLOCKS), dim3(ASMC_THREADS), 0, s, T);
    else hipLaunchKernelGGL(k_assemble, dim3(PS_NP + 1), dim3(ASM_THREADS), 0, s, T);
}
// lds_bytes: what the Eigen-order kernel needs (the chain kernel's size is its own)
void vio_launch_pose_solve(const DeviceTables &T, size_t lds_bytes, hipStream_t s) {
    if (T.solve_order == 1 && (T.gn_flags & 16)) hipLaunchKernelGGL(k_pose_solve_cs, dim3(1), dim3(PS_THREADS), CH_LDS_DOUBLES * sizeof(double), s, T);
    else if (T.solve_order == 1) hipLaunchKernelGGL(k_pose_solve_c, dim3(1), dim3(PS_THREADS), CH_LDS_DOUBLES * sizeof(double), s, T);
    else hipLaunchKernelGGL(k_pose_solve, dim3(1), dim3(PS_THREADS), lds_bytes, s, T);
}
// test entry of the chain solve: one image, one lambda (tests/test_gpu_chain_solve.py)
#ifdef VIO_DEBUG_ENTRY_POINTS
void vio_launch_chain_solve_test(const double *img, double lambda, double *x_nat, double *lds_dump, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_solve_test, dim3(1), dim3(PS_THREADS), CH_LDS_CORE * sizeof(double), s, img, lambda, x_nat, lds_dump);
}
#endif
void vio_launch_prior_simg(const DeviceTables &T, hipStream_t s) {
    (void)hipMemsetAsync(T.prior_simg, 0, (size_t)CH_OFF_CC * sizeof(double), s);
    (void)hipMemsetAsync(T.prior_flags, 0, CPI_FLAGS * sizeof(int32_t), s);
    hipLaunchKernelGGL(k_prior_simg, dim3(99), dim3(192), 0, s, T);
    hipLaunchKernelGGL(k_prior_compact, dim3(1), dim3(PS_THREADS), 0, s, T);
}
int vio_chain_s_doubles() { return CH_OFF_CC; }
int vio_chain_pre_lds_doubles() { return CPI_END; }
int vio_chain_prior_flags() { return CPI_FLAGS; }
// where the elements of an IMU item's 3 x 3 tiles go in the chain image: [10][63][9] (p1 | p2 << 16; p2 0xffff: no mirror; p1 CH_OFF_X: nowhere).  An element (a, b) of
// the vertex blocks on or above the diagonal feeds entry (max, min) of the pair; inside a diagonal vertex block only a >= b is read
// (d_hs_rest: "upper vertex blocks are computed, lower ones mirrored", problem.cc:347-355)
void vio_chain_imu_map(uint32_t *out) {
    for (int k = 0; k < 10; ++k)
        for (int tau = 0; tau < 63; ++tau) {
            int ca, cb;
            cpi_tile(tau, ca, cb);
            for (int u = 0; u < 3; ++u)
                for (int v = 0; v < 3; ++v) {
                    const int a = 3 * ca + u, b = 3 * cb + v;
                    uint32_t m = 0xffff0000u | (uint32_t)CH_OFF_X;      // nowhere: the dummy position (the solution's place, unused by the chain workgroup)
                    const bool use = cpi_vb(ca) < cpi_vb(cb) || a >= b;
                    if (use) {
                        const int I = 6 + 15 * k + (a > b ? a : b), J = 6 + 15 * k + (a > b ? b : a);
                        int p1, p2;
                        ch_entry_pos(I, J, p1, p2);
                        const bool s_row = (I - 6) % 15 >= 6 || (J - 6) % 15 >= 6;       // a speed-bias variable is involved: an entry of the chain's blocks
                        if (s_row && p1 >= 0 && p1 < CH_OFF_CC) m = (uint32_t)p1 | ((p2 >= 0 ? (uint32_t)p2 : 0xffffu) << 16);
                    }
                    out[(k * 63 + tau) * 9 + 3 * u + v] = m;
                }
        }
}
void vio_launch_chain_pre(const DeviceTables &T, hipStream_t s) { hipLaunchKernelGGL(k_chain_pre, dim3(1), dim3(PS_THREADS), CH_LDS_CORE * sizeof(double), s, T); }
int vio_chain_image_doubles() { return CH_PACKED; }
int vio_chain_y_offset() { return CH_OFF_Y; }
int vio_chain_lds_core_doubles() { return CH_LDS_CORE; }
void vio_chain_entry_pos(int i, int j, int *p1, int *p2) { ch_entry_pos(i, j, *p1, *p2); }
int vio_chain_dim(int i) { return ch_dim(i); }
void vio_launch_backsub(const DeviceTables &T, int mode, hipStream_t s) {
    if (T.lm_dim == 3) hipLaunchKernelGGL(k_backsub_xyz, dim3(T.n_items + T.n_imu_items), dim3(BS_THREADS), 0, s, T, mode);
    else hipLaunchKernelGGL(k_backsub, dim3(T.n_items + T.n_imu_items), dim3(BS_THREADS), 0, s, T, mode);
}
void vio_launch_step_sum(const DeviceTables &T, int mode, hipStream_t s) { hipLaunchKernelGGL(k_step_sum, dim3(1), dim3(256), 0, s, T, mode); }
void vio_launch_lm_decide(const DeviceTables &T, int mode, int sum_local, hipStream_t s) {
    hipLaunchKernelGGL(k_lm_decide, dim3(1), dim3(256), 0, s, T, mode, sum_local);
}
__global__ void k_set_lambda(LmState *lm, double lambda) {
    if (threadIdx.x == 0) lm->lambda = lambda;
}
void vio_launch_set_lambda(LmState *lm, double lambda, hipStream_t s) { hipLaunchKernelGGL(k_set_lambda, dim3(1), dim3(64), 0, s, lm, lambda); }
// rollback after a rejected step is implicit (the trial copies are simply not made current); an explicit
// RollbackStates after vio_update_states flips the current index back
__global__ void k_flip(LmState *lm) {
    if (threadIdx.x == 0) lm->cur ^= 1;
}
void vio_launch_flip(LmState *lm, hipStream_t s) { hipLaunchKernelGGL(k_flip, dim3(1), dim3(64), 0, s, lm); }
void vio_launch_init_lm(const DeviceTables &T, int max_iter, hipStream_t s) {
    hipLaunchKernelGGL(k_init_lm, dim3(1), dim3(256), 0, s, T, max_iter);
}
int lin_threads_host() { return LIN_THREADS; }
int lin_threads_half_host() { return LIN_THREADS_H; }
int lin_lds_doubles_host(int G, int K, int nb, int use_ext) {
    return lin_lds_doubles(G, K, nb, use_ext);
}
int xyz_lds_doubles_host(int G, int K) { return xyz_lds_doubles(G, K); }
int vio_set_kernel_attributes() {
    hipError_t e1 = hipFuncSetAttribute((const void *)k_linearize, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    hipError_t e2 = hipFuncSetAttribute((const void *)k_pose_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (hipFuncSetAttribute((const void *)k_linearize_b, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_h, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_hb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_g, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_gh, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_gb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_linearize_ghb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_pose_solve_b, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_pose_solve_c, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_pose_solve_cb, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_pose_solve_cs, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void *)k_chain_pre, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
#ifdef VIO_DEBUG_ENTRY_POINTS
    if (hipFuncSetAttribute((const void *)k_chain_solve_test, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return -1;
#endif
    hipError_t e3 = hipFuncSetAttribute((const void *)k_linearize_xyz, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e3 == hipSuccess) e3 = hipFuncSetAttribute((const void *)k_linearize_xyz_b, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e3 == hipSuccess) e3 = hipFuncSetAttribute((const void *)k_linearize_xyz_h, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e3 == hipSuccess) e3 = hipFuncSetAttribute((const void *)k_linearize_xyz_hb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    return (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? 0 : -1;
}
