// vio_pose_solve_chain.h — the structure-exploiting pose solve ("chain order"), included by vio_kernels.hip.
//
// (H_pp_schur_ + lambda I) dx = b_pp_schur_ (VM/src/backend/problem.cc:434-439) solved by an UNPIVOTED blocked LDL^T in a STATIC
// elimination order that follows the structure of the reduced system instead of Eigen's sort of the diagonal:
//
//   1. the 11 speed-bias blocks S_f (9 variables each, inside a block in the order bg, ba, v) — a block-tridiagonal chain
//      (the IMU factor f couples S_f with S_f+1 and with the poses f, f+1; the marginalisation prior adds S_0 <-> everything
//      in the camera block, never S_f <-> S_g for |f - g| > 1) — eliminated from BOTH ENDS towards the middle
//      ("twisted" order 0,10,1,9,2,8,3,7,4,6,5): two independent chains of 9-pivot factorisations, 6 levels deep instead of 11,
//      no fill inside the chain;
//   2. the camera block C = [pose 0 .. pose 10 | ext] (72 variables, 5 tiles of 16: 16,16,16,16,8), dense, blocked as before.
//
// Accuracy (tests/golden/ldlt.npz against the 50-digit solutions of ldlt_exact.npz, tools/chain_solve_model.py): this order with
// true divisions is 4.7e-14 / 1.4e-9 / 4.2e-5 from the exact solution at lambda = 5e5 / 1e3 / 1, where Eigen's pivoted LDLT is
// 4.1e-12 / 1.2e-7 / 1.4e-4.  The divisions matter: the bias random walk makes blocks [[W, -W], [-W, W]] with W ~ 1e16, and
// l = u / d comes out as exactly -1 only when it is a correctly rounded quotient (a reciprocal-multiply leaves eps * 1e16 in the
// Schur complement, 40x the error).  Hence: the tiles keep L (scaled columns, l = u / d by d_div), updates are formed as
// (L D) L^T.
//
// Storage (doubles; the image k_assemble_chain writes to T.Pg and this kernel copies into LDS):
//   SC[e][t]   16 x 9 (row stride 11)   camera tile t (rows) x speed-bias block e (columns)          55 tiles
//   SO[e]       9 x 9 (row stride 11)   succ(e) (rows) x e (columns), succ(e) = e + 1 (e < 5), e - 1 (e > 5)   10 tiles
//   SD[e]       9 x 9 (row stride 11)   diagonal block of S_e                                         11 tiles
//   CC[I][J]   16 x 16 (row stride 17)  camera tiles, I >= J                                          15 tiles
//   Y          yS[11][16] | yC[80]      right-hand side in "chain dimension" order
// Row stride 11 (odd) keeps both MFMA operand images of a 9-column tile free of LDS bank conflicts: a quarter-wave of an A-image
// read is 16 rows of one column, 11 r mod 16 distinct (stride 10 put rows r and r + 8 into the same banks: measured, round 4);
// a C-image access is 16 consecutive columns of one row.  Columns 9 and 10 are padding: zero in the image, never written.
#ifndef VIO_POSE_SOLVE_CHAIN_H
#define VIO_POSE_SOLVE_CHAIN_H

#define CH_NS 11
#define CH_TS 11            // row stride of a 9-column tile: odd, so that the 16 rows of an operand image fall into 16 different bank pairs
#define CH_SCSZ (16 * CH_TS)
#define CH_S9SZ (9 * CH_TS)
#define CH_OFF_SC 0
#define CH_OFF_SO (CH_OFF_SC + CH_NS * 5 * CH_SCSZ)        // 8800
#define CH_OFF_SD (CH_OFF_SO + 10 * CH_S9SZ)                // 9700
#define CH_OFF_CC (CH_OFF_SD + CH_NS * CH_S9SZ)             // 10690
#define CH_OFF_Y (CH_OFF_CC + 15 * PS_TS)                   // 14770
#define CH_YC 176                                           // dimension index of camera variable 0
#define CH_NDIM 256
#define CH_PACKED ((CH_OFF_Y + CH_NDIM + 1) & ~1)             // what travels through HBM (an even count: copied as double2)
#define CH_SET_STRIDE 16128
#define CH_OFF_SM CH_PACKED                                 // LDS only: M_e = L_ee^-T of every speed-bias block, 9 x 9 (stride 11)
#define CH_OFF_MC (CH_OFF_SM + CH_NS * CH_S9SZ)             // 16 x 17: M_K of the camera tile being factored
#define CH_OFF_D (CH_OFF_MC + PS_TS)                        // [256] pivots by dimension
#define CH_OFF_X (CH_OFF_D + CH_NDIM)                       // [256] solution by dimension
#define CH_OFF_I9 (CH_OFF_X + CH_NDIM)                      // 9 x 9 identity (stride 10): the rows F's second half-row of lanes starts from
#define CH_OFF_I16 (CH_OFF_I9 + CH_S9SZ)                    // 16 x 16 identity (stride 17)
#define CH_OFF_NZ (CH_OFF_I16 + PS_TS)                      // 64 ints: which SC tiles of the image hold a non-zero
#define CH_LDS_CORE (CH_OFF_NZ + 32)
static_assert(CH_NS * CH_S9SZ >= CH_NS * 64, "the back-substitution's scratch (11 x 64) lives in the SD tiles");
static_assert(CH_PACKED % 2 == 0 && CH_SET_STRIDE >= CH_PACKED && CH_SET_STRIDE <= PS_SET_STRIDE, "chain image fits a set of Pg");

__host__ __device__ inline int ch_sc(int e, int t) { return CH_OFF_SC + (e * 5 + t) * CH_SCSZ; }
__host__ __device__ inline int ch_so(int e) { return CH_OFF_SO + (e < 5 ? e : e - 1) * CH_S9SZ; }
__host__ __device__ inline int ch_sd(int e) { return CH_OFF_SD + e * CH_S9SZ; }
__host__ __device__ inline int ch_sm(int e) { return CH_OFF_SM + e * CH_S9SZ; }
__host__ __device__ inline int ch_tix(int I, int J) { return (I * (I + 1) / 2 + J) * (16 * 17); }
__host__ __device__ inline int ch_cc(int I, int J) { return CH_OFF_CC + ch_tix(I, J); }
__host__ __device__ inline int ch_cc_elem(int r, int c) { return ch_cc(r >> 4, c >> 4) + (r & 15) * 17 + (c & 15); }
// natural index of H_pp_schur_ (0 .. 170: ext | (pose, v, ba, bg) x 11) -> chain dimension
__host__ __device__ inline int ch_dim(int i) {
    if (i < 6) return CH_YC + 66 + i;
    const int f = (i - 6) / 15, r = (i - 6) % 15;
    if (r < 6) return CH_YC + 6 * f + r;
    const int c = r - 6;                                    // v 0..2, ba 3..5, bg 6..8  ->  bg 0..2, ba 3..5, v 6..8
    return f * 16 + (c < 3 ? 6 + c : (c < 6 ? c : c - 6));
}
// where entry (i, j) of the symmetric matrix lives in the image: p1 (and p2, the mirror image inside a diagonal block), or -1 when
// the pair has no storage (speed-bias blocks that are not neighbours: must be zero, checked by the host before this path is chosen)
__host__ __device__ inline void ch_entry_pos(int i, int j, int &p1, int &p2) {
    const int di = ch_dim(i), dj = ch_dim(j);
    p1 = -1; p2 = -1;
    const bool ci = di >= CH_YC, cj = dj >= CH_YC;
    if (ci && cj) {
        int a = di - CH_YC, b = dj - CH_YC;
        if (a < b) { const int t = a; a = b; b = t; }
        p1 = ch_cc_elem(a, b);
        if (a != b && (a >> 4) == (b >> 4)) p2 = ch_cc_elem(b, a);
    } else if (ci != cj) {
        const int c = (ci ? di : dj) - CH_YC, s = ci ? dj : di;
        p1 = ch_sc(s >> 4, c >> 4) + (c & 15) * CH_TS + (s & 15);
    } else {
        const int ei = di >> 4, ki = di & 15, ej = dj >> 4, kj = dj & 15;
        if (ei == ej) {
            p1 = ch_sd(ei) + ki * CH_TS + kj;
            if (ki != kj) p2 = ch_sd(ei) + kj * CH_TS + ki;
        } else if (ei - ej == 1 || ej - ei == 1) {
            // the column block is the one eliminated first: the one farther from block 5
            const int ai = ei > 5 ? ei - 5 : 5 - ei, aj = ej > 5 ? ej - 5 : 5 - ej;
            if (ai > aj) p1 = ch_so(ei) + kj * CH_TS + ki;      // rows: block ej, columns: block ei
            else p1 = ch_so(ej) + ki * CH_TS + kj;
        }
    }
}

#ifdef __HIPCC__
#ifndef CH_TRUE_DIV
#define CH_TRUE_DIV 1
#endif
// a / d with r = d_fast_rcp(d): one correction step makes the quotient correctly rounded in all but rare cases, and exact
// whenever a / d is representable (the -1 of the random-walk blocks); 0 for d == 0
__device__ __forceinline__ double d_div(double a, double d, double r) {
#if CH_TRUE_DIV
    const double q = a * r;
    const double rem = fma(-d, q, a);
    return fma(rem, r, q);
#else
    return a * r;
#endif
}

// F: factor the NP x NP diagonal block `tile` (row stride TS) in place, one wave; ps_factor_diag's scheme (see there) with true
// divisions and a pivot count: lanes 16..31 carry the rows of the identity (read from sI, row stride TS) and end up with M = L^-T
// (written to M, row stride MS, entries [r][c] for r, c < NP only).  After it: tile[TS j + c] = U(c, j) = L(c, j) d_j for c >= j
// (the pivots on the diagonal).
template <int NP, int TS, int MS>
__device__ __noinline__ void ch_factor(lds_double *tile, lds_double *sI, lds_double *M, int lane) {
    asm volatile("" : "+v"(tile));
    const bool ident = (lane >> 4) == 1;
    const int row = min(lane & 15, NP - 1);           // lanes past the block repeat its last row (identical stores)
    lds_double *p0 = (ident ? sI : tile) + row * TS;
    double a0[NP], u[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) a0[j] = p0[j];
    lds_double *wp = ident ? M + row * MS : tile + row;
    const int ws = ident ? 1 : TS;
    __builtin_amdgcn_sched_barrier(0);   // every row is in registers before the first publish overwrites the tile
    wp[0] = a0[0];
    double d = d_readlane(a0[0], 0);
#pragma unroll
    for (int c = 1; c < NP; ++c) u[c] = tile[c];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        const double l0 = d_div(a0[j], d, d_fast_rcp(d));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
        if (j + 1 < NP) {
            a0[j + 1] = fma(-l0, u[j + 1], a0[j + 1]);
            wp += ws;
            wp[0] = a0[j + 1];
            d = d_readlane(a0[j + 1], j + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = j + 2; c < NP; ++c) {
            a0[c] = fma(-l0, u[c], a0[c]);
            if (((c - j - 2) & 3) == 3 || c + 1 == NP) {
#pragma unroll
                for (int e = c - ((c - j - 2) & 3); e <= c; ++e) u[e] = tile[(j + 1) * TS + e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// out = init + sum_{j < 9} arr[j] v_j, v_j from lane j of the 16-lane row (DPP row_newbcast), two chains
#define CH_DOT9(out, init, vin, arr) do { double x__ = (init), x2__ = 0.0; const double v__ = (vin); \
            asm volatile("s_nop 1\n\t" \
                         "v_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %1, %2, %10 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
                         "v_fmac_f64_dpp %0, %2, %11 row_newbcast:8 row_mask:0xf bank_mask:0xf" \
                         : "+v"(x__), "+v"(x2__) \
                         : "v"(v__), "v"(arr[0]), "v"(arr[1]), "v"(arr[2]), "v"(arr[3]), "v"(arr[4]), "v"(arr[5]), "v"(arr[6]), "v"(arr[7]), "v"(arr[8])); \
            (out) = x__ + x2__; } while (0)

#include "vio_chain_core.h"

// ---------------------------------------------------------------------------------------------------------
// k_assemble_c: H_pp_schur_ (reduced visual system + IMU blocks + prior, problem.cc:365-384) written straight into the chain
// image: every entry's place is a function of its indices (ch_entry_pos), no rank sort, no permutation.  Workgroup b < 171 owns
// natural row b (entries (b, t), t <= b); workgroup 171 the right-hand sides and, in the GN / LM loops, the step test's sums.
// ---------------------------------------------------------------------------------------------------------
#define ASMC_THREADS 192
#define ASMC_BLOCKS (VIO_PD + 1)
__device__ __forceinline__ void d_assemble_chain_body(const DeviceTables &T) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (d_gated_off(T.lm, T.lm_gate)) return;
    const int cur = d_cur(T);
    const int valid = d_imu_mask(T);
    const int wset = d_set_w(T);
    double *Pg = T.Pg + wset * CH_SET_STRIDE;
    if (b < VIO_PD) {
        if (t < VIO_PD && (t <= b || T.natural_hs)) {
            double wv, wr;
            d_hs_entry(T, valid, max(b, t), min(b, t), wv, wr);
            const double v = wv + wr;
            if (T.natural_hs) T.Hs[b * VIO_PD + t] = v;
            if (t <= b) {
                int p1, p2;
                ch_entry_pos(b, t, p1, p2);
                if (p1 >= 0) Pg[p1] = v;
                if (p2 >= 0) Pg[p2] = v;
            }
        }
        return;
    }
    // the last workgroup: right-hand sides + the step test of the PREVIOUS iteration (see d_assemble_body)
    __shared__ double sSum[8];
    const bool test_prev = d_step_owed(T, 1);
    const int rset = d_set_r(T);
    double p_dx = 0.0, p_bf = 0.0, p_er = 0.0, p_lambda = 0.0, p_chi = 0.0, p_step = 0.0, p_lmchi = 0.0, p_imu[10];
    if (test_prev) {
        p_lambda = T.lm->lambda;
        if (t < VIO_PD) { p_dx = T.dx[t]; p_bf = T.bfull[rset * 176 + t]; }
        if (T.has_prior && t < VIO_PRD) p_er = T.errprior[cur * 160 + t];
        if (t == 0) {
            p_chi = d_vis(T, VIS_CHI); p_step = d_vis(T, VIS_STEP + 1); p_lmchi = T.lm->chi;
#pragma unroll
            for (int k = 0; k < 10; ++k) p_imu[k] = ((valid >> k) & 1) ? T.imu_out[k * IMU_OUT + IMU_CHI] : 0.0;
        }
    }
    if (t < VIO_PD) Pg[CH_OFF_Y + ch_dim(t)] = d_rhs_entries(T, valid, t, cur, wset);
    if (test_prev) {
        double sp = 0.0, e2 = 0.0;
        if (t < VIO_PD) sp = p_dx * (p_lambda * p_dx + p_bf);
        if (T.has_prior && t < VIO_PRD) e2 = p_er * p_er;
        d_block_sum2<ASMC_THREADS>(sp, e2, sSum, t);
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
}
__global__ __launch_bounds__(ASMC_THREADS) void k_assemble_c(DeviceTables T) { d_assemble_chain_body(T); }

// ---- the three-launch path: k_reduce_c = k_reduce + k_assemble_c (hooks of d_reduce_body<true>) ----
__device__ __forceinline__ void d_fused_store(const DeviceTables &T, int i, int j, double v) {      // entry (i, j), i >= j, into the image
    double *Pg = T.Pg + d_set_w(T) * CH_SET_STRIDE;
    int p1, p2;
    ch_entry_pos(i, j, p1, p2);
    if (p1 >= 0) Pg[p1] = v;
    if (p2 >= 0) Pg[p2] = v;
}
__device__ void d_fused_pair(const DeviceTables &T, int b, int tid, double tot) {
    // block b = VIS_PAIR(P, Q), P <= Q; thread = entry (a, bq) of the 6 x 6 block.  A diagonal block holds both halves: the lower one is
    // what d_hs_entry reads
    int P = 0;
    while (VIS_PAIR(P + 1, P + 1) <= b) ++P;
    const int Q = P + (b - VIS_PAIR(P, P));
    const int a = tid / 6, bq = tid - 6 * a;
    if (P == Q && a < bq) return;
    const int i = cam_to_full(6 * P + a), j = cam_to_full(6 * Q + bq);
    const int I = max(i, j), J = min(i, j);
    d_fused_store(T, I, J, tot + d_hs_rest(T, d_imu_mask(T), I, J));
}
__device__ void d_fused_vec(const DeviceTables &T, int P, int tid, double bd, double bc, double dg) {
    const int valid = d_imu_mask(T), cur = d_cur(T), wset = d_set_w(T);
    const int i = cam_to_full(6 * P + tid);
    const double extra = d_rhs_rest(T, valid, i, cur);
    const double bred = bd - bc;
    T.bs[i] = bred + extra;
    T.bfull[wset * 176 + i] = bd + extra;
    T.diagfull[i] = dg + d_hs_rest(T, valid, i, i);
    T.Pg[wset * CH_SET_STRIDE + CH_OFF_Y + ch_dim(i)] = bred + extra;
}
__device__ void d_fused_sb_row(const DeviceTables &T, int blk, int tid) {
    // natural row i of speed-bias variable r (frame r / 9, component r % 9), five rows per workgroup: the entries (i, t) towards every
    // camera variable and towards the speed-bias variables t <= i; no visual part
    const int r = 5 * blk + tid / VIO_PD, t = tid % VIO_PD;
    if (tid >= 5 * VIO_PD || r >= 99) return;
    const int valid = d_imu_mask(T), cur = d_cur(T), wset = d_set_w(T);
    const int i = 12 + 15 * (r / 9) + r % 9;
    if (full_to_cam(t) >= 0 || t <= i) {
        const int I = max(i, t), J = min(i, t);
        d_fused_store(T, I, J, d_hs_rest(T, valid, I, J));
    }
    if (t == i) {
        const double extra = d_rhs_rest(T, valid, i, cur);
        T.bs[i] = extra;
        T.bfull[wset * 176 + i] = extra;
        T.diagfull[i] = d_hs_rest(T, valid, i, i);
        T.Pg[wset * CH_SET_STRIDE + CH_OFF_Y + ch_dim(i)] = extra;
    }
}
__global__ __launch_bounds__(RED_THREADS) void k_reduce_c(ReduceTables R, DeviceTables T) { d_reduce_body<true>(R, &T); }
__global__ __launch_bounds__(RED_THREADS) void k_reduce_cb(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    const bool test_prev = (a.gn_flags & 1) != 0, err_prev = test_prev && T.has_prior;
    if ((int)blockIdx.x >= VIO_NPAIR + VIO_NCB + 1 + RED_SB_BLOCKS && !err_prev) return;
    const int loop = T.cur_hint == -2, cur = loop ? 0 : T.cur_hint;
    ReduceTables R{T.list_off, T.list, T.slab, T.vis, test_prev ? T.step_part : nullptr, T.n_items, a.gate, T.lm,
                   err_prev ? T.Jtinv : nullptr, err_prev ? T.bprior + cur * 176 : nullptr, err_prev ? T.errprior + cur * 160 : nullptr, loop};
    d_reduce_body<true>(R, &T);
}
__global__ __launch_bounds__(ASMC_THREADS) void k_assemble_cb(BatchArgs a) { const DeviceTables T = d_batch_tables(a); d_assemble_chain_body(T); }

// ---------------------------------------------------------------------------------------------------------
// k_pose_solve_c: k_pose_solve's frame (the verdict of vio_solve's loop, the two sets, the trial states, the pair table, the
// stepwise prior update) around ch_factor_solve.
// ---------------------------------------------------------------------------------------------------------
#define CH_OFF_DX CH_LDS_CORE                       // 176 solution in natural order
#define CH_OFF_R (CH_OFF_DX + 176)                  // 112 rotations
#define CH_OFF_B (CH_OFF_R + 112)                   // 176 trial b_prior
#define CH_OFF_STATE (CH_OFF_B + 176)               // 184
#define CH_LDS_DOUBLES (CH_OFF_STATE + 184)         // 17448
__device__ __forceinline__ void d_pose_solve_chain_body(const DeviceTables &T) {
    double *P = dyn_smem;
    double *sX = P + CH_OFF_X, *sDx = P + CH_OFF_DX, *sR = P + CH_OFF_R, *sB = P + CH_OFF_B, *sState = P + CH_OFF_STATE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LmState *lm = T.lm;
    __shared__ LmRegs sLm;
    if (d_gated_off(lm, T.lm_gate)) return;
    const bool lm_loop = T.cur_hint == -2;
    int cur = lm_loop ? 0 : d_cur(T);
    double lambda = lm_loop ? 0.0 : lm->lambda;
    const int n = PS_N;
#ifdef VIO_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#define CH_OUT(slot) do { if (tid == 0 && T.dbg) T.dbg[slot] = __builtin_amdgcn_s_memtime() - t_start; } while (0)
#else
#define CH_OUT(slot) do { } while (0)
#endif
    // the image k_assemble_c wrote (vio_solve's loop: the set of the trial state first; a rejected step solves the other set again)
    int set = lm_loop ? (lm->sys ^ lm->pending) : 0;
    for (int pass = 0; pass < 2; ++pass) {
        const double2 *src = reinterpret_cast<const double2 *>(T.Pg + set * CH_SET_STRIDE);
        double2 *dst = reinterpret_cast<double2 *>(P);
        static_assert((CH_PACKED / 2 + PS_THREADS - 1) / PS_THREADS == 8, "copy below is written for 8 rounds");
#define CH_LD(q) const double2 v##q = src[min(tid + q * PS_THREADS, CH_PACKED / 2 - 1)];
#define CH_ST(q) dst[min(tid + q * PS_THREADS, CH_PACKED / 2 - 1)] = v##q;
        CH_LD(0) CH_LD(1) CH_LD(2) CH_LD(3) CH_LD(4) CH_LD(5) CH_LD(6) CH_LD(7)
        if (pass == 0) {
            double stv = (tid < STATE_STRIDE) ? T.state[cur * STATE_STRIDE + tid] : 0.0;
            const double stv1 = (lm_loop && tid < STATE_STRIDE) ? T.state[STATE_STRIDE + tid] : 0.0;
            // Three-launch path (gn_flags bit 0): the test of the PREVIOUS step, which k_assemble's last workgroup runs in the four-launch
            // path, is formed here: chi2 of the state that step led to (this linearisation's, summed by k_reduce_c, + IMU + ||err_prior||)
            // and the gain ratio's denominator (the landmark part from k_reduce_c, the pose part left by the previous k_pose_solve_c
            // in sp_part).  The sums are k_assemble's: three wave partials added in wave order.
            const bool test_here = d_step_owed(T, 1);
            double t_chi = 0.0, t_step = 0.0, t_lmchi = 0.0, t_imu = 0.0, t_sp = 0.0;
            if (test_here) {
                const int lc = lm_loop ? (lm->cur ^ lm->pending) : cur;          // the copy this linearisation was made at
                double e2 = 0.0;
                if (T.has_prior && tid < VIO_PRD) { const double er = T.errprior[lc * 160 + tid]; e2 = er * er; }
                if (tid < 192) { const double w = d_wave_sum_to_lane63(e2); if (lane == 63) sB[tid >> 6] = w; }
                if (tid == 0) {
                    const int valid = d_imu_mask(T);
                    t_chi = T.vis[VIS_CHI]; t_step = T.vis[VIS_STEP + 1]; t_lmchi = lm->chi;
#pragma unroll
                    for (int k = 0; k < 10; ++k) if ((valid >> k) & 1) t_imu += T.imu_out[k * IMU_OUT + IMU_CHI];
                    t_sp = (T.sp_part[0] + T.sp_part[1]) + T.sp_part[2];
                }
            }
            if (lm_loop && tid == 0 && !test_here) {
                int go = 1, rej = 0, sys = lm->sys;
                d_lm_load(lm, sLm);
                if (lm->pending) {
                    d_lm_verdict(sLm, lm, 0, lm->chi_try, lm->scale, sLm.cur);
                    if (sLm.accepted) sys ^= 1; else rej = 1;
                    go = !sLm.stop;
                }
                sX[0] = go ? 1.0 : 0.0; sX[1] = (double)sLm.cur; sX[2] = sLm.lambda; sX[3] = (double)sys; sX[4] = (double)rej;
            }
            CH_ST(0) CH_ST(1) CH_ST(2) CH_ST(3) CH_ST(4) CH_ST(5) CH_ST(6) CH_ST(7)
            __syncthreads();
            if (test_here) {
                if (tid == 0) {
                    double total = t_chi + t_imu;
                    if (T.has_prior) total += sqrt((sB[0] + sB[1]) + sB[2]);       // err_prior_.norm(), not squared (problem.cc:554-556)
                    const double tempChi = 0.5 * total;
                    const double scale = 0.5 * (t_step + t_sp) + 1e-6;
                    lm->chi_try = tempChi;
                    lm->scale = scale;
                    if (!lm_loop) {
                        lm->rho = (t_lmchi - tempChi) / scale;
                        lm->trials += 1;
                        lm->chi = tempChi;
                        lm->cur = cur;
                        lm->accepted = 1;
                        lm->naccepted += 1;
                        lm->need_linearize = 1;
                        lm->false_cnt = 0;
                        if (!isfinite(tempChi)) lm->finite = 0;
                    } else {
                        int go = 1, rej = 0, sys = lm->sys;
                        d_lm_load(lm, sLm);
                        if (lm->pending) {
                            d_lm_verdict(sLm, lm, 0, tempChi, scale, sLm.cur);
                            if (sLm.accepted) sys ^= 1; else rej = 1;
                            go = !sLm.stop;
                        }
                        sX[0] = go ? 1.0 : 0.0; sX[1] = (double)sLm.cur; sX[2] = sLm.lambda; sX[3] = (double)sys; sX[4] = (double)rej;
                    }
                }
                if (lm_loop) __syncthreads();
            }
            int set_now = 0;
            if (lm_loop) {
                if (tid == 0) { d_lm_store(lm, sLm); lm->sys = (int)sX[3]; lm->pending = sX[0] != 0.0 ? 1 : 0; }
                if (sX[0] == 0.0) return;
                cur = (int)sX[1]; lambda = sX[2]; set_now = (int)sX[3];
                if (cur) stv = stv1;
            }
            if (tid < STATE_STRIDE) sState[tid] = stv;
            if (set_now == set) break;
            set = set_now;
            __syncthreads();
        } else {
            CH_ST(0) CH_ST(1) CH_ST(2) CH_ST(3) CH_ST(4) CH_ST(5) CH_ST(6) CH_ST(7)
            __syncthreads();
        }
#undef CH_LD
#undef CH_ST
    }
    const int trial = cur ^ 1;
    // lambda on the 171 pivots (problem.cc:434-436); the 8 padding variables of the last camera tile are identity rows
    if (tid < n) {
        const int d = ch_dim(tid);
        if (d >= CH_YC) P[ch_cc_elem(d - CH_YC, d - CH_YC)] += lambda;
        else P[ch_sd(d >> 4) + (d & 15) * (CH_TS + 1)] += lambda;
    } else if (tid < n + 8) {
        P[ch_cc_elem(72 + tid - n, 72 + tid - n)] = 1.0;
    } else if (tid >= 256 && tid < 256 + CH_S9SZ + PS_TS) {
        const int i = tid - 256;
        P[CH_OFF_I9 + i] = (i < CH_S9SZ) ? ((i / CH_TS == i % CH_TS) ? 1.0 : 0.0) : (((i - CH_S9SZ) / PS_TROW == (i - CH_S9SZ) % PS_TROW) ? 1.0 : 0.0);
    } else if (tid >= 640 && tid < 640 + CH_NDIM) {
        P[CH_OFF_D + tid - 640] = 1.0;                  // pivots of padding dimensions
    }
    for (int i = tid; i < CH_NS * CH_S9SZ; i += PS_THREADS) P[CH_OFF_SM + i] = 0.0;      // (the padding column of every M_e must read as zero)
    __syncthreads();
    CH_OUT(0);
    // The camera part of the step (poses, extrinsic) is known two phases before the speed-bias part: wave 13 forms the trial poses
    // (UpdateStates, problem.cc:456-463; vertex_pose.cc:7-19) and their rotations while the speed-bias owners finish their sums, and the
    // fourteen waves that do not walk the chains form the pair table of the trial states meanwhile (in the camera tiles' space, free once
    // the camera part is solved).  The stepwise path (prior update here) keeps k_pose_solve's serial tail.
    const bool prior_here = T.has_prior && !(T.gn_flags & 4);
    const int lm_dim = T.lm_dim;
    double *pairtab_trial = T.pairtab + trial * PAIRTAB_STRIDE;
    double *pw = P + CH_OFF_CC;                                 // trial ext + poses at [0, 84), rotations at [96, 204)
    auto mid1 = [&](int ln) {
        if (prior_here) return;
        if (ln < 12) {
            const double *p = (ln == 0) ? sState + STATE_EXT : sState + STATE_POSE + 7 * (ln - 1);
            const double *d = (ln == 0) ? sX + CH_YC + 66 : sX + CH_YC + 6 * (ln - 1);
            double dd[6], tmp[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) dd[k] = d[k];
            d_pose_plus(p, dd, tmp);
            double *q = (ln == 0) ? pw + STATE_EXT : pw + STATE_POSE + 7 * (ln - 1);
#pragma unroll
            for (int k = 0; k < 7; ++k) q[k] = tmp[k];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        if (lm_dim != 3) d_pair_rotations(pw, pw + 96, ln);
    };
    auto mid2 = [&](int idx, int ln) {
        if (prior_here || lm_dim == 3) return;
        d_pair_rows(pw, pairtab_trial, pw + 96, ln + 64 * idx, 14 * 64);
    };
    ch_factor_solve(P, tid, mid1, mid2, T.dbg);
    CH_OUT(2);
    for (int i = tid; i < n; i += PS_THREADS) sDx[i] = sX[ch_dim(i)];
    if (tid < 192 && T.sp_part) {
        // sum_i dx_i (lambda dx_i + b_i), b_ of the system just solved: three wave partials, as k_assemble's step test sums them
        const double dxi = (tid < n) ? sX[ch_dim(min(tid, n - 1))] : 0.0;
        const double bi = (tid < n) ? T.bfull[set * 176 + tid] : 0.0;
        const double w = d_wave_sum_to_lane63((tid < n) ? dxi * (lambda * dxi + bi) : 0.0);
        if (lane == 63) T.sp_part[tid >> 6] = w;
    }
    if (!prior_here) {
        // the trial states: poses from wave 13's copy, speed-bias = current + dx
        if (tid >= 128 && tid < 128 + 84) sState[tid - 128] = P[CH_OFF_CC + tid - 128];
        else if (tid >= 256 && tid < 256 + 99) { const int e = tid - 256; sState[STATE_SB + e] += sX[(e / 9) * 16 + (e % 9 < 3 ? 6 + e % 9 : (e % 9 < 6 ? e % 9 : e % 9 - 6))]; }
        __syncthreads();
        if (tid >= 192 && tid < 192 + n) T.dx[tid - 192] = sDx[tid - 192];
        if (tid >= 384 && tid < 384 + STATE_STRIDE) T.state[trial * STATE_STRIDE + (tid - 384)] = sState[tid - 384];
        CH_OUT(3);
        return;
    }
    __syncthreads();

    // the stepwise path: k_pose_solve's tail (see there)
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
        if (T.lm_dim != 3) d_pair_rotations(sState, sR, tid);
        __syncthreads();
    } else {
        {
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
#pragma unroll
        for (int r = 0; r < PS_JT_ROWS; ++r) {
            const int i = (uwave - 2) + 14 * r;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int j = lane + 64 * q;
                jp[r][q] = (i < VIO_PRD && j < VIO_PRD) ? T.Jtinv[i * VIO_PRD + j] : 0.0;
            }
        }
        __syncthreads();
        {
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
    if (tid >= 192 && tid < 192 + n) T.dx[tid - 192] = sDx[tid - 192];
    if (tid >= 384 && tid < 384 + STATE_STRIDE) T.state[trial * STATE_STRIDE + (tid - 384)] = sState[tid - 384];
    CH_OUT(3);
}
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_c(DeviceTables T) { d_pose_solve_chain_body(T); }
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_cb(BatchArgs a) { const DeviceTables T = d_batch_tables(a); d_pose_solve_chain_body(T); }

#ifdef VIO_DEBUG_ENTRY_POINTS
// Diagnostic / test entry: solve one image (CH_PACKED doubles, lambda NOT yet on its diagonal) and return x by natural index;
// lds_dump (optional): the first CH_LDS_CORE doubles of LDS after the solve (L tiles, M, pivots, x) for tools/chain_solve_model.py
__global__ __launch_bounds__(PS_THREADS) void k_chain_solve_test(const double *img, double lambda, double *x_nat, double *lds_dump) {
    double *P = dyn_smem;
    const int tid = threadIdx.x;
    for (int i = tid; i < CH_PACKED; i += PS_THREADS) P[i] = img[i];
    for (int i = CH_PACKED + tid; i < CH_LDS_CORE; i += PS_THREADS) P[i] = 0.0;
    __syncthreads();
    if (tid < PS_N) {
        const int d = ch_dim(tid);
        if (d >= CH_YC) P[ch_cc_elem(d - CH_YC, d - CH_YC)] += lambda;
        else P[ch_sd(d >> 4) + (d & 15) * (CH_TS + 1)] += lambda;
    } else if (tid < PS_N + 8) {
        P[ch_cc_elem(72 + tid - PS_N, 72 + tid - PS_N)] = 1.0;
    } else if (tid >= 256 && tid < 256 + CH_S9SZ + PS_TS) {
        const int i = tid - 256;
        P[CH_OFF_I9 + i] = (i < CH_S9SZ) ? ((i / CH_TS == i % CH_TS) ? 1.0 : 0.0) : (((i - CH_S9SZ) / PS_TROW == (i - CH_S9SZ) % PS_TROW) ? 1.0 : 0.0);
    } else if (tid >= 640 && tid < 640 + CH_NDIM) {
        P[CH_OFF_D + tid - 640] = 1.0;                  // pivots of padding dimensions
    }
    for (int i = tid; i < CH_NS * CH_S9SZ; i += PS_THREADS) P[CH_OFF_SM + i] = 0.0;      // (the padding column of every M_e must read as zero)
    __syncthreads();
    ch_factor_solve(P, tid, [](int) {}, [](int, int) {});
    if (tid < PS_N) x_nat[tid] = P[CH_OFF_X + ch_dim(tid)];
    if (lds_dump) for (int i = tid; i < CH_LDS_CORE; i += PS_THREADS) lds_dump[i] = P[i];
}
#endif  // VIO_DEBUG_ENTRY_POINTS
#endif  // __HIPCC__
#endif
