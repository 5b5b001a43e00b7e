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
#define CH_OFF_SO (CH_OFF_SC + CH_NS * 5 * CH_SCSZ)        // 9680
#define CH_OFF_SD (CH_OFF_SO + 10 * CH_S9SZ)                // 10670
#define CH_OFF_CC (CH_OFF_SD + CH_NS * CH_S9SZ)             // 11759
#define CH_OFF_Y (CH_OFF_CC + 15 * PS_TS)                   // 15839
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
// RIDE: the third 16-lane row carries the rows of ANOTHER tile through the same column operations — `rtile` (row stride TS), a block
// of rows below the diagonal block in the same block column — and leaves l = a / d of every pivot in place of the entry: the tile
// comes out as L (what (A M) / d gives) without a product, a reciprocal or a second pass; `scratch` = 64 doubles nobody reads (the
// stores a lane has no use for go there: one instruction stream for all four rows of lanes).
template <int NP, int TS, int MS, bool RIDE = false>
__device__ __noinline__ void ch_factor(lds_double *tile, lds_double *sI, lds_double *M, int lane, lds_double *rtile = nullptr, lds_double *scratch = nullptr) {
    asm volatile("" : "+v"(tile));
    const bool ident = (lane >> 4) == 1;
    const bool ride = RIDE && (lane >> 4) == 2;
    const int row = min(lane & 15, NP - 1);           // lanes past the block repeat its last row (identical stores)
    lds_double *p0 = (ident ? sI : (ride ? rtile : tile)) + row * TS;
    double a0[NP], u[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) a0[j] = p0[j];
    lds_double *wp = ident ? M + row * MS : (ride ? scratch + lane : tile + row);
    const int ws = ident ? 1 : (ride ? 0 : TS);
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
        if (RIDE) a0[j] = l0;                   // (the entry is dead in every lane: published, or — the riding rows — replaced by its quotient)
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
    if (RIDE && ride) {
        // the riding rows leave as L, in place (kept in registers until here: a store per pivot cost 40 ticks a pivot, tools/microbench/factor_tile.hip)
        lds_double *lp = rtile + row * TS;
#pragma unroll
        for (int j = 0; j < NP; ++j) lp[j] = a0[j];
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
__device__ __forceinline__ bool d_fused_pair_ij(int b, int tid, int &I, int &J) {
    // block b = VIS_PAIR(P, Q), P <= Q; thread = entry (a, bq) of the 6 x 6 block.  A diagonal block holds both halves: the lower one is
    // what d_hs_entry reads
    int P = 0;
    while (VIS_PAIR(P + 1, P + 1) <= b) ++P;
    const int Q = P + (b - VIS_PAIR(P, P));
    const int a = tid / 6, bq = tid - 6 * a;
    if (P == Q && a < bq) return false;
    const int i = cam_to_full(6 * P + a), j = cam_to_full(6 * Q + bq);
    I = max(i, j); J = min(i, j);
    return true;
}
__device__ double d_fused_pair_pre(const DeviceTables &T, int b, int tid) {
    int I, J;
    return d_fused_pair_ij(b, tid, I, J) ? d_hs_rest(T, d_imu_mask(T), I, J) : 0.0;
}
__device__ void d_fused_pair(const DeviceTables &T, int b, int tid, double tot, double rest) {
    int I, J;
    if (d_fused_pair_ij(b, tid, I, J)) d_fused_store(T, I, J, tot + rest);
}
__device__ void d_fused_vec_pre(const DeviceTables &T, int P, int tid, double &extra, double &dgrest) {
    const int valid = d_imu_mask(T), i = cam_to_full(6 * P + tid);
    extra = d_rhs_rest(T, valid, i, d_cur(T));
    dgrest = d_hs_rest(T, valid, i, i);
}
__device__ void d_fused_vec(const DeviceTables &T, int P, int tid, double bd, double bc, double dg, double extra, double dgrest) {
    const int wset = d_set_w(T);
    const int i = cam_to_full(6 * P + tid);
    const double bred = bd - bc;
    T.bs[i] = bred + extra;
    T.bfull[wset * 176 + i] = bd + extra;
    T.diagfull[i] = dg + dgrest;
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
template <bool SPLIT>
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
    // SPLIT (the GN loop, gn_flags bit 4): the speed-bias chain of this system was eliminated under the linearisation (d_chain_pre_item, a
    // workgroup of k_linearize's grid: it depends on the IMU factors, the prior and lambda only); what it left — L_SC, L_SO, M_e, the pivots,
    // w_e and the tile mask, at their LDS offsets in T.cfi — comes in with the camera block and the camera right-hand side of the image
    int set = lm_loop ? (lm->sys ^ lm->pending) : 0;
    const double *cfi = SPLIT ? T.cfi : nullptr;
    double xs0 = 0.0, xs1 = 0.0, xd = 0.0, xy = 0.0, xe = 0.0;
#ifndef CH_NO_EARLY_START
    const bool early = !SPLIT && !lm_loop && !(T.gn_flags & 32);
#else
    const bool early = false;
#endif
    // (the pose part of b_ for the gain ratio's denominator, read in the kernel's last lines: requested here, a memory latency earlier)
    const double bf_early = (early && tid >= 192 && tid < 192 + n) ? T.bfull[tid - 192] : 0.0;
    if (early) {
        // The GN loop and the stepwise solves (lambda and the set are known at entry): NO BARRIER IN FRONT OF THE CHAIN'S FIRST LEVEL.  The
        // two chain waves fetch their own chain's tiles — SO and SD of blocks 0..5 resp. 6..10, 8 KB each — put lambda on them and start;
        // the other fourteen waves bring in the rest of the image (110 KB: one CU's load path needs 2 k ticks for it on top of the latency)
        // and meet them at barrier 0.  Level 0, 3 k ticks, used to begin when the whole copy was through (4.7 k).
        const double *img = T.Pg;
        if (uwave < 2) {
            const int so0 = CH_OFF_SO + (uwave ? 5 * CH_S9SZ : 0), sd0 = CH_OFF_SD + (uwave ? 6 * CH_S9SZ : 0);
            const int nsd = uwave ? 5 * CH_S9SZ : 6 * CH_S9SZ, e0 = uwave ? 6 : 0, ne = uwave ? 5 : 6;
            double vo[8], vd[10];
#pragma unroll
            for (int q = 0; q < 8; ++q) vo[q] = img[so0 + min(lane + 64 * q, 5 * CH_S9SZ - 1)];
#pragma unroll
            for (int q = 0; q < 10; ++q) vd[q] = img[sd0 + min(lane + 64 * q, nsd - 1)];
            // under the latency: the identity F starts its second row of lanes from, M_e = 0 (its padding column must read as zero), pivots 1
            for (int i = lane; i < CH_S9SZ; i += 64) P[CH_OFF_I9 + i] = (i / CH_TS == i % CH_TS) ? 1.0 : 0.0;
            for (int i = lane; i < ne * CH_S9SZ; i += 64) P[CH_OFF_SM + e0 * CH_S9SZ + i] = 0.0;
            for (int i = lane; i < ne * 16; i += 64) P[CH_OFF_D + e0 * 16 + i] = 1.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) P[so0 + min(lane + 64 * q, 5 * CH_S9SZ - 1)] = vo[q];
#pragma unroll
            for (int q = 0; q < 10; ++q) P[sd0 + min(lane + 64 * q, nsd - 1)] = vd[q];
            // lambda on the chain's pivots (problem.cc:434-436)
            if (lane < ne * 9) P[ch_sd(e0 + lane / 9) + (lane % 9) * (CH_TS + 1)] += lambda;
        } else {
            const int wt = tid - 128;                                   // 0..895
            // the speed-bias / camera coupling: a wave owns whole tiles (88 double2 each: lanes 0..63, then 0..23) and notes which of them hold
            // a non-zero (ch_scan_tiles' flags, without its pass over the image)
            const double2 *src2 = reinterpret_cast<const double2 *>(img);
            double2 *dst2 = reinterpret_cast<double2 *>(P);
            double2 va[4], vb[4], vc[3];
            static_assert(CH_OFF_CC % 2 == 1 && CH_PACKED % 2 == 0 && (CH_PACKED - CH_OFF_CC - 1) / 2 <= 3 * 896, "the copy of the camera block below");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = min(uwave - 2 + 14 * k, 54);
                va[k] = src2[t * (CH_SCSZ / 2) + lane];
                vb[k] = src2[t * (CH_SCSZ / 2) + 64 + min(lane, CH_SCSZ / 2 - 65)];
            }
            // the camera block and the right-hand side
#pragma unroll
            for (int q = 0; q < 3; ++q) vc[q] = src2[min((CH_OFF_CC + 1) / 2 + wt + 896 * q, CH_PACKED / 2 - 1)];
            const double vc0 = img[CH_OFF_CC];                          // (the block starts at an odd offset: its first element apart)
            double st[3] = {0.0, 0.0, 0.0}, er[3] = {0.0, 0.0, 0.0};
            const bool test_here = uwave == 2 && d_step_owed(T, 1);
            double tv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, ichi = 0.0;
            if (test_here) {
                // everything the test reads, requested with the image — the IMU edges' chi2 as ONE load (lane = edge): one at a time behind
                // its valid bit they are ten dependent round trips
                tv[0] = T.vis[VIS_CHI]; tv[1] = T.vis[VIS_STEP + 1]; tv[2] = lm->chi; tv[3] = T.sp_part[0]; tv[4] = T.sp_part[1]; tv[5] = T.sp_part[2];
                ichi = T.imu_out[min(lane, 9) * IMU_OUT + IMU_CHI];
            }
            if (uwave == 2) {
#pragma unroll
                for (int k = 0; k < 3; ++k) st[k] = T.state[cur * STATE_STRIDE + min(lane + 64 * k, STATE_STRIDE - 1)];
                if (test_here && T.has_prior) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) if (lane + 64 * k < VIO_PRD) er[k] = T.errprior[cur * 160 + lane + 64 * k];
                }
            }
            // under the latency: the 16 x 16 identity, the camera part's pivots = 1
            for (int i = wt; i < PS_TS; i += 896) P[CH_OFF_I16 + i] = (i / PS_TROW == i % PS_TROW) ? 1.0 : 0.0;
            if (wt < CH_NDIM - CH_YC) P[CH_OFF_D + CH_YC + wt] = 1.0;
            int *sNZ = (int *)(P + CH_OFF_NZ);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = uwave - 2 + 14 * k;
                if (t < 55) {
                    dst2[t * (CH_SCSZ / 2) + lane] = va[k];
                    if (lane < CH_SCSZ / 2 - 64) dst2[t * (CH_SCSZ / 2) + 64 + lane] = vb[k];
                    const bool nz = va[k].x != 0.0 || va[k].y != 0.0 || (lane < CH_SCSZ / 2 - 64 && (vb[k].x != 0.0 || vb[k].y != 0.0));
                    const unsigned long long any = __ballot(nz);
                    if (lane == 0) sNZ[t] = any != 0ull ? 1 : 0;
                }
            }
            // lambda on the camera block's pivots as they pass; the 8 padding variables of the last tile are identity rows
            auto cam_value = [&](int o, double v) {
                if (o < CH_OFF_Y) {
                    const int ti = (o - CH_OFF_CC) / PS_TS, w = (o - CH_OFF_CC) % PS_TS, r = w / PS_TROW, c = w % PS_TROW;
                    const bool dtile = ti == 0 || ti == 2 || ti == 5 || ti == 9 || ti == 14;
                    if (dtile && r == c && c < 16) v = (ti == 14 && r >= 8) ? 1.0 : v + lambda;
                }
                return v;
            };
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i2 = (CH_OFF_CC + 1) / 2 + wt + 896 * q;
                if (i2 < CH_PACKED / 2) { double2 v = vc[q]; v.x = cam_value(2 * i2, v.x); v.y = cam_value(2 * i2 + 1, v.y); dst2[i2] = v; }
            }
            if (wt == 0) P[CH_OFF_CC] = cam_value(CH_OFF_CC, vc0);
            if (uwave == 2) {
                // the state, and the test of the previous step (see the other prologue: the same sums — three wave partials added in wave
                // order — by one wave, nobody waits for anybody)
#pragma unroll
                for (int k = 0; k < 3; ++k) if (lane + 64 * k < STATE_STRIDE) sState[lane + 64 * k] = st[k];
                if (test_here) {
                    double w[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) w[k] = d_wave_sum_to_lane63(er[k] * er[k]);
                    const int valid = d_imu_mask(T);
                    double t_imu = 0.0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) { const double c = d_readlane(ichi, k); if ((valid >> k) & 1) t_imu += c; }
                    if (lane == 63) {
                        double total = tv[0] + t_imu;
                        if (T.has_prior) total += sqrt((w[0] + w[1]) + w[2]);          // err_prior_.norm(), not squared (problem.cc:554-556)
                        const double tempChi = 0.5 * total;
                        const double scale = 0.5 * (tv[1] + ((tv[3] + tv[4]) + tv[5])) + 1e-6;
                        lm->chi_try = tempChi;
                        lm->scale = scale;
                        lm->rho = (tv[2] - tempChi) / scale;
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
    } else {
    if (SPLIT) {
        xs0 = cfi[CH_OFF_SM + tid];
        if (tid < CH_NS * CH_S9SZ - PS_THREADS) xs1 = cfi[CH_OFF_SM + PS_THREADS + tid];
        if (tid < CH_YC) { xd = cfi[CH_OFF_D + tid]; xy = cfi[CH_OFF_Y + tid]; }
        if (tid < 2) xe = cfi[CH_OFF_NZ + tid];
    }
    for (int pass = 0; pass < 2; ++pass) {
        const double2 *src = reinterpret_cast<const double2 *>(T.Pg + set * CH_SET_STRIDE);
        const double2 *csrc = reinterpret_cast<const double2 *>(cfi);
        double2 *dst = reinterpret_cast<double2 *>(P);
        static_assert((CH_PACKED / 2 + PS_THREADS - 1) / PS_THREADS == 8, "copy below is written for 8 rounds");
        static_assert(CH_OFF_SD % 2 == 0, "the range that comes from cfi is whole double2s");
#define CH_FROM_CFI(i2) (SPLIT && (i2) < CH_OFF_SD / 2)
#define CH_LD(q) const double2 v##q = (CH_FROM_CFI(tid + q * PS_THREADS) ? csrc : src)[min(tid + q * PS_THREADS, CH_PACKED / 2 - 1)];
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
                if (uwave == 0) {
                    // (the IMU edges' chi2 as ONE load, lane = edge: one at a time behind its valid bit they are ten dependent round trips)
                    const int valid = d_imu_mask(T);
                    const double ichi = T.imu_out[min(lane, 9) * IMU_OUT + IMU_CHI];
                    t_chi = T.vis[VIS_CHI]; t_step = T.vis[VIS_STEP + 1]; t_lmchi = lm->chi;
                    t_sp = (T.sp_part[0] + T.sp_part[1]) + T.sp_part[2];
#pragma unroll
                    for (int k = 0; k < 10; ++k) { const double c = d_readlane(ichi, k); if ((valid >> k) & 1) t_imu += c; }
                }
            }
            // (LmState for the verdict below is requested here, with the image: behind the barrier it was one more round trip per slot)
            int pend0 = 0, sys0 = 0;
            if (lm_loop && tid == 0 && test_here) { d_lm_load(lm, sLm); pend0 = lm->pending; sys0 = lm->sys; }
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
                        int go = 1, rej = 0, sys = sys0;
                        if (pend0) {
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
#undef CH_FROM_CFI
    }
    // lambda on the 171 pivots (problem.cc:434-436); the 8 padding variables of the last camera tile are identity rows
    if (tid < n) {
        const int d = ch_dim(tid);
        if (d >= CH_YC) P[ch_cc_elem(d - CH_YC, d - CH_YC)] += lambda;
        else if (!SPLIT) P[ch_sd(d >> 4) + (d & 15) * (CH_TS + 1)] += lambda;       // (SPLIT: the chain's pivots carry it already)
    } else if (tid < n + 8) {
        P[ch_cc_elem(72 + tid - n, 72 + tid - n)] = 1.0;
    } else if (tid >= 256 && tid < 256 + CH_S9SZ + PS_TS) {
        const int i = tid - 256;
        P[CH_OFF_I9 + i] = (i < CH_S9SZ) ? ((i / CH_TS == i % CH_TS) ? 1.0 : 0.0) : (((i - CH_S9SZ) / PS_TROW == (i - CH_S9SZ) % PS_TROW) ? 1.0 : 0.0);
    } else if (tid >= 640 && tid < 640 + CH_NDIM) {
        if (!SPLIT || tid - 640 >= CH_YC) P[CH_OFF_D + tid - 640] = 1.0;           // pivots of padding dimensions
    }
    if (SPLIT) {
        P[CH_OFF_SM + tid] = xs0;
        if (tid < CH_NS * CH_S9SZ - PS_THREADS) P[CH_OFF_SM + PS_THREADS + tid] = xs1;
        if (tid < CH_YC) { P[CH_OFF_D + tid] = xd; P[CH_OFF_Y + tid] = xy; }      // (the image's own speed-bias rows came in with the copy: replaced by w_e)
        if (tid < 2) P[CH_OFF_NZ + tid] = xe;
    } else {
        for (int i = tid; i < CH_NS * CH_S9SZ; i += PS_THREADS) P[CH_OFF_SM + i] = 0.0;      // (the padding column of every M_e must read as zero)
    }
    __syncthreads();
    }
    CH_OUT(0);
    const int trial = cur ^ 1;
    // The camera part of the step (poses, extrinsic) is known two phases before the speed-bias part: wave 13 forms the trial poses
    // (UpdateStates, problem.cc:456-463; vertex_pose.cc:7-19) and their rotations while the speed-bias owners finish their sums, and the
    // fourteen waves that do not walk the chains form the pair table of the trial states meanwhile (in the camera tiles' space, free once
    // the camera part is solved).  The stepwise path (prior update here) keeps k_pose_solve's serial tail.
    const bool prior_here = T.has_prior && !(T.gn_flags & 4);
    const int lm_dim = T.lm_dim;
    double *pairtab_trial = T.pairtab + trial * PAIRTAB_STRIDE;
    double *pw = P + CH_OFF_CC;                                 // trial ext + poses at [0, 84), rotations at [96, 204)
    auto mid1 = [&](int ln) {
        if (prior_here || CH_DIAG_SKIP == 2) return;
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
        if (prior_here || lm_dim == 3 || CH_DIAG_SKIP == 1 || CH_DIAG_SKIP == 2) return;
        d_pair_rows(pw, pairtab_trial, pw + 96, ln + 64 * idx, 14 * 64);
    };
    const bool ext_trivial = T.ext_fixed && !T.marg_mode;       // (the extrinsic's rows of this system: lambda on the diagonal, nothing beside it)
    if (SPLIT) {
        unsigned long long t0__ = 0ull;
#ifdef VIO_STAMPS
        t0__ = __builtin_amdgcn_s_memtime();
        if (tid == 0) { g_ch_dbg = T.dbg; g_ch_t0 = t0__; }
#endif
        const unsigned lo = (unsigned)P[CH_OFF_NZ], hi = (unsigned)P[CH_OFF_NZ + 1];
        const unsigned long long eff = ((unsigned long long)__builtin_amdgcn_readfirstlane(hi) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(lo);
        ch_camera_solve<true>(P, tid, ch_lane(lane), eff, mid1, mid2, T.dbg, t0__, ext_trivial);
    } else {
        ch_factor_solve(P, tid, mid1, mid2, T.dbg, ext_trivial, early);
    }
    CH_OUT(2);
    for (int i = tid; i < n; i += PS_THREADS) sDx[i] = sX[ch_dim(i)];
    if (tid >= 192 && tid < 384 && T.sp_part) {
        // sum_i dx_i (lambda dx_i + b_i), b_ of the system just solved: three wave partials, as k_assemble's step test sums them
        const int i = tid - 192;
        const double dxi = (i < n) ? sX[ch_dim(min(i, n - 1))] : 0.0;
        const double bi = early ? bf_early : ((i < n) ? T.bfull[set * 176 + i] : 0.0);
        const double w = d_wave_sum_to_lane63((i < n) ? dxi * (lambda * dxi + bi) : 0.0);
        if (lane == 63) T.sp_part[i >> 6] = w;
    }
    if (!prior_here) {
        // the trial states: poses from wave 13's copy, speed-bias = current + dx
        if (tid >= 128 && tid < 128 + 84) sState[tid - 128] = P[CH_OFF_CC + tid - 128];
        else if (tid >= 256 && tid < 256 + 99) { const int e = tid - 256; sState[STATE_SB + e] += sX[(e / 9) * 16 + (e % 9 < 3 ? 6 + e % 9 : (e % 9 < 6 ? e % 9 : e % 9 - 6))]; }
        d_lds_barrier();
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
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_c(DeviceTables T) { d_pose_solve_chain_body<false>(T); }
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_cb(BatchArgs a) { const DeviceTables T = d_batch_tables(a); d_pose_solve_chain_body<false>(T); }
// the GN loop's form: the camera block and the back-substitution behind a chain that d_chain_pre_item eliminated under the linearisation
__global__ __launch_bounds__(PS_THREADS) void k_pose_solve_cs(DeviceTables T) { d_pose_solve_chain_body<true>(T); }

// ---------------------------------------------------------------------------------------------------------
// The pre-elimination of the speed-bias chain (the GN loop, fixed lambda).  The chain's blocks — SD, SO, SC and the speed-bias rows of
// the right-hand side — hold IMU and prior terms only (the landmark Schur complement reaches the camera block alone, problem.cc:427-429),
// so with lambda known their elimination does not wait for the linearisation of the window's 80 000 observations: it runs beside it, on a
// CU of its own, and k_pose_solve_cs starts at the camera block.  ch_chain_pre_store: what the elimination leaves, at its LDS offsets.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ch_chain_pre_store(const double *P, int tid, unsigned long long eff, double *cfi) {
    {   // L_SC, L_SO: 5335 double2, every LDS read in flight before the first store
        const double2 *s2 = reinterpret_cast<const double2 *>(P);
        double2 *d2 = reinterpret_cast<double2 *>(cfi);
        constexpr int N2 = CH_OFF_SD / 2;
        static_assert(5 * PS_THREADS < N2 && N2 <= 6 * PS_THREADS, "six rounds");
        const double2 v0 = s2[tid], v1 = s2[tid + PS_THREADS], v2 = s2[tid + 2 * PS_THREADS], v3 = s2[tid + 3 * PS_THREADS], v4 = s2[tid + 4 * PS_THREADS];
        const int i5 = tid + 5 * PS_THREADS;
        double2 v5 = make_double2(0.0, 0.0);
        if (i5 < N2) v5 = s2[i5];
        d2[tid] = v0; d2[tid + PS_THREADS] = v1; d2[tid + 2 * PS_THREADS] = v2; d2[tid + 3 * PS_THREADS] = v3; d2[tid + 4 * PS_THREADS] = v4;
        if (i5 < N2) d2[i5] = v5;
    }
    for (int i = tid; i < CH_NS * CH_S9SZ; i += PS_THREADS) cfi[CH_OFF_SM + i] = P[CH_OFF_SM + i];               // M_e
    if (tid < CH_YC) { cfi[CH_OFF_D + tid] = P[CH_OFF_D + tid]; cfi[CH_OFF_Y + tid] = P[CH_OFF_Y + tid]; }      // pivots, w_e
    if (tid == 0) { cfi[CH_OFF_NZ] = (double)(unsigned)(eff & 0xffffffffull); cfi[CH_OFF_NZ + 1] = (double)(unsigned)(eff >> 32); }
}
// the chain image's constants around the S part (identity for F's second half-row, pivots of the padding dimensions, zero M tiles)
__device__ __forceinline__ void ch_chain_pre_init(double *P, int tid) {
    if (tid >= 256 && tid < 256 + CH_S9SZ) { const int i = tid - 256; P[CH_OFF_I9 + i] = (i / CH_TS == i % CH_TS) ? 1.0 : 0.0; }
    else if (tid >= 640 && tid < 640 + CH_YC) P[CH_OFF_D + tid - 640] = 1.0;
    for (int i = tid; i < CH_NS * CH_S9SZ; i += PS_THREADS) P[CH_OFF_SM + i] = 0.0;
}
// ---- the chain workgroup of k_linearize's grid (GN loop, gn_flags bit 4; blockIdx 0) ------------------------------------------------
// It forms the ten IMU items itself (nobody to wait for: the IMU workgroups that follow the items in the other paths are not launched),
// writes them to imu_out for k_reduce_c, assembles the chain's blocks from them and the prior, eliminates the chain, stores the factors.
// Same arithmetic as d_imu_item + d_hs_rest / d_rhs_rest + the chain phase of k_pose_solve_c, operation for operation:
//   * Jacobian blocks / residual: the same device functions, one WAVE per block kind (a uniform branch; d_imu_item runs the sixteen
//     branches one after the other in one wave), lane = IMU edge; what all blocks share (d_imu_common, R_i^T) is formed once, by wave 15
//     under the staging;
//   * J^T Info J: the dense sums of d_imu_item are chains of FMAs over the 15 residual rows in ascending order; J is 18 non-zero 3 x 3 blocks
//     of 50, and a term with an exactly zero factor leaves the sum as it is, so the chains run over the non-zero row blocks only (3 x 3
//     register tiles, a wave per column group so that the row blocks are the same for all its lanes; only the vertex blocks on and above
//     the diagonal, the ones Problem::MakeHessian computes, problem.cc:337-355);
//   * an entry of the image is 0 + T_(f-1) + T_f + prior (d_hs_rest's order; the two IMU terms commute): the items of even edges are
//     stored, those of odd edges added, then the prior's image (k_prior_simg: the masked H_prior entries at their places) is added.
#define CPI_PRE 0                               // staged pre-integrations [10][PRE_STRIDE]   (the image's place, free until the scatter)
#define CPI_J (CPI_PRE + 10 * PRE_STRIDE)       // J [10][15 x 30]
#define CPI_ST (CPI_J + 10 * 450)               // states (184)
#define CPI_CM (CPI_ST + 184)                   // per edge 40: ImuCommon (27) | R_i^T (9, at 28)
#define CPI_JTI(k) ((k) < 9 ? CH_OFF_CC + 450 * (k) : CH_LDS_CORE)       // J^T Info [30 x 15] of edge k (the camera tiles' place; edge 9 behind the core)
#define CPI_VEC (CH_LDS_CORE + 450)             // per edge 64: r (15) | Info r (15, at 16) | g (30, at 32)
#define CPI_BP (CPI_VEC + 640)                  // b_prior of the linearisation state (176)
#define CPI_PN (CPI_BP + 176)                   // the prior's list sizes (2 ints)
#define CPI_END (CPI_PN + 2)
#define CPI_NTILES (55 + 10 + 11)               // SC, SO, SD tiles of the chain image
#define CPI_FLAGS (CPI_NTILES + 99)             // prior_flags: tiles, then speed-bias rows
static_assert(CPI_CM + 400 <= CH_OFF_CC && CH_OFF_CC + 9 * 450 <= CH_OFF_Y, "staging fits the image's place");
static_assert(sizeof(ImuCommon) == 27 * 8, "ImuCommon is staged as 27 doubles");
__host__ __device__ inline int cpi_vb(int cg) { return cg < 2 ? 0 : (cg < 5 ? 1 : (cg < 7 ? 2 : 3)); }           // vertex of a column group of 3
__host__ __device__ inline void cpi_tile(int tau, int &ca, int &cb) {      // the 63 tiles (ca, cb) with vertex(ca) <= vertex(cb), row-major
    if (tau < 20) { ca = tau / 10; cb = tau % 10; }
    else if (tau < 44) { const int q = tau - 20; ca = 2 + q / 8; cb = 2 + q % 8; }
    else if (tau < 54) { const int q = tau - 44; ca = 5 + q / 5; cb = 5 + q % 5; }
    else { const int q = tau - 54; ca = 7 + q / 3; cb = 7 + q % 3; }
}
__device__ __forceinline__ int cpi_tau(int ca, int cb) { return ca < 2 ? ca * 10 + cb : (ca < 5 ? 20 + (ca - 2) * 8 + (cb - 2) : (ca < 7 ? 44 + (ca - 5) * 5 + (cb - 5) : 54 + (ca - 7) * 3 + (cb - 7))); }
// which of the five residual row blocks (P, R, V, BA, BG) of J are non-zero in column group cg (edge_imu.cc:74-153)
// (bit rb; column groups: pose_i P, R | speed-bias_i v, ba, bg | pose_j P, R | speed-bias_j v, ba, bg)
__device__ __forceinline__ unsigned cpi_rows(int cg) {
    return cg == 0 ? 0x01u : cg == 1 ? 0x07u : cg == 2 ? 0x05u : cg == 3 ? 0x0du : cg == 4 ? 0x17u : cg == 5 ? 0x01u : cg == 6 ? 0x02u : cg == 7 ? 0x04u : cg == 8 ? 0x08u : 0x10u;
}
// tile index of an image position p < CH_OFF_CC: SC tiles 0..54, SO 55..64, SD 65..75
__host__ __device__ inline int cpi_tile_of(int p) { return p < CH_OFF_SO ? p / CH_SCSZ : (p < CH_OFF_SD ? 55 + (p - CH_OFF_SO) / CH_S9SZ : 65 + (p - CH_OFF_SD) / CH_S9SZ); }
__host__ __device__ inline int cpi_tile_base(int t) { return t < 55 ? t * CH_SCSZ : (t < 65 ? CH_OFF_SO + (t - 55) * CH_S9SZ : CH_OFF_SD + (t - 65) * CH_S9SZ); }
__host__ __device__ inline int cpi_tile_size(int t) { return t < 55 ? CH_SCSZ : CH_S9SZ; }

__device__ __forceinline__ void d_chain_pre_item(const DeviceTables &T) {
    double *P = dyn_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cur = d_cur(T);
    const int valid = d_imu_mask(T);
    const bool owe_prior = d_step_owed(T, 2) && T.has_prior;
    const double lambda = T.lm->lambda;
#ifdef VIO_STAMPS
    // diagnostic build: phase stamps of this workgroup behind the items' and IMU workgroups' slots (tools/diag_chain_pre_stamps.py)
    unsigned long long *cdbg = T.dbg ? T.dbg + 16 * (size_t)(T.n_items + 11) : nullptr;
    const unsigned long long cp_t0 = __builtin_amdgcn_s_memtime();
#define CP_STAMP(i) do { if (tid == 0 && cdbg) cdbg[i] = __builtin_amdgcn_s_memtime() - cp_t0; } while (0)
#else
    unsigned long long *cdbg = nullptr;
    const unsigned long long cp_t0 = 0ull;
#define CP_STAMP(i) do { } while (0)
#endif
    // ---- stage: pre-integrations, states; constants of the image; what the Jacobian blocks share; b_prior of this state ----
    {
        double pv[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int i = tid + q * PS_THREADS; pv[q] = i < 10 * PRE_STRIDE ? T.pre[i] : 0.0; }
        const double sv = tid < STATE_STRIDE ? T.state[cur * STATE_STRIDE + tid] : 0.0;
        // the prior's lists (k_prior_compact): n entries of the chain's blocks, m speed-bias rows of H_prior that hold anything
        int pn = 0, pm = 0, prow = -1;
        double bpv = 0.0;
        bool bp_skip = false;
        if (T.has_prior) {
            pn = T.prior_list[0]; pm = T.prior_list[1];
            // thread 128 + r: b_prior of speed-bias row r as it stands (replaced below by b_prior - H_prior dx where that is owed and the row
            // of H_prior holds anything)
            if (tid >= 128 && tid < 128 + 99) {
                const int r = tid - 128;
                bpv = T.bprior[(owe_prior ? cur ^ 1 : cur) * 176 + 12 + 15 * (r / 9) + r % 9];
                bp_skip = owe_prior && T.prior_flags[CPI_NTILES + r] != 0;        // (a listed row: its dot product is the value)
            }
            if (owe_prior && wave < 15) prow = T.prior_list[2 + min(wave, 98)];         // (wave w: listed rows w, w + 15, ...: the first one now)
        }
        if (wave == 15 && lane < 10 && ((valid >> lane) & 1)) {
            // (from global memory: the staged copies are not there yet; the same values)
            const int k = lane;
            const double *pre = T.pre + k * PRE_STRIDE, *st = T.state + cur * STATE_STRIDE;
            ImuCommon c;
            double RiT[9];
            d_imu_common(pre, st + STATE_POSE + 7 * k, st + STATE_SB + 9 * k, st + STATE_POSE + 7 * k + 7, c);
            d_imu_rit(c, RiT);
            double *cm = P + CPI_CM + 40 * k;
            const double *cc = reinterpret_cast<const double *>(&c);
#pragma unroll
            for (int q = 0; q < 27; ++q) cm[q] = cc[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) cm[28 + q] = RiT[q];
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int i = tid + q * PS_THREADS; if (i < 10 * PRE_STRIDE) P[CPI_PRE + i] = pv[q]; }
        if (tid < STATE_STRIDE) P[CPI_ST + tid] = sv;
        double2 *zj = reinterpret_cast<double2 *>(P + CPI_J);
        static_assert(CPI_J % 2 == 0, "J is zeroed as double2");
        for (int i = tid; i < 10 * 450 / 2; i += PS_THREADS) zj[i] = make_double2(0.0, 0.0);
        if (tid < CH_YC) P[CH_OFF_Y + tid] = 0.0;
        ch_chain_pre_init(P, tid);
        if (T.has_prior && tid >= 128 && tid < 128 + 99 && !bp_skip) { const int r = tid - 128; P[CPI_BP + 12 + 15 * (r / 9) + r % 9] = bpv; }
        // b_prior - H_prior dx of the listed rows (d_bprior_dot: the sums the item workgroups that own those rows form in this launch)
        if (owe_prior && wave < 15 && wave < pm) {
            const bool in2 = lane + 128 < VIO_PD;
            const double x0 = T.dx[lane], x1 = T.dx[lane + 64], x2 = in2 ? T.dx[lane + 128] : 0.0;
            for (int q = wave; q < pm; q += 15) {
                const int r = q == wave ? prow : T.prior_list[2 + q];
                const int i = 12 + 15 * (r / 9) + r % 9;
                const double *hr = T.Hprior + (size_t)i * VIO_PD;
                const double v = d_bprior_dot(hr[lane], hr[lane + 64], in2 ? hr[lane + 128] : 0.0, x0, x1, x2, T.bprior[(cur ^ 1) * 176 + i]);
                if (lane == 63) P[CPI_BP + i] = v;
            }
        }
        if (tid == 0) { reinterpret_cast<int *>(P + CPI_PN)[0] = pn; reinterpret_cast<int *>(P + CPI_PN)[1] = pm; }
    }
    CP_STAMP(0);
    __syncthreads();
    CP_STAMP(1);
    // ---- I1: the Jacobian blocks and the residual; wave = block kind, lane = edge ----
    // (waves w, w + 4, w + 8, w + 12 share a SIMD: the residual (14) and the three blocks with quaternion products (2, 7, 12) go to four different ones)
    const int task = __builtin_amdgcn_readfirstlane((int)((0xfa85d964b310c72eull >> (4 * wave)) & 15ull));
    if (lane < 10 && ((valid >> lane) & 1)) {
        const int k = lane;
        const double *pre = P + CPI_PRE + k * PRE_STRIDE, *st = P + CPI_ST;
        const double *pi = st + STATE_POSE + 7 * k, *pj = pi + 7, *si = st + STATE_SB + 9 * k, *sj = si + 9;
        double *sJ = P + CPI_J + 450 * k, *sr = P + CPI_VEC + 64 * k;
        const double *cm = P + CPI_CM + 40 * k;
        if (task < 15) {
            // (what the blocks share stays in LDS and is read where it is used: a copy in registers — 27 + 9 doubles — beside a block's own
            // temporaries is what spilled in d_imu_item's sixteen-branch wave)
            const ImuCommon &c = *reinterpret_cast<const ImuCommon *>(cm);
            if (task < 14) {
                d_imu_jac_block(task, pre, T.gravity, pi, si, pj, sj, c, cm + 28, sJ);
            } else {
                double r[15];
                d_imu_residual(pre, T.gravity, pi, si, pj, sj, c, r);
                for (int i = 0; i < 15; ++i) sr[i] = r[i];
            }
        } else {
            for (int i = 0; i < 3; ++i) {       // identity blocks (edge_imu.cc:116-118,146-148)
                sJ[30 * (O_BA + i) + 6 + 3 + i] = -1.0; sJ[30 * (O_BG + i) + 6 + 6 + i] = -1.0;
                sJ[30 * (O_BA + i) + 21 + 3 + i] = 1.0; sJ[30 * (O_BG + i) + 21 + 6 + i] = 1.0;
            }
        }
    }
    CP_STAMP(2);
#ifdef VIO_STAMPS
    if (lane == 0 && cdbg) cdbg[32 + wave] = __builtin_amdgcn_s_memtime() - cp_t0;
#endif
    __syncthreads();
    CP_STAMP(3);
    // ---- I2: J^T Info (3 x 3 tiles; wave = column group of J, lanes = edge x 5 column groups of Info) and Info r ----
    if (wave < 10) {
        const int ca = wave, k = lane / 5, j0 = 3 * (lane % 5);
        if (lane < 50 && ((valid >> k) & 1)) {
            const double *sJ = P + CPI_J + 450 * k + 3 * ca, *sI = P + CPI_PRE + k * PRE_STRIDE + PRE_INFO + j0;
            double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            const unsigned rows = cpi_rows(ca);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                if (!((rows >> rb) & 1)) continue;          // (wave-uniform)
                double ja[3][3], ib[3][3];
#pragma unroll
                for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                    for (int u = 0; u < 3; ++u) { ja[ii][u] = sJ[30 * (3 * rb + ii) + u]; ib[ii][u] = sI[15 * (3 * rb + ii) + u]; }
#pragma unroll
                for (int ii = 0; ii < 3; ++ii)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int v = 0; v < 3; ++v) acc[u][v] = fma(ja[ii][u], ib[ii][v], acc[u][v]);
            }
            double *o = P + CPI_JTI(k) + 15 * 3 * ca + j0;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) o[15 * u + v] = acc[u][v];
        }
    } else if (wave < 13) {
        const int t = (wave - 10) * 64 + lane;
        const int k = t / 15, i = t % 15;
        if (t < 150 && ((valid >> k) & 1)) {
            const double *sI = P + CPI_PRE + k * PRE_STRIDE + PRE_INFO, *sr = P + CPI_VEC + 64 * k;
            double sum = 0;
            for (int j = 0; j < 15; ++j) sum = fma(sI[15 * i + j], sr[j], sum);
            P[CPI_VEC + 64 * k + 16 + i] = sum;
        }
    }
    CP_STAMP(4);
    __syncthreads();
    CP_STAMP(5);
    // ---- I3: T = (J^T Info) J in 3 x 3 register tiles (kept for the scatter, written to imu_out now); g = J^T Info r; chi = r^T Info r.
    //      Waves 0..14: one column group cb of J each (two waves for the groups with more than 64 tiles over the ten edges) ----
    double tv[9];
    unsigned tmap[9];
    int t_edge = -1;
#pragma unroll
    for (int q = 0; q < 9; ++q) { tv[q] = 0.0; tmap[q] = 0xffff0000u | (unsigned)CH_OFF_X; }
    if (wave < 15) {
        // wave -> (cb, first task, tasks): cb 0..4 one wave each, cb 5..9 two
        const int cb = wave < 5 ? wave : 5 + (wave - 5) / 2;
        const int nca = cb < 2 ? 2 : (cb < 5 ? 5 : (cb < 7 ? 7 : 10));          // tiles (ca, cb) with vertex(ca) <= vertex(cb)
        const int ntask = 10 * nca, half = (ntask + 1) / 2;
        const int q0 = wave < 5 ? 0 : ((wave - 5) & 1) * half, q1 = wave < 5 ? ntask : (((wave - 5) & 1) ? ntask : half);
        const int q = q0 + lane;
        const int k = q / nca, ca = q % nca;
        if (q < q1 && ((valid >> k) & 1)) {
            t_edge = k;
            const int tau = cpi_tau(ca, cb);
#pragma unroll
            for (int e = 0; e < 9; ++e) tmap[e] = T.imu_map[(k * 63 + tau) * 9 + e];
            const double *sA = P + CPI_JTI(k) + 15 * 3 * ca, *sJ = P + CPI_J + 450 * k + 3 * cb;
            double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
            const unsigned rows = cpi_rows(cb);
#pragma unroll
            for (int rb = 0; rb < 5; ++rb) {
                if (!((rows >> rb) & 1)) continue;          // (wave-uniform)
                double aa[3][3], jb[3][3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                    for (int u = 0; u < 3; ++u) { aa[jj][u] = sA[15 * u + 3 * rb + jj]; jb[jj][u] = sJ[30 * (3 * rb + jj) + u]; }
#pragma unroll
                for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int v = 0; v < 3; ++v) acc[u][v] = fma(aa[jj][u], jb[jj][v], acc[u][v]);
            }
            double *out = T.imu_out + k * IMU_OUT + IMU_T + 30 * 3 * ca + 3 * cb;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) { tv[3 * u + v] = acc[u][v]; out[30 * u + v] = acc[u][v]; }
        }
    }
    // g = J^T (Info r): 300 entries, twenty per wave of the fifteen above; chi = r^T Info r on wave 15
    if (wave < 15 && lane < 20) {
        const int t = wave * 20 + lane;
        const int k = t / 30, a = t % 30;
        if ((valid >> k) & 1) {
            const double *sJ = P + CPI_J + 450 * k + a, *sIr = P + CPI_VEC + 64 * k + 16;
            double ja[15], ir[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) { ja[i] = sJ[30 * i]; ir[i] = sIr[i]; }
            double sum = 0;
#pragma unroll
            for (int i = 0; i < 15; ++i) sum = fma(ja[i], ir[i], sum);
            P[CPI_VEC + 64 * k + 32 + a] = sum;
            T.imu_out[k * IMU_OUT + IMU_G + a] = sum;
        }
    } else if (wave == 15 && lane < 10) {
        const int k = lane;
        if ((valid >> k) & 1) {
            const double *sr = P + CPI_VEC + 64 * k, *sIr = sr + 16;
            double sum = 0;
            for (int i = 0; i < 15; ++i) sum = fma(sr[i], sIr[i], sum);
            T.imu_out[k * IMU_OUT + IMU_CHI] = sum;
        }
    }
    // the prior's entries (the list of k_prior_compact): requested now, added behind the items
    double pr[2] = {0.0, 0.0};
    int pr_pos[2] = {-1, -1};
    const int pn = T.has_prior ? reinterpret_cast<const int *>(P + CPI_PN)[0] : 0;
    if (pn <= 2 * PS_THREADS) {
#pragma unroll
        for (int h = 0; h < 2; ++h) if (tid + h * PS_THREADS < pn) { pr_pos[h] = T.prior_list[128 + tid + h * PS_THREADS]; pr[h] = T.prior_cval[tid + h * PS_THREADS]; }
    }
    CP_STAMP(6);
#ifdef VIO_STAMPS
    if (lane == 0 && cdbg) cdbg[48 + wave] = __builtin_amdgcn_s_memtime() - cp_t0;
#endif
    __syncthreads();
    CP_STAMP(7);
    // ---- the image: zero, items of the even edges, items of the odd edges, prior, right-hand side, lambda ----
    {
        double2 *z = reinterpret_cast<double2 *>(P);
        for (int i = tid; i < (CH_OFF_CC + 1) / 2; i += PS_THREADS) z[i] = make_double2(0.0, 0.0);      // (one double into the camera tiles' place: dead)
    }
    __syncthreads();
    CP_STAMP(8);
    // (an element without a place in the chain's blocks carries the dummy position CH_OFF_X in its low half; mirrors — the entries of a
    // diagonal block of the chain — are rare: a wave without any skips them as a whole)
    bool has_mirror = false;
#pragma unroll
    for (int q = 0; q < 9; ++q) has_mirror = has_mirror || (tmap[q] >> 16) != 0xffffu;
    has_mirror = has_mirror && t_edge >= 0;
    const bool wave_mirror = __ballot(has_mirror) != 0ull;
    if (t_edge >= 0 && (t_edge & 1) == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) P[tmap[q] & 0xffffu] = tv[q];            // (0 + v: the image is zero)
        if (wave_mirror) {
#pragma unroll
            for (int q = 0; q < 9; ++q) if ((tmap[q] >> 16) != 0xffffu) P[tmap[q] >> 16] = tv[q];
        }
    }
    CP_STAMP(14);
    __syncthreads();
    CP_STAMP(15);
    if (t_edge >= 0 && (t_edge & 1) == 1) {
        double old[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) old[q] = P[tmap[q] & 0xffffu];
#pragma unroll
        for (int q = 0; q < 9; ++q) P[tmap[q] & 0xffffu] = old[q] + tv[q];
        if (wave_mirror) {
#pragma unroll
            for (int q = 0; q < 9; ++q) if ((tmap[q] >> 16) != 0xffffu) P[tmap[q] >> 16] = old[q] + tv[q];
        }
    }
    __syncthreads();
    CP_STAMP(9);
    if (T.has_prior) {
        if (pn > 2 * PS_THREADS) { for (int i = tid; i < CH_OFF_CC; i += PS_THREADS) P[i] += T.prior_simg[i]; }     // (a prior that fills the chain's blocks)
        else {
            if (pr_pos[0] >= 0) P[pr_pos[0]] += pr[0];
            if (pr_pos[1] >= 0) P[pr_pos[1]] += pr[1];
        }
    }
    if (tid < 99) {
        // row i of b for a speed-bias variable: d_rhs_rest's sum
        const int f = tid / 9, i = 12 + 15 * f + tid % 9;
        double extra = 0.0;
        for (int k = f - 1; k <= f; ++k) {
            if (k < 0 || k >= 10 || !((valid >> k) & 1)) continue;
            const int a = i - (6 + 15 * k);
            extra -= P[CPI_VEC + 64 * k + 32 + a];
        }
        if (T.has_prior) extra += P[CPI_BP + i];
        P[CH_OFF_Y + ch_dim(i)] = extra;
    }
    __syncthreads();
    CP_STAMP(10);
    if (tid < CH_NS * 9) { const int e = tid / 9, k = tid % 9; P[ch_sd(e) + k * (CH_TS + 1)] += lambda; }
    __syncthreads();
    CP_STAMP(11);
    const unsigned long long eff = ch_chain_elimination<false>(P, tid, ch_lane(lane), cdbg, cp_t0);
    CP_STAMP(12);
    ch_chain_pre_store(P, tid, eff, T.cfi);
    CP_STAMP(13);
#undef CP_STAMP
}

// H_prior's contribution to the chain's blocks at their places in the image (the entries of the 99 speed-bias rows, masked as d_hs_rest
// masks them: problem.cc:372-381), and which tiles / rows hold a non-zero: built once per prior, used by d_chain_pre_item every iteration
__global__ __launch_bounds__(192) void k_prior_simg(DeviceTables T) {
    const int r = blockIdx.x, t = threadIdx.x;
    if (r >= 99 || t >= VIO_PD) return;
    const int i = 12 + 15 * (r / 9) + r % 9;
    // (the whole row counts for b_prior' = b_prior - H_prior dx, which is not masked: problem.cc:473)
    if (T.Hprior[i * VIO_PD + t] != 0.0) T.prior_flags[CPI_NTILES + r] = 1;
    if (!(full_to_cam(t) >= 0 || t <= i)) return;
    const int I = max(i, t), J = min(i, t);
    const bool mask = T.ext_fixed && !T.marg_mode && (I < 6 || J < 6);
    const double v = mask ? 0.0 : T.Hprior[I * VIO_PD + J];
    int p1, p2;
    ch_entry_pos(I, J, p1, p2);
    if (p1 >= 0) { T.prior_simg[p1] = v; if (v != 0.0) T.prior_flags[cpi_tile_of(p1)] = 1; }
    if (p2 >= 0) T.prior_simg[p2] = v;
}

// ... and the non-zero entries of that image as a list (position ascending), the speed-bias rows of H_prior that hold anything as another:
// what d_chain_pre_item reads every iteration is a few hundred entries (Problem::Marginalize fills the rows of frame 0's speed-bias alone)
__global__ __launch_bounds__(PS_THREADS) void k_prior_compact(DeviceTables T) {
    __shared__ int sCnt[PS_THREADS + 1];
    const int tid = threadIdx.x;
    constexpr int PER = (CH_OFF_CC + PS_THREADS - 1) / PS_THREADS;
    int c = 0;
    for (int q = 0; q < PER; ++q) { const int p = tid * PER + q; if (p < CH_OFF_CC && T.prior_simg[p] != 0.0) ++c; }
    sCnt[tid + 1] = c;
    if (tid == 0) sCnt[0] = 0;
    __syncthreads();
    if (tid == 0) for (int i = 1; i <= PS_THREADS; ++i) sCnt[i] += sCnt[i - 1];
    __syncthreads();
    int o = sCnt[tid];
    for (int q = 0; q < PER; ++q) {
        const int p = tid * PER + q;
        if (p < CH_OFF_CC) { const double v = T.prior_simg[p]; if (v != 0.0) { T.prior_list[128 + o] = p; T.prior_cval[o] = v; ++o; } }
    }
    if (tid == 0) {
        T.prior_list[0] = sCnt[PS_THREADS];
        int m = 0;
        for (int r = 0; r < 99; ++r) if (T.prior_flags[CPI_NTILES + r]) T.prior_list[2 + m++] = r;
        T.prior_list[1] = m;
    }
}

// Stage-one form (diagnostic, VIO_GN_SPLIT=2): the chain of the image k_reduce_c wrote, in a launch of its own between k_reduce_c and
// k_pose_solve_cs — the same arithmetic as the workgroup inside k_linearize's grid, fed from the assembled image instead of from the IMU items
__global__ __launch_bounds__(PS_THREADS) void k_chain_pre(DeviceTables T) {
    double *P = dyn_smem;
    const int tid = threadIdx.x;
    const double lambda = T.lm->lambda;
    const double *img = T.Pg + d_set_w(T) * CH_SET_STRIDE;
    for (int i = tid; i < CH_OFF_CC; i += PS_THREADS) P[i] = img[i];
    if (tid < CH_YC) P[CH_OFF_Y + tid] = img[CH_OFF_Y + tid];
    ch_chain_pre_init(P, tid);
    __syncthreads();
    if (tid < CH_NS * 9) { const int e = tid / 9, k = tid % 9; P[ch_sd(e) + k * (CH_TS + 1)] += lambda; }
    __syncthreads();
    const unsigned long long eff = ch_chain_elimination<false>(P, tid, ch_lane(tid & 63), T.dbg, 0ull);
    ch_chain_pre_store(P, tid, eff, T.cfi);
}

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
