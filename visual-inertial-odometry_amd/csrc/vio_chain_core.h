// vio_chain_core.h — ch_factor_solve: the factorisation and solve of the chain-order pose system on its LDS image
// (included by vio_pose_solve_chain.h, which documents the layout).
//
// Schedule (16 waves; "level" l = 0..5 of the speed-bias chain, blocks eA = l on wave 0 and eB = 10 - l on wave 1):
//   waves 0, 1   F(e) with the rows of SO[e] riding (they leave as L_SO[e]) -> pivots -> SD[succ] -= (L D) L^T on the matrix core -> barrier l.
//                They never wait for anybody: what they read (SD, SO) nobody else writes.  They run at instruction priority 3.
//   waves 2..15  after barrier l ("phase l"): one fused task per (chain, camera tile t): L_SC[e][t] = (SC[e][t] M_e) / d, formed
//                transposed, and the fill SC[succ][t] -= (L D) L_SO[e]^T from registers — no cross-wave dependency inside a phase;
//                the camera-block updates CC(I,J) -= (L_SC[e][I] D) L_SC[e][J]^T of the PREVIOUS level (they need two waves' tiles:
//                one barrier later), all fifteen tiles, fill the rest of the phase; wave 15 carries the right-hand side (priority 2).
//   behind barrier 5 a short phase 5a (the five L_SC[5][t]: wave 0 its own tile's), one barrier, then F(0) on wave 0 while the other waves
//                — not 4, 8, 12: they share wave 0's SIMD — give the camera tiles the terms of blocks 4, 6 and 5;
//   then the camera block (5 tiles, F on wave 0 with look-ahead, two barriers per tile) and the back-substitution.
// Operand images of a 9-column tile (row stride 11): the k index of the matrix core runs 0..11 in three steps; the third step's
// lanes with k > 8 read the tile's padding column (column 9, zero in the image and never written), so no load is predicated.
#ifndef VIO_CHAIN_CORE_H
#define VIO_CHAIN_CORE_H
#ifndef CH_DIAG_SKIP
#define CH_DIAG_SKIP 0          // timing experiments only (tools/build_diag.sh <name> -DCH_DIAG_SKIP=n): the named part is left out, the results are wrong
#endif

#ifdef VIO_STAMPS
#define CH_STAMP(slot) do { if (dbg && (tid & 63) == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dbg[slot] = __builtin_amdgcn_s_memtime() - t_start__; } } while (0)
#else
#define CH_STAMP(slot) do { } while (0)
#endif

#ifdef VIO_STAMPS
__device__ unsigned long long *g_ch_dbg = nullptr;
__device__ unsigned long long g_ch_t0 = 0;
#define CH_TSTAMP(cond, slot) do { if ((cond) && g_ch_dbg && (threadIdx.x & 63) == 0) { g_ch_dbg[slot] = __builtin_amdgcn_s_memtime() - g_ch_t0; } } while (0)
#else
#define CH_TSTAMP(cond, slot) do { } while (0)
#endif
struct ChLane {         // per-lane offsets (doubles) inside a tile
    int oA0, oA2;       // A image of a 9-column tile: [row r16][k = g], + 4 for the second step, [row r16][8 or the padding column]
    int oM0, oM2;       // B image of M_e / C image of a 9-column tile: [k = g][r16], + 4 rows per step; third step [8][r16] or M's padding
    int r16, g;
    bool r9;
};

// ---- structural zeros ----
// SC[e][t] (camera tile t x speed-bias block e) is zero in the image unless the IMU factors or the prior couple them: block e reaches
// the poses e-1, e, e+1 (two tiles), the prior's speed-bias block everything it was marginalised against.  The workers, idle during
// level 0, test the 55 tiles (ch_scan_tiles); after barrier 0 every wave derives the tiles that are non-zero WHEN BLOCK e IS
// ELIMINATED — its own and, through the fill, those of the blocks before it in its chain — as one 64-bit mask (bit 5 e + t) in scalar
// registers (ch_eff_mask).  A zero tile has a zero L: its product, its fill, its camera-block terms, its right-hand-side term and its
// back-substitution term are skipped, exactly.
__device__ __forceinline__ void ch_scan_tiles(double *P, int wi, int lane) {
    int *sNZ = (int *)(P + CH_OFF_NZ);
    for (int tile = wi; tile < 55; tile += 14) {
        const double *t = P + CH_OFF_SC + tile * CH_SCSZ;
        bool nz = t[lane] != 0.0 || t[lane + 64] != 0.0;
        if (lane + 128 < CH_SCSZ) nz = nz || t[lane + 128] != 0.0;
        const unsigned long long any = __ballot(nz);
        if (lane == 0) sNZ[tile] = any != 0ull ? 1 : 0;
    }
}
__device__ __forceinline__ unsigned long long ch_eff_mask(const double *P, int lane) {
    const int *sNZ = (const int *)(P + CH_OFF_NZ);
    const unsigned long long nz = __ballot(lane < 55 && sNZ[min(lane, 54)] != 0);
    unsigned long long eff = 0ull;
    unsigned a = 0, b = 0;
    for (int e = 0; e < 5; ++e) {
        a |= (unsigned)(nz >> (5 * e)) & 31u;
        b |= (unsigned)(nz >> (5 * (10 - e))) & 31u;
        eff |= (unsigned long long)a << (5 * e);
        eff |= (unsigned long long)b << (5 * (10 - e));
    }
    eff |= (unsigned long long)(((unsigned)(nz >> 25) & 31u) | a | b) << 25;
    return eff;
}
__device__ __forceinline__ bool ch_bit(unsigned long long eff, int e, int t) { return (eff >> (5 * e + t)) & 1ull; }

// the chain wave's work of one level: F(e) with the rows of SO[e] riding (they come out as L_SO[e] = SO[e] L_ee^-T D_e^-1), the pivots, and
// (upd) SD[n] -= (L D) L^T.  e == 5: F only.  `scr`: 64 doubles of scratch of this wave's own.
// (Until round 5 L_SO was formed behind F as the product (SO M_e) / d on the matrix core and the update followed from registers: three
// products, three reciprocals and a store more on the chain's critical path, 0.7 k ticks a level.)
__device__ __forceinline__ void ch_chain_level(double *P, const ChLane &L, int e, int n, bool upd, int lane, double *scr) {
    double *sD = P + CH_OFF_D;
#ifdef VIO_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    if (e == 5) ch_factor<9, CH_TS, CH_TS>((lds_double *)(P + ch_sd(e)), (lds_double *)(P + CH_OFF_I9), (lds_double *)(P + ch_sm(e)), lane);
    else ch_factor<9, CH_TS, CH_TS, true>((lds_double *)(P + ch_sd(e)), (lds_double *)(P + CH_OFF_I9), (lds_double *)(P + ch_sm(e)), lane,
                                          (lds_double *)(P + ch_so(e)), (lds_double *)scr);
#ifdef VIO_STAMPS
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
    if (lane < 9) sD[e * 16 + lane] = P[ch_sd(e) + lane * (CH_TS + 1)];
    if (e == 5 || !upd) return;
    const double *tt = P + ch_so(e);
    double *td = P + ch_sd(n);
    const double a0 = tt[L.oA0], a1 = tt[L.oA0 + 4], a2 = tt[L.oA2];
    const double p0 = sD[e * 16 + L.g], p1 = sD[e * 16 + L.g + 4], p2 = sD[e * 16 + L.g + 8];
    ps_v4d acc2 = {td[L.oM0], td[L.oM0 + 4 * CH_TS], td[L.oM0 + 8 * CH_TS], 0.0};
    __builtin_amdgcn_sched_barrier(0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 * p0, -a0, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * p1, -a1, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2 * p2, -a2, acc2, 0, 0, 0);
    if (L.r9) { td[L.oM0] = acc2[0]; td[L.oM0 + 4 * CH_TS] = acc2[1]; if (L.g == 0) td[L.oM0 + 8 * CH_TS] = acc2[2]; }
#ifdef VIO_STAMPS
    if (e == 2 && g_ch_dbg && lane == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        g_ch_dbg[224] = ts0 - g_ch_t0; g_ch_dbg[225] = ts1 - g_ch_t0; g_ch_dbg[226] = ts2 - g_ch_t0;
    }
#endif
}
// F on a camera tile (np = 16, or 8 for the last one): M to sM, the pivots to sDp[0 .. np)
// np == 2: the last tile when the extrinsic is fixed (vio_config.ext_fixed, the reference's ESTIMATE_EXTRINSIC = 0): its variables 2..7
// are the extrinsic's — rows without an entry off the diagonal (no extrinsic block in the plan, the prior's masked: d_hs_rest), lambda
// on it — so their six pivots are the diagonal as it stands and their rows of L and M the identity's: two pivots instead of eight.
__device__ __forceinline__ void ch_factor_tile(double *tile, double *sM, double *sDp, int np, int lane) {
    if (np == 16) ch_factor<16, PS_TROW, PS_TROW>((lds_double *)tile, (lds_double *)(dyn_smem + CH_OFF_I16), (lds_double *)sM, lane);
    else if (np == 8) ch_factor<8, PS_TROW, PS_TROW>((lds_double *)tile, (lds_double *)(dyn_smem + CH_OFF_I16), (lds_double *)sM, lane);
    else ch_factor<2, PS_TROW, PS_TROW>((lds_double *)tile, (lds_double *)(dyn_smem + CH_OFF_I16), (lds_double *)sM, lane);
    if (lane < (np == 2 ? 8 : np)) sDp[lane] = tile[lane * (PS_TROW + 1)];
}

// a worker's fused task: L_SC[E][t] = (SC[E][t] M_E) / d (transposed product), stored; with FILL, acc2 (the C image of SC[N][t],
// loaded by the caller) -= (L D) L_SO[E]^T
template <int E, bool FILL>
__device__ __forceinline__ void ch_fused(double *P, const ChLane &L, int t, ps_v4d &acc2) {
    const double *sD = P + CH_OFF_D;
    double *tt = P + ch_sc(E, 0) + t * CH_SCSZ;
    const double *mm = P + ch_sm(E);
    const double a0 = tt[L.oA0], a1 = tt[L.oA0 + 4], a2 = tt[L.oA2];
    const double b0 = mm[L.oM0], b1 = mm[L.oM0 + 4 * CH_TS], b2 = mm[L.oM2];
    const double p0 = sD[E * 16 + L.g], p1 = sD[E * 16 + L.g + 4], p2 = sD[E * 16 + L.g + 8];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (FILL) { const double *so = P + ch_so(E); s0 = so[L.oA0]; s1 = so[L.oA0 + 4]; s2 = so[L.oA2]; }
    __builtin_amdgcn_sched_barrier(0);
    ps_v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b2, a2, acc, 0, 0, 0);
    const double q0 = d_fast_rcp(p0), q1 = d_fast_rcp(p1), q2 = d_fast_rcp(p2);
    const double l0 = d_div(acc[0], p0, q0), l1 = d_div(acc[1], p1, q1);
    const double l2 = (L.g == 0) ? d_div(acc[2], p2, q2) : 0.0, u2 = (L.g == 0) ? acc[2] : 0.0;
    tt[L.oA0] = l0; tt[L.oA0 + 4] = l1; tt[L.oA2] = l2;
    if (FILL) {
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], -s0, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[1], -s1, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(u2, -s2, acc2, 0, 0, 0);
    }
}
__device__ __forceinline__ void ch_ld_c9(const double *tc, const ChLane &L, ps_v4d &acc) {        // C image of a 16 x 9 tile
    acc[0] = tc[L.oM0]; acc[1] = tc[L.oM0 + 4 * CH_TS]; acc[2] = tc[L.oM0 + 8 * CH_TS]; acc[3] = tc[L.oM0 + 12 * CH_TS];
}
__device__ __forceinline__ void ch_st_c9(double *tc, const ChLane &L, const ps_v4d &acc) {
    if (L.r9) { tc[L.oM0] = acc[0]; tc[L.oM0 + 4 * CH_TS] = acc[1]; tc[L.oM0 + 8 * CH_TS] = acc[2]; tc[L.oM0 + 12 * CH_TS] = acc[3]; }
}
// acc (C image of CC(I,J)) -= (L_SC[e][I] D_e) L_SC[e][J]^T
__device__ __forceinline__ void ch_cc_term(const double *P, const ChLane &L, int e, int I, int J, ps_v4d &acc) {
    const double *sD = P + CH_OFF_D + e * 16;
    const double *ta = P + ch_sc(e, 0) + I * CH_SCSZ, *tb = P + ch_sc(e, 0) + J * CH_SCSZ;
    const double a0 = ta[L.oA0], a1 = ta[L.oA0 + 4], a2 = ta[L.oA2];
    const double b0 = tb[L.oA0], b1 = tb[L.oA0 + 4], b2 = tb[L.oA2];
    const double p0 = sD[L.g], p1 = sD[L.g + 4], p2 = sD[L.g + 8];
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 * p0, -b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * p1, -b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2 * p2, -b2, acc, 0, 0, 0);
}
// The camera tiles take the chain's terms LEVEL BY LEVEL, all fifteen of them (round 5; until then only the six tiles camera step 0
// consumes did, the other nine took all eleven blocks' terms "in one go in the U phase before their consumption, when fourteen waves have
// nothing to do while wave 0 factors a tile" — but those U phases then lasted as long as F, 3.3 - 4.3 k ticks of 33 chained products per tile,
// and slowed F, which shares their matrix cores: tools/diag_chain_stamps.py).  During the chain the workers finish a phase in 2.1 k ticks of
// the 3.3 k a level takes: the terms of the previous level's two blocks — two tiles a worker at most, and only those whose L tiles are both
// non-zero — fit in what is left.  A tile is first CONSUMED at the S phase of camera step K = J (I > J) or K = J - 1 (I == J); until then
// additions to it commute, and every tile's accumulation order is: chain blocks 0, 10, 1, 9, ..., 4, 6, 5, then the camera steps.
__device__ __forceinline__ void ch_tile_ij(int task, int &I, int &J) { I = (task >= 1) + (task >= 3) + (task >= 6) + (task >= 10); J = task - I * (I + 1) / 2; }
// camera tile `task` (0..14, row-major in the lower triangle): the terms of level pl (blocks pl and 10 - pl; pl == 5: block 5 alone), those
// whose two L tiles are non-zero
__device__ __forceinline__ void ch_cc_early(double *P, const ChLane &L, int task, int pl, unsigned long long eff) {
    int I, J;
    ch_tile_ij(task, I, J);
    const bool ta = ch_bit(eff, pl, I) && ch_bit(eff, pl, J), tb = pl < 5 && ch_bit(eff, 10 - pl, I) && ch_bit(eff, 10 - pl, J);
    if (!ta && !tb) return;
    double *tc = P + ch_cc(0, 0) + task * PS_TS + L.g * PS_TROW + L.r16;
    ps_v4d acc;
    acc[0] = tc[0]; acc[1] = tc[4 * PS_TROW]; acc[2] = tc[8 * PS_TROW]; acc[3] = tc[12 * PS_TROW];
    if (ta) ch_cc_term(P, L, pl, I, J, acc);
    if (tb) ch_cc_term(P, L, 10 - pl, I, J, acc);
    tc[0] = acc[0]; tc[4 * PS_TROW] = acc[1]; tc[8 * PS_TROW] = acc[2]; tc[12 * PS_TROW] = acc[3];
}
// camera tile `task`: the last three blocks' terms in one go — level 4 (blocks 4, 6), then block 5 (the start of the camera block, below)
__device__ __forceinline__ void ch_cc_last3(double *P, const ChLane &L, int task, unsigned long long eff) {
    int I, J;
    ch_tile_ij(task, I, J);
    const bool t4 = ch_bit(eff, 4, I) && ch_bit(eff, 4, J), t6 = ch_bit(eff, 6, I) && ch_bit(eff, 6, J), t5 = ch_bit(eff, 5, I) && ch_bit(eff, 5, J);
    if (!t4 && !t6 && !t5) return;
    double *tc = P + ch_cc(0, 0) + task * PS_TS + L.g * PS_TROW + L.r16;
    ps_v4d acc;
    acc[0] = tc[0]; acc[1] = tc[4 * PS_TROW]; acc[2] = tc[8 * PS_TROW]; acc[3] = tc[12 * PS_TROW];
    if (t4) ch_cc_term(P, L, 4, I, J, acc);
    if (t6) ch_cc_term(P, L, 6, I, J, acc);
    if (t5) ch_cc_term(P, L, 5, I, J, acc);
    tc[0] = acc[0]; tc[4 * PS_TROW] = acc[1]; tc[8 * PS_TROW] = acc[2]; tc[12 * PS_TROW] = acc[3];
}
// acc (C image of tile (I,J)) -= the terms of the speed-bias blocks whose L tiles I and J are both non-zero, in elimination order
// (0, 10, 1, 9, ..., 4, 6, 5).  The blocks are compacted into a list (4 bits each, scalar); the operands of block i + 2 are requested
// before the products of block i are issued (three register sets: an LDS round trip takes longer than one block's three products).
__device__ __forceinline__ void ch_cc_deferred_terms(const double *P, const ChLane &L, int I, int J, ps_v4d &acc, unsigned long long eff) {
    const double *sD = P + CH_OFF_D;
    const double *pa = P + ch_sc(0, 0) + I * CH_SCSZ, *pb = P + ch_sc(0, 0) + J * CH_SCSZ;
    unsigned long long list = 0ull;
    int n = 0;
    for (int i = 0; i < 11; ++i) {
        const int e = (i == 10) ? 5 : ((i & 1) ? 10 - (i >> 1) : (i >> 1));
        if (ch_bit(eff, e, I) && ch_bit(eff, e, J)) { list |= (unsigned long long)e << (4 * n); ++n; }
    }
    if (n == 0) return;
    double a[3][3], b[3][3], p[3][3];
#define CH_DEF_E(i) ((int)((list >> (4 * (i))) & 15ull))
#define CH_DEF_LD(S, E) do { const int e__ = (E); const double *ta__ = pa + e__ * 5 * CH_SCSZ, *tb__ = pb + e__ * 5 * CH_SCSZ;  \
        a[S][0] = ta__[L.oA0]; a[S][1] = ta__[L.oA0 + 4]; a[S][2] = ta__[L.oA2];                                                \
        b[S][0] = tb__[L.oA0]; b[S][1] = tb__[L.oA0 + 4]; b[S][2] = tb__[L.oA2];                                                \
        p[S][0] = sD[e__ * 16 + L.g]; p[S][1] = sD[e__ * 16 + L.g + 4]; p[S][2] = sD[e__ * 16 + L.g + 8]; } while (0)
#define CH_DEF_MM(S) do {                                                                                                       \
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S][0] * p[S][0], -b[S][0], acc, 0, 0, 0);                                  \
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S][1] * p[S][1], -b[S][1], acc, 0, 0, 0);                                  \
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S][2] * p[S][2], -b[S][2], acc, 0, 0, 0); } while (0)
    CH_DEF_LD(0, CH_DEF_E(0));
    if (n > 1) CH_DEF_LD(1, CH_DEF_E(1));
    for (int i = 0; i < n; i += 3) {
        if (i + 2 < n) CH_DEF_LD(2, CH_DEF_E(i + 2));
        __builtin_amdgcn_sched_barrier(0);
        CH_DEF_MM(0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < n) {
            if (i + 3 < n) CH_DEF_LD(0, CH_DEF_E(i + 3));
            __builtin_amdgcn_sched_barrier(0);
            CH_DEF_MM(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i + 2 < n) {
            if (i + 4 < n) CH_DEF_LD(1, CH_DEF_E(i + 4));
            __builtin_amdgcn_sched_barrier(0);
            CH_DEF_MM(2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef CH_DEF_E
#undef CH_DEF_LD
#undef CH_DEF_MM
}
// y_C -= L_SC[E] w_E for the rows of this lane (80 rows: lane, and 64 + lane for lane < 16)
template <int E>
__device__ __forceinline__ void ch_yc_term(double *P, int lane) {
    double *sY = P + CH_OFF_Y;
    double w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = sY[E * 16 + k];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = pass * 64 + lane;
        if (i < 80) {
            const double *l = P + ch_sc(E, 0) + (i >> 4) * CH_SCSZ + (i & 15) * CH_TS;
            double y = sY[CH_YC + i], lv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) lv[k] = l[k];
            __builtin_amdgcn_sched_barrier(0);       // (every operand requested before the chain starts: left alone, each step waits for its own load)
#pragma unroll
            for (int k = 0; k < 9; ++k) y = fma(-lv[k], w[k], y);
            sY[CH_YC + i] = y;
        }
    }
}
// the right-hand side of phase LEV (one wave): w_e = M_e^T y_e for the level's blocks, y_succ -= L_SO[e] w_e, and y_C -= L_SC w of the
// previous level's blocks
template <int LEV>
__device__ __forceinline__ void ch_rhs_phase(double *P, int lane) {
    double *sY = P + CH_OFF_Y;
    constexpr int eA = LEV, eB = 10 - LEV;
    const int k = lane & 15, half = lane >> 4;           // half 0: chain A, half 1: chain B
    const int e = half == 0 ? eA : eB;
    const bool on = k < 9 && (half == 0 || (half == 1 && LEV < 5));
    double w = 0.0;
    if (on) {
        const double *m = P + ch_sm(e) + k;
        double yv[9], mv[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) { yv[j] = sY[e * 16 + j]; mv[j] = m[j * CH_TS]; }
        __builtin_amdgcn_sched_barrier(0);           // (operands first, then the chain — see ch_yc_term)
#pragma unroll
        for (int j = 0; j < 9; ++j) w = fma(yv[j], mv[j], w);        // column k of M_e (zero below its diagonal)
    }
    __builtin_amdgcn_wave_barrier();
    if (on) sY[e * 16 + k] = w;
    __builtin_amdgcn_wave_barrier();
    if (LEV < 5) {
        if (LEV < 4) {
            const int n = half == 0 ? eA + 1 : eB - 1;
            if (on) {
                const double *l = P + ch_so(e) + k * CH_TS;
                double y = sY[n * 16 + k], lv[9], wv[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) { lv[j] = l[j]; wv[j] = sY[e * 16 + j]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 9; ++j) y = fma(-lv[j], wv[j], y);
                sY[n * 16 + k] = y;
            }
        } else if (lane < 9) {          // both chains end in block 5
            const double *la = P + ch_so(4) + k * CH_TS, *lb = P + ch_so(6) + k * CH_TS;
            double y = sY[5 * 16 + k], lv[18], wv[18];
#pragma unroll
            for (int j = 0; j < 9; ++j) { lv[j] = la[j]; wv[j] = sY[4 * 16 + j]; lv[9 + j] = lb[j]; wv[9 + j] = sY[6 * 16 + j]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 18; ++j) y = fma(-lv[j], wv[j], y);
            sY[5 * 16 + k] = y;
        }
    }
}
// y_C -= L_SC[e] w_e for the 16 rows of camera tile t (lanes 0..15)
__device__ __forceinline__ void ch_yc_tile(double *P, int e, int t, int lane) {
    if (lane < 16) {
        double *sY = P + CH_OFF_Y;
        const double *l = P + ch_sc(e, 0) + t * CH_SCSZ + lane * CH_TS;
        double y = sY[CH_YC + 16 * t + lane], lv[9], wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { lv[k] = l[k]; wv[k] = sY[e * 16 + k]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 9; ++k) y = fma(-lv[k], wv[k], y);
        sY[CH_YC + 16 * t + lane] = y;
    }
}

// what the workers (waves 2..15, wi = wave - 2) do after barrier LEV
// CAM = false (the pre-elimination of the GN loop, ch_chain_pre: the camera block and its right-hand side are not there yet): the terms
// of CC and y_C are left to the kernel that has them (ch_camera_solve<true> applies them in the same order)
template <int LEV, bool CAM = true>
__device__ __forceinline__ void ch_worker_phase(double *P, const ChLane &L, int wi, int lane, unsigned long long eff) {
    constexpr int eA = LEV, eB = 10 - LEV;
#ifdef CH_DIAG_IDLE_SIMDS
    if ((((wi + 2) & 3) == 0) || (CH_DIAG_IDLE_SIMDS > 1 && ((wi + 2) & 3) == 1) || CH_DIAG_IDLE_SIMDS > 3) return;      // timing experiment only: wrong results
#endif
    if (wi == 13) { ch_rhs_phase<LEV>(P, lane); return; }
    int first_cc, busy;
    if (LEV < 4) {
        busy = wi < 10;
        if (wi < 5) {
            if (ch_bit(eff, eA, wi)) {
                ps_v4d acc2;
                ch_ld_c9(P + ch_sc(eA + 1, 0) + wi * CH_SCSZ, L, acc2);
                ch_fused<eA, true>(P, L, wi, acc2);
                ch_st_c9(P + ch_sc(eA + 1, 0) + wi * CH_SCSZ, L, acc2);
            }
        } else if (wi < 10) {
            if (ch_bit(eff, eB, wi - 5)) {
                ps_v4d acc2;
                ch_ld_c9(P + ch_sc(eB - 1, 0) + (wi - 5) * CH_SCSZ, L, acc2);
                ch_fused<eB, true>(P, L, wi - 5, acc2);
                ch_st_c9(P + ch_sc(eB - 1, 0) + (wi - 5) * CH_SCSZ, L, acc2);
            }
        }
        first_cc = wi >= 10 ? wi - 10 : wi + 3;           // the three idle workers take the first tasks, and 13 / 14 after them
    } else {
        busy = wi < 5;
        if (wi < 5) {
            if (LEV == 4) {                               // both chains fill SC[5][t]: one wave, chain A's term first
                if (ch_bit(eff, 4, wi) || ch_bit(eff, 6, wi)) {
                    ps_v4d acc2;
                    ch_ld_c9(P + ch_sc(5, 0) + wi * CH_SCSZ, L, acc2);
                    if (ch_bit(eff, 4, wi)) ch_fused<4, true>(P, L, wi, acc2);
                    if (ch_bit(eff, 6, wi)) ch_fused<6, true>(P, L, wi, acc2);
                    ch_st_c9(P + ch_sc(5, 0) + wi * CH_SCSZ, L, acc2);
                }
            } else if (ch_bit(eff, 5, wi)) {
                ps_v4d dummy = {0.0, 0.0, 0.0, 0.0};
                ch_fused<5, false>(P, L, wi, dummy);
            }
        }
        first_cc = wi >= 5 ? wi - 5 : wi + 8;             // eight idle workers: tasks 0..7 and 8..14 after them
    }
    if (CAM && LEV > 0 && wi < 5) {
        // the right-hand side rows of camera tile wi: the terms of the previous level's blocks (their w came out one phase ago)
        if (ch_bit(eff, LEV - 1, wi)) ch_yc_tile(P, LEV - 1, wi, lane);
        if (ch_bit(eff, 11 - LEV, wi)) ch_yc_tile(P, 11 - LEV, wi, lane);
    }
    if (CAM && LEV > 0) {
        // the camera-block terms of the previous level, all fifteen tiles: the workers without a fused task take two each, the busy ones
        // one each behind their task
        if (LEV < 4) {
            if (wi >= 10 && wi < 13) { ch_cc_early(P, L, 2 * (wi - 10), LEV - 1, eff); ch_cc_early(P, L, 2 * (wi - 10) + 1, LEV - 1, eff); }   // tiles 0..5
            else if (wi < 9) ch_cc_early(P, L, 6 + wi, LEV - 1, eff);                                                                         // tiles 6..14
        } else if (wi < 5) {
            ch_cc_early(P, L, 10 + wi, LEV - 1, eff);                                                                        // tiles 10..14
        } else if (wi < 13) {
            ch_cc_early(P, L, wi - 5, LEV - 1, eff);                                                                         // tiles 0..7
            if (wi < 7) ch_cc_early(P, L, wi + 3, LEV - 1, eff);                                                             // tiles 8, 9
        }
    }
    (void)first_cc;
    (void)busy;
}

// What follows the solve for the camera variables (trial poses, pair table) need not wait for the speed-bias part of the solution:
// mid1(lane) is called by wave 13 once the camera part is out (sX[CH_YC ..]; phase Y: it has nothing else to do), mid2(index, lane) one
// barrier later by the fourteen waves that do not walk the chains (phase Z), index 0..13.
__device__ __forceinline__ ChLane ch_lane(int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    ChLane L;
    L.r16 = r16; L.g = g; L.r9 = r16 < 9;
    L.oA0 = r16 * CH_TS + g; L.oA2 = r16 * CH_TS + (g == 0 ? 8 : 9);
    L.oM0 = g * CH_TS + r16; L.oM2 = (g == 0) ? 8 * CH_TS + r16 : 8 * CH_TS + 9;
    return L;
}

// The speed-bias chain: six levels (all 16 waves; ends behind a barrier).  Returns the mask of the SC tiles that are non-zero when their
// block is eliminated.  CAM: the camera block and the camera part of the right-hand side are in the image and take the chain's terms
// level by level (the fused solve); without them (the pre-elimination of the GN loop) the chain leaves L_SC, L_SO, M_e, the pivots and
// w_e = M_e^T y_e, and whoever has the camera block applies the terms (ch_camera_solve<true>: the same operations in the same order).
template <bool CAM>
__device__ __forceinline__ unsigned long long ch_chain_elimination(double *P, const int tid, const ChLane &L, unsigned long long *dbg, unsigned long long t_start__,
                                                                   const bool scanned = false) {
    const int lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = L.g;
    const bool r9 = L.r9;
    double *sD = P + CH_OFF_D;
    (void)dbg; (void)t_start__;
    unsigned long long eff = 0ull;
#ifndef CH_NO_PRIO
    // the waves on the critical path ask the instruction arbiter for priority over the waves of their SIMD: the chain waves (F, the
    // level's products) and the right-hand-side wave, whose dependent fp64 chains otherwise queue behind the workers' matrix-core streams
    if (uwave < 2) __builtin_amdgcn_s_setprio(3);
    else if (uwave == 15) __builtin_amdgcn_s_setprio(2);
#endif
    if (uwave < 2) {
        // the two chain waves: wave 0 blocks 0..4 and then 5, wave 1 blocks 10..6 (one copy of the code, the block a run-time value)
        const int dir = uwave == 0 ? 1 : -1;
        int e = uwave == 0 ? 0 : 10;
        for (int lev = 0; lev < 6; ++lev, e += dir) {
            if (lev < 5 || uwave == 0) {
                if (lev == 5) {
                    // chain B's term of SD[5] (chain A's went in from registers)
                    double *td = P + ch_sd(5);
                    const double *so = P + ch_so(6);
                    const double a0 = so[L.oA0], a1 = so[L.oA0 + 4], a2 = so[L.oA2];
                    const double p0 = sD[6 * 16 + g], p1 = sD[6 * 16 + g + 4], p2 = sD[6 * 16 + g + 8];
                    ps_v4d acc = {td[L.oM0], td[L.oM0 + 4 * CH_TS], td[L.oM0 + 8 * CH_TS], 0.0};
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0 * p0, -a0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * p1, -a1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2 * p2, -a2, acc, 0, 0, 0);
                    if (r9) { td[L.oM0] = acc[0]; td[L.oM0 + 4 * CH_TS] = acc[1]; if (g == 0) td[L.oM0 + 8 * CH_TS] = acc[2]; }
                }
                ch_chain_level(P, L, e, e + dir, !(uwave == 1 && lev == 4), lane, P + CH_OFF_X + 64 * uwave);
                if (uwave == 0) CH_STAMP(64 + 4 * lev);
            }
            CH_STAMP(256 + 16 * lev + uwave);              // (diagnostic build: when each wave reaches barrier `lev`: tools/diag_chain_stamps.py)
            __syncthreads();
            if (uwave == 0) CH_STAMP(65 + 4 * lev);
        }
        eff = ch_eff_mask(P, lane);
        if (uwave == 0) CH_STAMP(233);
        if (CAM) {
            // phase 5a (see the workers'): wave 0 forms L_SC[5][0] itself — the tile its CC(0,0) waits for —, wave 1 gives CC(0,0) the terms
            // of level 4
            if (uwave == 0) { if (ch_bit(eff, 5, 0)) { ps_v4d dummy = {0.0, 0.0, 0.0, 0.0}; ch_fused<5, false>(P, L, 0, dummy); } }
            else ch_cc_early(P, L, 0, 4, eff);
            if (uwave == 0) CH_STAMP(234);
            if (uwave == 1) CH_STAMP(235);
        }
        __syncthreads();
    } else {
        const int wi = uwave - 2;
        if (!scanned) ch_scan_tiles(P, wi, lane);                  // (nothing else to do during level 0; `scanned`: the copy-in noted the flags)
        CH_STAMP(256 + uwave);
        __syncthreads();                                           // barrier 0: level 0 is out
        eff = ch_eff_mask(P, lane);
        ch_worker_phase<0, CAM>(P, L, wi, lane, eff); if (uwave == 2) CH_STAMP(112);
        CH_STAMP(256 + 16 + uwave);
        __syncthreads();
        ch_worker_phase<1, CAM>(P, L, wi, lane, eff); if (uwave == 2) CH_STAMP(113);
        CH_STAMP(256 + 32 + uwave);
        __syncthreads();
        ch_worker_phase<2, CAM>(P, L, wi, lane, eff); if (uwave == 2) CH_STAMP(114); CH_STAMP(140 + uwave);
        CH_STAMP(256 + 48 + uwave);
        __syncthreads();
        ch_worker_phase<3, CAM>(P, L, wi, lane, eff); if (uwave == 2) CH_STAMP(115);
        CH_STAMP(256 + 64 + uwave);
        __syncthreads();
        ch_worker_phase<4, CAM>(P, L, wi, lane, eff); if (uwave == 2) CH_STAMP(116); CH_STAMP(180 + uwave);
        CH_STAMP(256 + 80 + uwave);
        __syncthreads();
        if (CAM) {
            // Phase 5a, a short one: the five L_SC[5][t] (t = 0 on wave 0), w_5, and the right-hand side's terms of level 4.  Everything else
            // that used to sit between the last level and the camera block — the camera tiles' terms of level 4 and of block 5, 3.7 k ticks of
            // matrix-core work in front of F(0) — runs beside F(0) (ch_camera_solve's first phase): every wave one tile, three terms.
            if (wi >= 1 && wi < 5) { if (ch_bit(eff, 5, wi)) { ps_v4d dummy = {0.0, 0.0, 0.0, 0.0}; ch_fused<5, false>(P, L, wi, dummy); } }
            else if (wi == 13) ch_rhs_phase<5>(P, lane);
            else if (wi == 0 || (wi >= 5 && wi < 9)) {
                const int t = wi == 0 ? 0 : wi - 4;
                if (ch_bit(eff, 4, t)) ch_yc_tile(P, 4, t, lane);
                if (ch_bit(eff, 6, t)) ch_yc_tile(P, 6, t, lane);
            }
        } else {
            ch_worker_phase<5, CAM>(P, L, wi, lane, eff);
        }
        if (uwave == 2) CH_STAMP(117); CH_STAMP(200 + uwave);
        __syncthreads();
    }
    if (uwave == 0) CH_STAMP(87);
    return eff;
}

// The camera block (5 tiles: 16, 16, 16, 16, 8) and the back-substitution.  SPLIT: the chain was eliminated by another workgroup
// (ch_chain_pre: L_SC, L_SO, M_e, pivots and w_e are in the image, `eff` came with them): the terms the chain waves apply level by level
// in the fused solve — to the six camera tiles step 0 consumes and to y_C — are applied here in one go, in the chain's order
// (0, 10, 1, 9, ..., 4, 6, 5: the same accumulations, bit for bit).
template <bool SPLIT, typename Mid1, typename Mid2>
__device__ __forceinline__ void ch_camera_solve(double *P, const int tid, const ChLane &L, const unsigned long long eff, Mid1 mid1, Mid2 mid2,
                                                unsigned long long *dbg, unsigned long long t_start__, const bool ext_trivial = false) {
    const int lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = L.r16, g = L.g;
    const bool r9 = L.r9;
    double *sD = P + CH_OFF_D, *sX = P + CH_OFF_X, *sY = P + CH_OFF_Y, *sMc = P + CH_OFF_MC;
    (void)dbg; (void)t_start__;
    const int lofs = r16 * PS_TROW + g;      // A image of a 16 x 17 tile: row r16, k = g + 4q
    const int cofs = g * PS_TROW + r16;      // C / B image: row g + 4v, column r16
    if (SPLIT) {
#ifndef CH_NO_PRIO
        if (uwave == 0) __builtin_amdgcn_s_setprio(3);
        else if (uwave == 15) __builtin_amdgcn_s_setprio(2);
#endif
        if (uwave < 15) {
            // every camera tile takes all eleven blocks' terms now, tile (0,0) on the wave that factors it next
            int I, J;
            ch_tile_ij(uwave, I, J);
            double *tc = P + ch_cc(I, J) + cofs;
            ps_v4d acc;
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = tc[4 * PS_TROW * v];
            ch_cc_deferred_terms(P, L, I, J, acc, eff);
#pragma unroll
            for (int v = 0; v < 4; ++v) tc[4 * PS_TROW * v] = acc[v];
            if (uwave == 0) ch_factor_tile(P + ch_cc(0, 0), sMc, sD + CH_YC, 16, lane);
            else if (uwave >= 10) {
                // y_C -= L_SC[e] w_e for the rows of camera tile t, block after block in the chain's order (the waves of the last tile row:
                // their tiles couple to few blocks)
                const int t = uwave - 10;
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    const int e = (i == 10) ? 5 : ((i & 1) ? 10 - (i >> 1) : (i >> 1));
                    if (ch_bit(eff, e, t)) ch_yc_tile(P, e, t, lane);
                }
            }
        }
    } else
    if (uwave == 0) {
        // CC(0,0) -= (L D) L^T of block 5 first: it is what F(0) waits for
        double *tc = P + ch_cc(0, 0) + cofs;
        ps_v4d acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] = tc[4 * PS_TROW * v];
        if (ch_bit(eff, 5, 0)) ch_cc_term(P, L, 5, 0, 0, acc);
#pragma unroll
        for (int v = 0; v < 4; ++v) tc[4 * PS_TROW * v] = acc[v];
        ch_factor_tile(P + ch_cc(0, 0), sMc, sD + CH_YC, 16, lane);
    } else if (uwave == 15) {
        ch_yc_term<5>(P, lane);
    } else {
        // tiles 1..14: the terms of level 4 and of block 5.  Waves 4, 8 and 12 share wave 0's SIMD: they sit it out (F(0) is the critical
        // path), their tiles go to waves 1, 2, 3 as a second one
#ifndef CH_PREAMBLE_ALL_WAVES
        if ((uwave & 3) != 0) {
            ch_cc_last3(P, L, uwave, eff);
            if (uwave <= 3) ch_cc_last3(P, L, 4 * uwave, eff);
        }
#else
        ch_cc_last3(P, L, uwave, eff);
#endif
    }
    if (uwave == 0) CH_STAMP(88);
    __syncthreads();
    if (uwave == 0) CH_STAMP(89);
    for (int K = 0; K < 5; ++K) {
        const int nk = 4 - K;                                   // tiles below the diagonal
        const int d0 = CH_YC + 16 * K;
        // ---- S phase ----
        if (uwave == 0 && nk > 0) {
            double *tt = P + ch_cc(K + 1, K);
            double *td = P + ch_cc(K + 1, K + 1) + cofs;
            const double *pd = sD + d0 + g;
            double av[4], bv[4], pv[4], lv[4], qv[4];
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = tt[lofs + 4 * q]; bv[q] = sMc[cofs + 4 * PS_TROW * q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { pv[q] = pd[4 * q]; acc2[q] = td[4 * PS_TROW * q]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[q], av[q], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) qv[q] = d_fast_rcp(pv[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { lv[q] = d_div(acc[q], pv[q], qv[q]); tt[lofs + 4 * q] = lv[q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[q], -lv[q], acc2, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) td[4 * PS_TROW * q] = acc2[q];
        } else if (uwave >= 1 && uwave < nk) {
            double *tt = P + ch_cc(K + 1 + uwave, K);
            double av[4], bv[4];
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = tt[lofs + 4 * q]; bv[q] = sMc[cofs + 4 * PS_TROW * q]; }
            const double dd = sD[d0 + r16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
            const double rr = d_fast_rcp(dd);
#pragma unroll
            for (int q = 0; q < 4; ++q) tt[cofs + 4 * PS_TROW * q] = d_div(acc[q], dd, rr);
        } else if (uwave == 14) {
            // M_K takes the diagonal tile's place (the back-substitution multiplies by it; the pivots are in sD)
            double mk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mk[q] = sMc[(g + 4 * q) * PS_TROW + r16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) P[ch_cc(K, K) + (g + 4 * q) * PS_TROW + r16] = mk[q];
        } else if (uwave == 15) {
            double y = 0.0;
            if (lane < 16) {
                double yv[16], mv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) { yv[j] = sY[d0 + j]; mv[j] = sMc[j * PS_TROW + lane]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; ++j) y = fma(yv[j], mv[j], y);
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) sY[d0 + lane] = y;
        }
        if (uwave == 0) CH_STAMP(90 + 4 * K);
        __syncthreads();
        if (uwave == 0) CH_STAMP(91 + 4 * K);
        if (nk == 0) break;
        // ---- U (+ F(K+1) on wave 0): tiles 1 .. ntile-1 of the trailing triangle in row-major order (tile 0 = (K+1,K+1) is wave 0's),
        //      then the right-hand side ----
        if (uwave == 0) {
            if (K + 1 == 4) {
                // the last tile has 8 variables: M starts as the identity, rows / columns 8..15 stay that way
                for (int i = lane; i < PS_TS; i += 64) sMc[i] = (i / PS_TROW == i % PS_TROW) ? 1.0 : 0.0;
                ch_factor_tile(P + ch_cc(4, 4), sMc, sD + CH_YC + 64, ext_trivial ? 2 : 8, lane);
            } else {
                ch_factor_tile(P + ch_cc(K + 1, K + 1), sMc, sD + d0 + 16, 16, lane);
            }
        } else if (CH_DIAG_SKIP != 5) {
            const int ntile = nk * (nk + 1) / 2;
            const int nitem = ntile + 1;
            // (waves 4, 8 and 12 share wave 0's SIMD — its vector ALU and its matrix core: they sit the phase out, F(K+1) is the critical path
            // and the ten tasks of the widest step fit the other twelve waves in one round)
            const bool same_simd = (uwave & 3) == 0;
            const int widx = uwave - 1 - (uwave > 4) - (uwave > 8) - (uwave > 12);          // 0..11 over waves 1,2,3,5,6,7,9,10,11,13,14,15
            for (int t = same_simd ? nitem : 1 + widx; t < nitem; t += 12) {
                if (t < ntile) {
                    const int ii = (t >= 1) + (t >= 3) + (t >= 6), jj = t - ii * (ii + 1) / 2;
                    const int ti = K + 1 + ii, tj = K + 1 + jj;
                    const double *ta = P + ch_cc(ti, K) + lofs, *tb = P + ch_cc(tj, K) + lofs;
                    double *tc = P + ch_cc(ti, tj) + cofs;
                    double av[4], bv[4], dk[4];
                    ps_v4d acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { av[q] = ta[4 * q]; bv[q] = tb[4 * q]; acc[q] = tc[4 * PS_TROW * q]; dk[q] = sD[d0 + g + 4 * q]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q] * dk[q], -bv[q], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) tc[4 * PS_TROW * q] = acc[q];
                } else {
                    // y_I -= L_IK w_K for the rows below
                    const int c = 16 * (K + 1) + lane;
                    if (c < 80) {
                        const double *l = P + ch_cc(c >> 4, K) + (c & 15) * PS_TROW;
                        double y = sY[CH_YC + c], lv[16], wv[16];
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) { lv[kk] = l[kk]; wv[kk] = sY[d0 + kk]; }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) y = fma(-lv[kk], wv[kk], y);
                        sY[CH_YC + c] = y;
                    }
                }
            }
            if (uwave == 2) CH_STAMP(128 + K);
        }
        if (uwave == 0) CH_STAMP(92 + 4 * K);
        __syncthreads();
        if (uwave == 0) CH_STAMP(93 + 4 * K);
    }
    if (uwave == 0) CH_STAMP(110);

    // ================= back-substitution =================
    // x_col = M_col (w_col / d_col - sum_{J after col} L_{J,col}^T x_J).
    //   phase X   wave 15 alone resolves the camera tiles from the bottom, no barriers: 16-lane row K of the wave owns tile K (tile 4 first,
    //             by row 3), x_J goes through LDS to the other rows (one write + one read per tile, in order inside a wave) and into the
    //             16-term products by DPP.  Meanwhile wave e < 11 (owner of speed-bias block e) forms G_e = M_e L_SO[e]^T in the place of
    //             L_SO[e] (only it reads or writes that tile now): x_e = M_e v_e - G_e x_succ(e).
    //   phase Y   the owners: v_e = w_e / d_e - sum_t L_SC[e][t]^T x_C[t] (lane = (k, rows q mod 4), the quarters added in a fixed order),
    //             g_e = M_e v_e (x_5 = g_5); waves 11 / 12 request the rows of G of their chain.
    //   phase Z   waves 11 / 12 walk the two chains x_e = g_e - G_e x_succ(e) (e = 4..0, 6..10) out of registers, no barriers.
 {
#ifndef CH_NO_PRIO
        if (uwave >= 11 && uwave <= 13) __builtin_amdgcn_s_setprio(2);      // the chain walkers and the wave of the trial poses
#endif
        const int k9 = min(r16, 8);
        if (uwave == 15 && CH_DIAG_SKIP != 3) {
            const int K = g;                                                         // the tile of this 16-lane row (0..3)
            double mw[16], ucol[16];
            const double dd = sD[CH_YC + 16 * K + r16];
            const double own_v = d_div(sY[CH_YC + 16 * K + r16], dd, d_fast_rcp(dd));
            double acc = 0.0;
            {   // x of tile 4, by row 3
                const double *m4 = P + ch_cc(4, 4) + r16 * PS_TROW;
#pragma unroll
                for (int j = 0; j < 16; ++j) mw[j] = m4[j];
                const double d4 = sD[CH_YC + 64 + r16];
                const double v4 = d_div(sY[CH_YC + 64 + r16], d4, d_fast_rcp(d4));
                double x;
                PS_DOT16(x, 0.0, v4, mw);
                if (g == 3) sX[CH_YC + 64 + r16] = x;
                __builtin_amdgcn_wave_barrier();         // (other lanes read it: LDS is in order inside a wave, the compiler must keep it so)
                asm volatile("" ::: "memory");
            }
            {
                const double *src = P + ch_cc(K, K) + r16 * PS_TROW;                 // row r16 of M_K
#pragma unroll
                for (int j = 0; j < 16; ++j) mw[j] = src[j];
            }
#pragma unroll
            for (int J = 4; J >= 1; --J) {
                // column r16 of L(J, K) for the rows K < J (the others read a valid tile and discard)
                const double *src = P + ch_cc(J, min(K, J - 1)) + r16;
#pragma unroll
                for (int r = 0; r < 16; ++r) ucol[r] = src[r * PS_TROW];
                const double xj = sX[CH_YC + 16 * J + r16];
                PS_DOT16(acc, acc, xj, ucol);
                if (J >= 1) {
                    double x;
                    PS_DOT16(x, 0.0, own_v - acc, mw);
                    if (K == J - 1) sX[CH_YC + 16 * K + r16] = x;                    // tile J - 1 is complete
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        } else if (uwave < 11 && uwave != 5 && CH_DIAG_SKIP != 4) {
            const int e = uwave;
            const double *m = P + ch_sm(e) + k9 * CH_TS, *so = P + ch_so(e);
            double gk[3], mv[9], sv[3][9];
#pragma unroll
            for (int i = 0; i < 9; ++i) mv[i] = m[i];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int j = min(g + 4 * v, 8);
#pragma unroll
                for (int i = 0; i < 9; ++i) sv[v][i] = so[j * CH_TS + i];
            }
            __builtin_amdgcn_sched_barrier(0);       // (operands first: left alone, every step of a chain waits for its own load)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) sum = fma(mv[i], sv[v][i], sum);
                gk[v] = sum;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r9) {
                double *sg = P + ch_so(e) + k9 * CH_TS;
                sg[g] = gk[0]; sg[g + 4] = gk[1];
                if (g == 0) sg[8] = gk[2];
            }
        }
        if (uwave == 0) CH_STAMP(111);
        if (uwave == 15) CH_STAMP(170);
        __syncthreads();
        // ---- phases Y and Z (the two chain waves apart: their registers hold the rows of G) ----
        if ((uwave == 11 || uwave == 12) && CH_DIAG_SKIP != 4) {
            double gc[5][9], vc[5];
#pragma unroll
            for (int s2 = 0; s2 < 5; ++s2) {
                const int eb = (uwave == 11) ? 4 - s2 : 6 + s2;
                const double *src = P + ch_so(eb) + k9 * CH_TS;
#pragma unroll
                for (int j = 0; j < 9; ++j) gc[s2][j] = src[j];
            }
            __syncthreads();
            if (lane < 16) {
                double xs = sX[5 * 16 + k9];
#pragma unroll
                for (int s2 = 0; s2 < 5; ++s2) vc[s2] = sX[((uwave == 11) ? 4 - s2 : 6 + s2) * 16 + k9];
#pragma unroll
                for (int s2 = 0; s2 < 5; ++s2) {
                    const int eb = (uwave == 11) ? 4 - s2 : 6 + s2;
                    double t;
                    CH_DOT9(t, 0.0, xs, gc[s2]);
                    const double xn = vc[s2] - t;
                    if (r9) sX[eb * 16 + r16] = xn;
                    xs = xn;
                }
            }
            if (uwave == 11) CH_STAMP(175);
        } else {
            if (uwave == 13) {
                mid1(lane);
                CH_STAMP(171);
            } else if (uwave < 11 && CH_DIAG_SKIP != 4) {
                const int e = uwave;
                double sacc = 0.0, lv[5][4], xv[5][4];
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    if (ch_bit(eff, e, t)) {
                        const double *src = P + ch_sc(e, t) + g * CH_TS + k9;
                        const double *xj = sX + CH_YC + 16 * t + g;
#pragma unroll
                        for (int v = 0; v < 4; ++v) { lv[t][v] = src[4 * v * CH_TS]; xv[t][v] = xj[4 * v]; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    if (ch_bit(eff, e, t)) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) sacc = fma(lv[t][v], xv[t][v], sacc);
                    }
                }
                double *scr = P + CH_OFF_SD;           // (the diagonal blocks of the chain are dead by now)
                scr[e * 64 + lane] = sacc;
                __builtin_amdgcn_wave_barrier();
                if (lane < 16) {
                    const double c = ((scr[e * 64 + r16] + scr[e * 64 + 16 + r16]) + scr[e * 64 + 32 + r16]) + scr[e * 64 + 48 + r16];
                    const double dd = sD[e * 16 + k9];
                    const double v = d_div(sY[e * 16 + k9], dd, d_fast_rcp(dd)) - c;
                    const double *src = P + ch_sm(e) + k9 * CH_TS;
                    double m9[9], gv;
#pragma unroll
                    for (int j = 0; j < 9; ++j) m9[j] = (j < k9) ? 0.0 : src[j];
                    CH_DOT9(gv, 0.0, v, m9);
                    if (r9) sX[e * 16 + r16] = gv;
                }
            }
            if (uwave == 0) CH_STAMP(172);
            __syncthreads();
            if (uwave == 0) CH_STAMP(173);
            mid2(uwave < 11 ? uwave : uwave - 2, lane);
            if (uwave == 0) CH_STAMP(174);
        }
        if (uwave == 0) CH_STAMP(63);
        d_lds_barrier();              // (mid2's stores — the pair table of the trial states — drain under what follows: nobody reads them in this kernel)
    }
}

// the whole solve on one workgroup's image (k_pose_solve_c outside the GN loop's split form, the diagnostic entry)
template <typename Mid1, typename Mid2>
__device__ __forceinline__ void ch_factor_solve(double *P, const int tid, Mid1 mid1, Mid2 mid2, unsigned long long *dbg = nullptr, const bool ext_trivial = false,
                                                const bool scanned = false) {
    unsigned long long t_start__ = 0ull;
#ifdef VIO_STAMPS
    t_start__ = __builtin_amdgcn_s_memtime();
    if (tid == 0) { g_ch_dbg = dbg; g_ch_t0 = t_start__; }
#endif
    const ChLane L = ch_lane(tid & 63);
    const unsigned long long eff = ch_chain_elimination<true>(P, tid, L, dbg, t_start__, scanned);
    ch_camera_solve<false>(P, tid, L, eff, mid1, mid2, dbg, t_start__, ext_trivial);
}
#endif
