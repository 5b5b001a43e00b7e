// vio_kernels_xyz.h — the XYZ-landmark variants of the per-item kernels (included by vio_kernels.hip).
//
// Landmarks are world points (VertexPointXYZ, VM/include/backend/vertex_point_xyz.h:16) observed through
// EdgeReprojectionXYZ (VM/src/backend/edge_reprojection.cc:130-180): vertices (landmark, pose), the camera extrinsic a
// constant of the edge.  Every landmark owns a 3x3 block of Hmm, inverted without damping (problem.cc:419-425), three
// columns of Hpm per observing pose, and its Schur term is  Hpm Hmm^-1 Hmp = W H^-1 W^T  with W (6K x 3).
//
//   k_linearize_xyz   one workgroup of 1024 threads per item = up to G landmarks seen from the same K frames
//     phase 0    thread k < K: the frame's camera map  p_c = A_k p_w + d_k,  A_k = ric^T R_k^T,  d_k = -ric^T (R_k^T P_k + tic)
//     phase 1    thread per observation (a wave holds observations of ONE frame index k): residual, J_feature (2x3),
//                J_pose (2x6), robust weight; the whitened pose rows L J_pose go to sRows (the operand of the direct
//                products), W_k = (L J_pose)^T (L J_feature) to the landmark record, the observation's terms of H_ll and
//                b_l to sAux; the pose part of b is summed inside the wave (one partial per wave)
//     phase 1.5  thread per landmark: H_ll, b_l, H_ll^-1 (partial-pivot LU, as Eigen's dynamic-size inverse()), H_ll^-1 b_l
//     phase 2    fp64 MFMA tiles: the direct blocks C_k = sum_g (L J_pose)^T (L J_pose), two observation indices per 16x16
//                tile, and the lower tiles of the Schur term - sum_g W_g Y_g^T, Y = W H_ll^-1 (tempH, problem.cc:427) over the
//                6K pattern columns: the matrix core's k index runs over 4 landmarks, 3 steps per group (one per
//                coordinate); Y is formed in registers from W and H_ll^-1 on the way into the B operand, never stored:
//                the record of a landmark is 18K + 29 doubles and a 5-frame item holds 82 landmarks, so that the 20k
//                landmarks of the bench window are ONE round of workgroups on the 256 CUs (with W, Y and the b terms
//                in LDS it held 56: 357 items, two rounds)
//     combine    thread per slab element, same slab layout as k_linearize (k_reduce and k_assemble are shared)
//   k_backsub_xyz     delta_l = H_ll^-1 (b_l - W^T dx_pose) (problem.cc:445), trial points, chi2 of the trial state
//
// HBM layout differences to the inverse-depth plan: invd[2][3][Ns] holds the points coordinate-major, dxl[3][Ns],
// lw per item = 9 fields x G: the 6 distinct entries of H_ll, b_l (3).  W is not kept: the back-substitution (k_backsub_xyz,
// or the GN head of the next k_linearize_xyz) forms it again from the state it was linearised at.
#ifndef VIO_KERNELS_XYZ_H
#define VIO_KERNELS_XYZ_H

// landmark record: W 18nb | H 6 | Hinv 9 | b_l 3 | Hinv b_l 3 | GN head: new point 3, delta 3, gain-ratio term 1 (+1: odd stride)
__host__ __device__ inline int xyz_lrec(int nb) { return 18 * nb + 29; }
__host__ __device__ inline int xyz_ntd(int K) { return (K + 1) >> 1; }         // direct tiles: two observation indices each
__host__ __device__ inline int xyz_plane(int G) { return G * 24 + 8; }
__host__ __device__ inline int xyz_tiles(int K) { const int ts = (6 * K + 15) >> 4; return xyz_ntd(K) + ts * (ts + 1) / 2; }
#define XYZ_FRAME_TAB (VIO_NF * 12 + 16)        // A_k (9), d_k (3) per frame, then ric (9), tic (3)
#define XYZ_BP_TAB (VIO_NF * 2 * 6)             // pose part of b: per frame index, per 64-landmark chunk (G <= 128)
// Phase 2 has few, long chains (K = 5: three direct tiles of 41 products and three Schur tiles of 62 on 16 waves: 103 products on
// each of two SIMDs, 41 and 62 on the others).  Where the space the per-observation records leave behind holds them, every Schur
// tile is formed in three parts over a third of the landmarks each (83 / 83 / 83 / 63) and the combine phase adds the parts in order.
__host__ __device__ inline int xyz_split(int G, int K) {
    const int ts = (6 * K + 15) >> 4, nts = ts * (ts + 1) / 2;
    const int aux = G * K * 9, head = 3 * G * K + 12 * K + 12 + 176 + 12 * G;
    const int room = aux > head ? aux : head;
    return ((xyz_ntd(K) + 3 * nts) * 256 + 6 * K * LIN_VS <= room && G >= 64) ? 3 : 1;      // (items of the half-width kernel stay whole: 12 chains on 8 waves wrap)
}
__host__ __device__ inline int xyz_lds_doubles(int G, int K) {
    const int ts_ = (6 * K + 15) >> 4, nts_ = ts_ * (ts_ + 1) / 2;
    int aux = G * K * 9, part = (xyz_ntd(K) + xyz_split(G, K) * nts_) * 256 + 6 * K * LIN_VS;
    int head = 3 * G * K + 12 * K + 12 + 176 + 12 * G;      // GN head: W^T dx per observation, the old camera maps, dx, (H_ll, b_l, point) per landmark
    int shared = aux > part ? aux : part;
    if (head > shared) shared = head;
    shared = (shared + 1) & ~1;
    return XYZ_FRAME_TAB + XYZ_BP_TAB + 3 * (LIN_THREADS / 64) + xyz_ntd(K) * xyz_plane(G) + G * xyz_lrec(K) + shared;
}

// Hmm.block(idx, idx, 3, 3).inverse() the way Eigen evaluates it for a block of a dynamic matrix: PartialPivLU
// (first largest |entry| of the column, the column divided by the pivot, rank-1 update), then P * I through the unit-lower
// and the upper triangular solves, the diagonal applied as a multiplication by its reciprocal.  Row-major.
__device__ __forceinline__ void d_inverse3(const double *A, double *X) {
    double lu[9];
    int piv[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) lu[k] = A[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double big = fabs(lu[3 * k + k]);
#pragma unroll
        for (int i = k + 1; i < 3; ++i) if (fabs(lu[3 * i + k]) > big) { big = fabs(lu[3 * i + k]); best = i; }
        piv[k] = best;
        if (big != 0.0) {
            if (best != k) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    // (selects, not run-time indices: lu stays in registers)
                    const double rk = lu[3 * k + j];
                    const double rb = (best == 1) ? lu[3 + j] : lu[6 + j];
                    lu[3 * k + j] = rb;
                    if (best == 1) lu[3 + j] = rk; else lu[6 + j] = rk;
                }
            }
#pragma unroll
            for (int i = k + 1; i < 3; ++i) lu[3 * i + k] /= lu[3 * k + k];
        }
#pragma unroll
        for (int i = k + 1; i < 3; ++i)
#pragma unroll
            for (int j = k + 1; j < 3; ++j) lu[3 * i + j] -= lu[3 * i + k] * lu[3 * k + j];
    }
    double Xr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (piv[k] != k) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double rk = Xr[3 * k + j];
                const double rb = (piv[k] == 1) ? Xr[3 + j] : Xr[6 + j];
                Xr[3 * k + j] = rb;
                if (piv[k] == 1) Xr[3 + j] = rk; else Xr[6 + j] = rk;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double b = Xr[3 * i + j];
#pragma unroll
            for (int r = i + 1; r < 3; ++r) Xr[3 * r + j] -= b * lu[3 * r + i];
        }
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            const double a = 1.0 / lu[3 * i + i];
            const double b = (Xr[3 * i + j] *= a);
#pragma unroll
            for (int r = 0; r < i; ++r) Xr[3 * r + j] -= b * lu[3 * r + i];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) X[k] = Xr[k];
}

// camera map of frame f at the state `st`: out[0..8] = A = ric^T R_f^T, out[9..11] = d = -ric^T (R_f^T P_f + tic)
__device__ __forceinline__ void d_xyz_frame(const double *st, int f, const double *ric, double *out) {
    double Rf[9], M[9], u[3], w[3];
    d_quat_to_R(st + STATE_POSE + 7 * f + 3, Rf);
    d_m3_mul(Rf, ric, M);                                   // A = (R_f ric)^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) out[3 * i + j] = M[3 * j + i];
    d_m3_tvec(Rf, st + STATE_POSE + 7 * f, u);              // R_f^T P_f
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] += st[STATE_EXT + k];
    d_m3_tvec(ric, u, w);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[9 + k] = -w[k];
}

// One EdgeReprojectionXYZ at the point pw seen through the camera map (A, d) of its frame: residual, jacobians
// (edge_reprojection.cc:147-180), robust weight (Edge::RobustInfo, edge.cc:48-74, information = s^2 I) as the whitening
// L (L^T L = the robustified information) and c = drho * Information * residual.  Shared by the linearisation and by the
// back-substitution, which forms W again from the state it was linearised at instead of reading it back from HBM.
struct XyzObs {
    double Jf0[3], Jf1[3], Jp0[6], Jp1[6];
    double L00, L01, L11, c0, c1, chi;
};
__device__ __forceinline__ void d_xyz_obs(const double *A, const double *ric, const double *tic, const double *pw, double u, double v,
                                          int loss_type, double loss_delta, double s_info, XyzObs &o) {
    const double info = s_info * s_info;
    double pc[3], pim[3];
    d_m3_vec(A, pw, pc);
#pragma unroll
    for (int m = 0; m < 3; ++m) pc[m] += A[9 + m];
    d_m3_vec(ric, pc, pim);                             // pts_imu_i = ric p_c + tic
#pragma unroll
    for (int m = 0; m < 3; ++m) pim[m] += tic[m];
    const double iz = 1.0 / pc[2];
    const double r0 = pc[0] * iz - u, r1 = pc[1] * iz - v;
    const double ra = -pc[0] * (iz * iz), rb = -pc[1] * (iz * iz);        // reduce = [iz 0 ra; 0 iz rb]
    double RR0[3], RR1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o.Jf0[c] = iz * A[c] + ra * A[6 + c];             // jacobian_feature = reduce * ric^T * Ri^T
        o.Jf1[c] = iz * A[3 + c] + rb * A[6 + c];
        RR0[c] = iz * ric[3 * c] + ra * ric[3 * c + 2];   // reduce * ric^T
        RR1[c] = iz * ric[3 * c + 1] + rb * ric[3 * c + 2];
        o.Jp0[c] = -o.Jf0[c]; o.Jp1[c] = -o.Jf1[c];       // reduce * ric^T * -Ri^T
    }
    // reduce * ric^T * hat(pts_imu_i): row * hat(v) = row x v
    o.Jp0[3] = RR0[1] * pim[2] - RR0[2] * pim[1]; o.Jp0[4] = RR0[2] * pim[0] - RR0[0] * pim[2]; o.Jp0[5] = RR0[0] * pim[1] - RR0[1] * pim[0];
    o.Jp1[3] = RR1[1] * pim[2] - RR1[2] * pim[1]; o.Jp1[4] = RR1[2] * pim[0] - RR1[0] * pim[2]; o.Jp1[5] = RR1[0] * pim[1] - RR1[1] * pim[0];

    const double e2 = r0 * (info * r0) + r1 * (info * r1);
    double rho0, rho1, rho2;
    d_loss(loss_type, loss_delta, e2, rho0, rho1, rho2);
    o.chi = (loss_type == 0) ? e2 : rho0;
    double lam2 = rho1;
    if (loss_type != 0 && rho1 + 2 * rho2 * e2 > 0.) lam2 = rho1 + 2 * rho2 * e2;
    const double al = sqrt(rho1), be = sqrt(fmax(lam2, 0.0));
    const double rn2 = r0 * r0 + r1 * r1;
    const double irn2 = rn2 > 0 ? 1.0 / rn2 : 0.0;
    const double gm = (be - al) * irn2;
    o.L00 = s_info * (al + gm * r0 * r0); o.L01 = s_info * (gm * r0 * r1); o.L11 = s_info * (al + gm * r1 * r1);
    o.c0 = rho1 * (info * r0); o.c1 = rho1 * (info * r1);               // drho * Information * residual
}

// the observation's term of W^T dx_pose, W = (L J_pose)^T (L J_feature) the Hpm block of (pose, landmark), without forming
// the jacobians:  W^T dx = J_feature^T (L L (J_pose dx)),  J_pose dx = reduce (-A dp + ric^T (p_imu x dtheta)) — the
// directional derivative of the projection along the pose step (J_pose = [-reduce A | reduce ric^T hat(p_imu)],
// edge_reprojection.cc:170-176) — and J_feature = reduce A.  A third of the arithmetic of d_xyz_obs.
__device__ __forceinline__ void d_xyz_obs_tdx(const double *A, const double *ric, const double *tic, const double *pw, double u, double v,
                                              int loss_type, double loss_delta, double s_info, const double *d, double *t) {
    const double info = s_info * s_info;
    double pc[3], pim[3];
    d_m3_vec(A, pw, pc);
#pragma unroll
    for (int m = 0; m < 3; ++m) pc[m] += A[9 + m];
    d_m3_vec(ric, pc, pim);
#pragma unroll
    for (int m = 0; m < 3; ++m) pim[m] += tic[m];
    const double iz = 1.0 / pc[2];
    const double r0 = pc[0] * iz - u, r1 = pc[1] * iz - v;
    const double ra = -pc[0] * (iz * iz), rb = -pc[1] * (iz * iz);
    const double cr[3] = {pim[1] * d[5] - pim[2] * d[4], pim[2] * d[3] - pim[0] * d[5], pim[0] * d[4] - pim[1] * d[3]};     // p_imu x dtheta
    double dp[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
        dp[m] = (ric[m] * cr[0] + ric[3 + m] * cr[1] + ric[6 + m] * cr[2]) - (A[3 * m] * d[0] + A[3 * m + 1] * d[1] + A[3 * m + 2] * d[2]);
    const double s0 = iz * dp[0] + ra * dp[2], s1 = iz * dp[1] + rb * dp[2];
    const double e2 = r0 * (info * r0) + r1 * (info * r1);
    double rho0, rho1, rho2;
    d_loss(loss_type, loss_delta, e2, rho0, rho1, rho2);
    double lam2 = rho1;
    if (loss_type != 0 && rho1 + 2 * rho2 * e2 > 0.) lam2 = rho1 + 2 * rho2 * e2;
    const double al = sqrt(rho1), be = sqrt(fmax(lam2, 0.0));
    const double rn2 = r0 * r0 + r1 * r1;
    const double irn2 = rn2 > 0 ? 1.0 / rn2 : 0.0;
    const double gm = (be - al) * irn2;
    const double L00 = s_info * (al + gm * r0 * r0), L01 = s_info * (gm * r0 * r1), L11 = s_info * (al + gm * r1 * r1);
    const double y0 = L00 * s0 + L01 * s1, y1 = L01 * s0 + L11 * s1;
    const double z0 = L00 * y0 + L01 * y1, z1 = L01 * y0 + L11 * y1;
    const double w0 = iz * z0, w1 = iz * z1, w2 = ra * z0 + rb * z1;                  // reduce^T z
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = A[c] * w0 + A[3 + c] * w1 + A[6 + c] * w2;
}

// delta_l = H_ll^-1 (b_l - W^T dx_pose) (problem.cc:445) and the landmark's term of the gain-ratio denominator
__device__ __forceinline__ double d_xyz_landmark_step(const double *h, const double *bl, const double *t, double lambda, double *dl) {
    const double Hm[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
    double Hi[9];
    d_inverse3(Hm, Hi);
    const double v0 = bl[0] - t[0], v1 = bl[1] - t[1], v2 = bl[2] - t[2];
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dl[i] = Hi[3 * i] * v0 + Hi[3 * i + 1] * v1 + Hi[3 * i + 2] * v2;
        scale += dl[i] * (lambda * dl[i] + bl[i]);
    }
    return scale;
}


template <int NT> __device__ __forceinline__ void d_linearize_xyz_body(const DeviceTables &T) {
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (d_gated_off(T.lm, T.lm_gate)) return;
    // GN mode (gn_flags bit 1), as in k_linearize: the previous step's b_prior' rows and landmark back-substitution come first
    const bool owe = d_step_owed(T, 2), owe_prior = owe && T.has_prior;
    if (b >= T.n_items) {
        if (owe_prior && (tid >> 6) == NT / 64 - 1) { const int c = d_cur(T); d_bprior_rows(T, c ^ 1, c, b, T.n_step_blocks, tid & 63); }
        d_imu_item<NT>(T, b - T.n_items, dyn_smem);
        return;
    }
    __shared__ ItemDesc sIt;
    const int cur = d_cur(T);
    const int64_t lw_r = d_set_r(T) * T.lw_set, lw_w = d_set_w(T) * T.lw_set;
    if (tid < (int)(sizeof(ItemDesc) / 4)) ((int32_t *)&sIt)[tid] = ((const int32_t *)(T.items + b))[tid];
    __syncthreads();
    const ItemDesc &it = sIt;
    const int G = it.G, K = it.K, nb = it.nb;              // nb == K: one pattern block per observing frame
    const int LREC = xyz_lrec(nb), PLANE = xyz_plane(G), NTD = xyz_ntd(K);
    const int offW = 0, offH = 18 * nb, offHI = offH + 6, offBL = offH + 15, offV = offH + 18, offPW = offH + 21, offDL = offH + 24, offSC = offH + 27;

    double *sFr = dyn_smem;                                // [11][12] camera maps of the item's frames (indexed by k), then ric, tic
    double *sCam = sFr + VIO_NF * 12;
    double *sBp = sFr + XYZ_FRAME_TAB;                     // [K][2][6] wave partials of the pose part of b
    double *sRed = sBp + XYZ_BP_TAB;                       // 3 * waves
    double *sRows = sRed + 3 * (LIN_THREADS / 64);         // NTD planes of G x 24
    double *sL = sRows + NTD * PLANE;                      // G * LREC
    double *sAux = sL + G * LREC;                          // G*K*9, dead after phase 1.5 ...
    double *sTile = sAux;                                  // ... then the tiles and the vector partials

    const double *st = T.state + cur * STATE_STRIDE;
    const double *xyz = T.invd + (size_t)cur * 3 * T.Ns + it.lm_base;
    const double *pts = T.pts_j + 2 * (size_t)it.obs_base;
    const double s_info = T.sqrt_info;
    if (owe_prior && (tid >> 6) == NT / 64 - 1) d_bprior_rows(T, cur ^ 1, cur, b, T.n_step_blocks, tid & 63);
    STAMP(T, 0);
    // phase 0: camera maps (their loads leave together with the head's: one global round trip for both)
    constexpr int kCamWave = (NT / 64 - 2) * 64;             // the last wave but one (wave 14 of 16): no head loads of its own in front of these
    if (tid >= kCamWave && tid < kCamWave + K) {
        const int k = tid - kCamWave;
        double ric[9], o[12];
        d_quat_to_R(st + STATE_EXT + 3, ric);
        d_xyz_frame(st, it.cam_block[k] - 1, ric, o);            // K may be 11: the frame comes from cam_block, not target[10]
#pragma unroll
        for (int q = 0; q < 12; ++q) sFr[12 * k + q] = o[q];
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) sCam[q] = ric[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) sCam[9 + q] = st[STATE_EXT + q];
        }
    }
    if (tid >= 128 && tid < 132) sCam[12 + (tid - 128)] = 0.0;        // zeros for the padding lanes of phase 2 to stream
    if (tid >= 256 && tid < 256 + XYZ_BP_TAB) sBp[tid - 256] = 0.0;
    // an odd K leaves the second half of the last plane unused: the direct products read it, so it holds zeros
    if (K & 1) for (int e = tid; e < G * 12; e += NT) sRows[(NTD - 1) * PLANE + (e / 12) * 24 + 12 + e % 12] = 0.0;
    // ---------------- GN head: delta_l of the PREVIOUS step (problem.cc:445) ----------------
    // W is not kept in HBM: every observation forms its block again at the state it was linearised at (the other copy of the
    // state and of the points) and multiplies it by the pose step; H_ll and b_l (9 values per landmark) come from lw.
    if (owe) {
        double *sT = sAux;                                 // 3 per observation
        double *sFrO = sAux + 3 * G * K, *sCamO = sFrO + 12 * K, *sDx = sCamO + 12;      // fits: xyz_lds_doubles
        const double *sto = T.state + (cur ^ 1) * STATE_STRIDE;
        const double *xyzo = T.invd + (size_t)(cur ^ 1) * 3 * T.Ns + it.lm_base;
        const double *lw = T.lw + lw_r + it.lw_base;
        double *sHb = sDx + 176;                           // 12 per landmark: H_ll (6), b_l (3), the old point
        for (int e = tid; e < 12 * G; e += NT) {
            const int q = e / G, g = e - q * G;
            sHb[12 * g + q] = q < 9 ? lw[e] : xyzo[(size_t)(q - 9) * T.Ns + g];
        }
        const double lambda_lm = T.lm->lambda;
        if (tid < 176) sDx[tid] = T.dx[tid];
        if (tid >= 192 && tid < 192 + K) {
            const int k = tid - 192;
            double ric[9], o[12];
            d_quat_to_R(sto + STATE_EXT + 3, ric);
            d_xyz_frame(sto, it.cam_block[k] - 1, ric, o);
#pragma unroll
            for (int q = 0; q < 12; ++q) sFrO[12 * k + q] = o[q];
            if (k == 0) {
#pragma unroll
                for (int q = 0; q < 9; ++q) sCamO[q] = ric[q];
#pragma unroll
                for (int q = 0; q < 3; ++q) sCamO[9 + q] = sto[STATE_EXT + q];
            }
        }
        __syncthreads();
        for (int o = tid; o < G * K; o += NT) {
            const int k = o / G, g = o - k * G;
            const double pw[3] = {sHb[12 * g + 9], sHb[12 * g + 10], sHb[12 * g + 11]};
            double t[3];
            d_xyz_obs_tdx(sFrO + 12 * k, sCamO, sCamO + 9, pw, pts[2 * o], pts[2 * o + 1], T.loss_type, T.loss_delta, s_info,
                          sDx + 6 + 15 * (it.cam_block[k] - 1), t);
#pragma unroll
            for (int c = 0; c < 3; ++c) sT[3 * o + c] = t[c];
        }
        __syncthreads();
        if (tid < G) {
            double t[3] = {0.0, 0.0, 0.0}, dl[3];
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) t[c] += sT[3 * (k * G + tid) + c];
            double hb[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) hb[q] = sHb[12 * tid + q];
            const double sc = d_xyz_landmark_step(hb, hb + 6, t, lambda_lm, dl);
            double *Lg = sL + (size_t)tid * LREC;
#pragma unroll
            for (int c = 0; c < 3; ++c) { Lg[offPW + c] = hb[9 + c] + dl[c]; Lg[offDL + c] = dl[c]; }
            Lg[offSC] = sc;
        }
    }
    __syncthreads();
    const double *ric = sCam, *tic = sCam + 9;

    STAMP(T, 1);
    // ---------------- phase 1: thread per observation ----------------
    // thread -> (k, g) with the landmarks of a frame index padded to whole waves: every wave works on one k, so that the
    // pose part of b, - sum_g drho J_pose^T Info r, is a sum inside the wave (it had 6 doubles per observation in LDS)
    double chi_acc = 0.0;
    const int Gp = ((G + 63) >> 6) << 6;
    for (int o2 = tid; o2 < K * Gp; o2 += NT) {
        const int k = o2 / Gp, g = o2 - k * Gp;
        const bool act = g < G;
        double bpv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (act) {
            const int o = k * G + g;
            double *Lg = sL + (size_t)g * LREC;
            double pw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) pw[c] = owe ? Lg[offPW + c] : xyz[(size_t)c * T.Ns + g];
            XyzObs ob;
            d_xyz_obs(sFr + 12 * k, ric, tic, pw, pts[2 * o], pts[2 * o + 1], T.loss_type, T.loss_delta, s_info, ob);
            chi_acc += ob.chi;
            const double *Jf0 = ob.Jf0, *Jf1 = ob.Jf1, *Jp0 = ob.Jp0, *Jp1 = ob.Jp1;
            const double L00 = ob.L00, L01 = ob.L01, L11 = ob.L11, c0 = ob.c0, c1 = ob.c1;
            double lf0[3], lf1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { lf0[c] = L00 * Jf0[c] + L01 * Jf1[c]; lf1[c] = L01 * Jf0[c] + L11 * Jf1[c]; }
            double *rec = sRows + (k >> 1) * PLANE + g * 24 + (k & 1) * 12;
            double *pk = sAux + (size_t)o * 9;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double lp0 = L00 * Jp0[i] + L01 * Jp1[i], lp1 = L01 * Jp0[i] + L11 * Jp1[i];
                rec[i] = lp0; rec[6 + i] = lp1;
#pragma unroll
                for (int c = 0; c < 3; ++c) Lg[offW + (6 * k + i) * 3 + c] = lp0 * lf0[c] + lp1 * lf1[c];     // Hpm block of (pose k, landmark)
                bpv[i] = Jp0[i] * c0 + Jp1[i] * c1;
            }
            pk[0] = lf0[0] * lf0[0] + lf1[0] * lf1[0]; pk[1] = lf0[0] * lf0[1] + lf1[0] * lf1[1]; pk[2] = lf0[0] * lf0[2] + lf1[0] * lf1[2];
            pk[3] = lf0[1] * lf0[1] + lf1[1] * lf1[1]; pk[4] = lf0[1] * lf0[2] + lf1[1] * lf1[2]; pk[5] = lf0[2] * lf0[2] + lf1[2] * lf1[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) pk[6 + c] = Jf0[c] * c0 + Jf1[c] * c1;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double wsum = d_wave_sum_to_lane63(bpv[i]);
            if ((tid & 63) == 63) sBp[(2 * k + (g >> 6)) * 6 + i] = wsum;
        }
    }
    __syncthreads();

    STAMP(T, 2);
    // ---------------- phase 1.5: thread per landmark: H_ll, b_l, H_ll^-1, H_ll^-1 b_l ----------------
    double maxh = 0.0;
    if (tid < G) {
        const int g = tid;
        double h[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
        for (int k = 0; k < K; ++k) {
            const double *pk = sAux + (size_t)(k * G + g) * 9;
#pragma unroll
            for (int q = 0; q < 6; ++q) h[q] += pk[q];
#pragma unroll
            for (int c = 0; c < 3; ++c) bl[c] += pk[6 + c];
        }
        const double Hm[9] = {h[0], h[1], h[2], h[1], h[3], h[4], h[2], h[4], h[5]};
        double Hi[9];
        d_inverse3(Hm, Hi);
        double *Lg = sL + (size_t)g * LREC;
#pragma unroll
        for (int q = 0; q < 6; ++q) Lg[offH + q] = h[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) Lg[offHI + q] = Hi[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) Lg[offBL + c] = -bl[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) Lg[offV + c] = Hi[3 * c] * -bl[0] + Hi[3 * c + 1] * -bl[1] + Hi[3 * c + 2] * -bl[2];      // H_ll^-1 b_l
        maxh = fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5])));
    }
    {
        // (GN) the previous step's gain-ratio partial: thread g holds landmark g's term, summed as k_backsub_xyz sums it
        const double sc = (owe && tid < G) ? sL[(size_t)tid * LREC + offSC] : 0.0;
        const double ws = d_wave_sum_to_lane63(chi_acc), wsc = d_wave_sum_to_lane63(sc), wm = d_wave_max_to_lane63(maxh);
        if ((tid & 63) == 63) { sRed[tid >> 6] = ws; sRed[NT / 64 + (tid >> 6)] = wsc; sRed[2 * (NT / 64) + (tid >> 6)] = wm; }
    }
    __syncthreads();
    STAMP(T, 3);
    // ---------------- phase 2: tiles on the matrix cores ----------------
    const int D = 6 * nb;
    const int TS = (D + 15) >> 4, nts = TS * (TS + 1) / 2;
    // parts a Schur tile is formed in (the half-width instances keep whole tiles at compile time: their 8 waves have the chains they
    // can carry, and the plain code is 2 % faster for them; the LDS size the host computed covers either)
    const int SPL = NT == LIN_THREADS ? xyz_split(G, K) : 1;
    double *sVec = sTile + (size_t)(NTD + SPL * nts) * 256;
    {
        const int wave = tid >> 6, lane = tid & 63, cl = lane & 15, rg = lane >> 4;
        const int nwork = NTD + SPL * nts;
        for (int wk = wave; wk < nwork; wk += NT / 64) {
            ps_v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (wk < NTD) {
                // C = V^T V, V = the 2G whitened pose rows of observation indices 2 wk and 2 wk + 1 side by side (12 columns);
                // one MFMA step takes two landmarks (lane row group rg: landmark rg >> 1, row rg & 1)
                // (a padding lane (column >= 12) streams its clamped neighbour: the tile's columns 12..15 are never read; one stride for
                // all lanes, the pointer only steps; chunks of 4 steps with the next chunk's loads pinned in front of this chunk's products)
                const int clc = min(cl, 11);
                const int off = (clc < 6 ? 0 : 12) + (rg & 1) * 6 + (clc < 6 ? clc : clc - 6);
                const double *pv = sRows + wk * PLANE + off + (rg >> 1) * 24;
                const int sp = 2 * 24;                                  // two landmarks per step
                const int chunks = (G >> 1) >> 2;                       // chunks of 4 steps whose landmarks all exist
                double va[4], xa[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) va[u] = pv[u * sp];          // (G >= 1: in range even if unused)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                int ch = 0;
                for (; ch + 2 <= chunks; ch += 2) {
                    pv += 4 * sp;
#pragma unroll
                    for (int u = 0; u < 4; ++u) xa[u] = pv[u * sp];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], va[u], acc, 0, 0, 0);
                    pv += ch + 2 < chunks ? 4 * sp : 0;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) va[u] = pv[u * sp];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u], xa[u], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ch < chunks) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u], va[u], acc, 0, 0, 0);
                }
                // the remaining steps (fewer than 4 full ones, plus the half step of an odd G), masked
                const double *pl0 = sRows + wk * PLANE + off;
                for (int st2 = 4 * chunks; 2 * st2 < G; ++st2) {
                    const int g = 2 * st2 + (rg >> 1);
                    const double vv = pl0[min(g, G - 1) * 24];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(g < G ? vv : 0.0, vv, acc, 0, 0, 0);
                }
            } else {
                const int ts = (wk - NTD) / SPL, part = (wk - NTD) - ts * SPL;
                int ta = 0;
                while ((ta + 1) * (ta + 2) / 2 <= ts) ++ta;
                const int tb = ts - ta * (ta + 1) / 2;
                const int a = 16 * ta + cl, bq = 16 * tb + cl;
                // one group of 4 landmarks (k index rg) per step, three matrix instructions (coordinate c) per group:
                // A = W[a][c], B = -Y[bq][c] with Y = W H_ll^-1 formed here from the three W of the row and H_ll^-1.
                // A row past D streams its clamped neighbour (what lands in the tile's unused rows / columns is never read): one
                // stride for all lanes, pointers that only step.  Two groups per trip, the operand registers taking turns, the loads
                // of the next group pinned in front of this group's products (left alone the compiler sinks every load to its use and
                // each group waits for an LDS latency and three 16-cycle address multiplications).
                const int ac = min(a, D - 1), bc = min(bq, D - 1);
                const double *pa = sL + (size_t)rg * LREC + offW + ac * 3, *pb = sL + (size_t)rg * LREC + offW + bc * 3;
                const double *ph = sL + (size_t)rg * LREC + offHI;
                const int sg = 4 * LREC;
                const int ngr = G >> 2;                                // groups whose four landmarks all exist
                const int glo = part * ngr / SPL, nfull = (part + 1) * ngr / SPL - glo;      // this part's groups
                pa += (size_t)glo * sg; pb += (size_t)glo * sg; ph += (size_t)glo * sg;
                double wa[3], wb[3], hi[9], xa[3], xb[3], xh[9];
#pragma unroll
                for (int c = 0; c < 3; ++c) { wa[c] = pa[c]; wb[c] = pb[c]; }
#pragma unroll
                for (int q = 0; q < 9; ++q) hi[q] = ph[q];
                __builtin_amdgcn_s_waitcnt(0xC07F);
                int st4 = 0;
                for (; st4 + 2 <= nfull; st4 += 2) {
                    pa += sg; pb += sg; ph += sg;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { xa[c] = pa[c]; xb[c] = pb[c]; }
#pragma unroll
                    for (int q = 0; q < 9; ++q) xh[q] = ph[q];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double yb = wb[0] * hi[c] + wb[1] * hi[3 + c] + wb[2] * hi[6 + c];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[c], -yb, acc, 0, 0, 0);
                    }
                    const int nx = st4 + 2 < nfull ? sg : 0;           // (past the last full group: the same group again, unused)
                    pa += nx; pb += nx; ph += nx;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { wa[c] = pa[c]; wb[c] = pb[c]; }
#pragma unroll
                    for (int q = 0; q < 9; ++q) hi[q] = ph[q];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double yb = xb[0] * xh[c] + xb[1] * xh[3 + c] + xb[2] * xh[6 + c];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[c], -yb, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (st4 < nfull) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double yb = wb[0] * hi[c] + wb[1] * hi[3 + c] + wb[2] * hi[6 + c];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[c], -yb, acc, 0, 0, 0);
                    }
                }
                if ((G & 3) && part == SPL - 1) {                       // the last, partial group: masked
                    const int g = 4 * ngr + rg, gc = min(g, G - 1);
                    const double m = g < G ? 1.0 : 0.0;
                    const double *La = sL + (size_t)gc * LREC + offW + ac * 3, *Lb = sL + (size_t)gc * LREC + offW + bc * 3;
                    const double *Lh = sL + (size_t)gc * LREC + offHI;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const double yb = Lb[0] * Lh[c] + Lb[1] * Lh[3 + c] + Lb[2] * Lh[6 + c];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(La[c] * m, -yb, acc, 0, 0, 0);
                    }
                }
            }
            double *tl = sTile + (size_t)wk * 256 + rg * 16 + cl;   // C/D image: row rg + 4v, column cl
#pragma unroll
            for (int v = 0; v < 4; ++v) tl[64 * v] = acc[v];
        }
        // Schur correction of b = sum_g Y_g b_l,g = sum_g W_g (H_ll^-1 b_l)_g   (fixed order)
        for (int e = tid; e < D * LIN_VS; e += NT) {
            const int part = e % LIN_VS, a = e / LIN_VS;
            double sum = 0.0;
            for (int g = part; g < G; g += LIN_VS) {
                const double *Lg = sL + (size_t)g * LREC;
                sum += Lg[offW + 3 * a] * Lg[offV] + Lg[offW + 3 * a + 1] * Lg[offV + 1] + Lg[offW + 3 * a + 2] * Lg[offV + 2];
            }
            sVec[e] = sum;
        }
    }
    __syncthreads();

    STAMP(T, 4);
    // ---------------- combine: thread per slab element ----------------
    {
        double *out = T.slab + it.out_base;
        const int n_out = it.n_rows * 6;
        const int n_pair = (nb * (nb + 1) / 2) * 36;
        double chi = 0.0, mh = 0.0;
        if (tid == 0) {
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { chi += sRed[w]; mh = fmax(mh, sRed[2 * (NT / 64) + w]); }
        }
        // entry (i, j) of the direct block of pattern block p
        auto cdir = [&](int p, int i, int j) -> double { return sTile[(size_t)(p >> 1) * 256 + ((p & 1) * 6 + i) * 16 + (p & 1) * 6 + j]; };
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
                const double *pt = sTile + (size_t)(NTD + SPL * (ta * (ta + 1) / 2 + tb)) * 256 + (hi & 15) * 16 + (lo & 15);
                v = pt[0];
                for (int sp_ = 1; sp_ < SPL; ++sp_) v += pt[sp_ * 256];      // the tile's parts, in order
                if (p == q) v += cdir(p, i, j);
            } else {
                const int ve = e - n_pair, which = ve / D, a = ve - which * D;
                if (which == 0) {           // direct b: the wave partials of phase 1 (a = 6 k + i; second chunk zero for G <= 64)
                    const int p = a / 6, i = a - 6 * p;
                    v = -(sBp[(2 * p) * 6 + i] + sBp[(2 * p + 1) * 6 + i]);
                } else if (which == 1) {
                    const double *pv = sVec + (size_t)a * LIN_VS;
#pragma unroll
                    for (int q = 0; q < LIN_VS; ++q) v += pv[q];
                } else {
                    const int p = a / 6, i = a - 6 * p;
                    v = cdir(p, i, i);
                }
            }
            out[e] = v;
        }
        if (tid == 0) { out[n_out] = chi; out[n_out + 1] = mh; }
        if (owe && tid == 64) {
            double sc = 0.0;
#pragma unroll
            for (int w = 0; w < BS_THREADS / 64; ++w) sc += sRed[NT / 64 + w];
            T.step_part[2 * b + STEP_SCALE] = sc; T.step_part[2 * b + STEP_CHI] = 0.0;
        }
        if (owe && tid < G) {               // the landmark update of the head, out to HBM now
            const size_t li = (size_t)it.lm_base + tid;
            const double *Lg = sL + (size_t)tid * LREC;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T.dxl[(size_t)c * T.Ns + li] = Lg[offDL + c];
                T.invd[((size_t)cur * 3 + c) * T.Ns + li] = Lg[offPW + c];
            }
        }
        // H_ll (6 distinct entries) and b_l of the item's landmarks for the back-substitution
        double *lw = T.lw + lw_w + it.lw_base;
        for (int e = tid; e < 9 * G; e += NT) {
            const int r = e / G, g = e - r * G;
            const double *Lg = sL + (size_t)g * LREC;
            lw[e] = r < 6 ? Lg[offH + r] : Lg[offBL + (r - 6)];
        }
    }
    STAMP(T, 5);
    STAMP_FLUSH(T);
}

__global__ __launch_bounds__(LIN_THREADS) void k_linearize_xyz(DeviceTables T) { d_linearize_xyz_body<LIN_THREADS>(T); }
__global__ __launch_bounds__(LIN_THREADS) void k_linearize_xyz_b(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;       // the grid is the widest window's
    d_linearize_xyz_body<LIN_THREADS>(T);
}
// half width, two workgroups to a CU: plans of the throughput policy (see k_linearize_h)
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_xyz_h(DeviceTables T) { d_linearize_xyz_body<LIN_THREADS_H>(T); }
__global__ __launch_bounds__(LIN_THREADS_H, 4) void k_linearize_xyz_hb(BatchArgs a) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;
    d_linearize_xyz_body<LIN_THREADS_H>(T);
}

// ---------------------------------------------------------------------------------------------------------
// k_backsub_xyz: one workgroup per item, one thread per landmark.  mode as k_backsub.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_backsub_xyz_body(const DeviceTables &T, int mode) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const LmState *lm = T.lm;
    if (d_gated_off(lm, T.lm_gate)) return;
    const int cur = d_cur(T);
    const int which = (mode == 1) ? cur : (cur ^ 1);
    if ((T.gn_flags & 8) && T.has_prior && (lane >> 6) == 1) d_bprior_rows(T, cur, cur ^ 1, b, T.n_step_blocks, lane & 63);
    if (b >= T.n_items) { d_backsub_imu_block(T, mode, which, b, lane); return; }
    __shared__ double sFr[VIO_NF * 12];       // camera maps at the state whose chi2 is wanted
    __shared__ double sFrO[VIO_NF * 12 + 12]; // mode 0: the maps (and ric, tic) at the state the step was linearised at
    __shared__ double sDxp[176];
    __shared__ ItemDesc sIt;
    if (mode == 0) { sDxp[lane] = T.dx[lane]; if (lane + BS_THREADS < 176) sDxp[lane + BS_THREADS] = T.dx[lane + BS_THREADS]; }
    if (lane < (int)(sizeof(ItemDesc) / 4)) ((int32_t *)&sIt)[lane] = ((const int32_t *)(T.items + b))[lane];
    __syncthreads();
    const ItemDesc &it = sIt;
    const int G = it.G, K = it.K;
    const double *st = T.state + which * STATE_STRIDE;
    if (lane < K) {
        double ric[9], o[12];
        d_quat_to_R(st + STATE_EXT + 3, ric);
        d_xyz_frame(st, it.cam_block[lane] - 1, ric, o);
#pragma unroll
        for (int k = 0; k < 12; ++k) sFr[12 * lane + k] = o[k];
    } else if (mode == 0 && lane >= 64 && lane < 64 + K) {
        const int k = lane - 64;
        const double *sto = T.state + cur * STATE_STRIDE;
        double ric[9], o[12];
        d_quat_to_R(sto + STATE_EXT + 3, ric);
        d_xyz_frame(sto, it.cam_block[k] - 1, ric, o);
#pragma unroll
        for (int q = 0; q < 12; ++q) sFrO[12 * k + q] = o[q];
        if (k == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) sFrO[VIO_NF * 12 + q] = ric[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) sFrO[VIO_NF * 12 + 9 + q] = sto[STATE_EXT + q];
        }
    }
    __syncthreads();
    double chi = 0.0, scale = 0.0;
    const double s_info = T.sqrt_info, info = s_info * s_info;
    if (lane < G) {
        const int g = lane;
        const size_t li = (size_t)it.lm_base + g;
        const size_t Ns = (size_t)T.Ns;
        double pw[3] = {T.invd[(size_t)cur * 3 * Ns + li], T.invd[(size_t)cur * 3 * Ns + Ns + li], T.invd[(size_t)cur * 3 * Ns + 2 * Ns + li]};
        if (mode == 0) {
            // W^T dx_pose with W formed again from the linearisation state (k_linearize_xyz keeps H_ll and b_l only);
            // summed per observation, then over the observations: the association of the GN head of k_linearize_xyz
            const double *lw = T.lw + it.lw_base;
            double hb[9], t[3] = {0.0, 0.0, 0.0}, dl[3];
#pragma unroll
            for (int q = 0; q < 9; ++q) hb[q] = lw[(size_t)q * G + g];
            for (int k = 0; k < K; ++k) {
                const size_t o = (size_t)it.obs_base + (size_t)k * G + g;
                double tk[3];
                d_xyz_obs_tdx(sFrO + 12 * k, sFrO + VIO_NF * 12, sFrO + VIO_NF * 12 + 9, pw, T.pts_j[2 * o], T.pts_j[2 * o + 1], T.loss_type, T.loss_delta, s_info,
                              sDxp + 6 + 15 * (it.cam_block[k] - 1), tk);
#pragma unroll
                for (int c = 0; c < 3; ++c) t[c] += tk[c];
            }
            scale = d_xyz_landmark_step(hb, hb + 6, t, lm->lambda, dl);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                T.dxl[(size_t)i * Ns + li] = dl[i];
                pw[i] += dl[i];
                T.invd[(size_t)(cur ^ 1) * 3 * Ns + (size_t)i * Ns + li] = pw[i];
            }
        }
        for (int k = 0; k < K; ++k) {
            const double *A = sFr + 12 * k;
            const size_t o = (size_t)it.obs_base + (size_t)k * G + g;
            double pc[3];
            d_m3_vec(A, pw, pc);
#pragma unroll
            for (int m = 0; m < 3; ++m) pc[m] += A[9 + m];
            const double iz = 1.0 / pc[2];
            const double r0 = pc[0] * iz - T.pts_j[2 * o], r1 = pc[1] * iz - T.pts_j[2 * o + 1];
            const double e2 = r0 * (info * r0) + r1 * (info * r1);
            double rho0, rho1, rho2;
            d_loss(T.loss_type, T.loss_delta, e2, rho0, rho1, rho2);
            chi += (T.loss_type == 0) ? e2 : rho0;
        }
    }
    __shared__ double sSum[2 * (BS_THREADS / 64)];
    chi = d_wave_sum_to_lane63(chi);
    scale = d_wave_sum_to_lane63(scale);
    if ((lane & 63) == 63) { sSum[2 * (lane >> 6)] = chi; sSum[2 * (lane >> 6) + 1] = scale; }
    __syncthreads();
    if (lane == 0) {
        double c = 0.0, sc = 0.0;
#pragma unroll
        for (int w = 0; w < BS_THREADS / 64; ++w) { c += sSum[2 * w]; sc += sSum[2 * w + 1]; }
        double *part = (mode == 1) ? T.chi_part : T.step_part;
        part[2 * b + STEP_CHI] = c; part[2 * b + STEP_SCALE] = sc;
    }
}

__global__ __launch_bounds__(BS_THREADS) void k_backsub_xyz(DeviceTables T, int mode) { d_backsub_xyz_body(T, mode); }
__global__ __launch_bounds__(BS_THREADS) void k_backsub_xyz_b(BatchArgs a, int mode) {
    const DeviceTables T = d_batch_tables(a);
    if ((int)blockIdx.x >= T.n_items + T.n_imu_items) return;
    d_backsub_xyz_body(T, mode);
}

#endif
