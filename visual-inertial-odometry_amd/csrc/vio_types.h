// vio_types.h — device-side layouts shared by the HIP kernels and the C-ABI host code.
//
// Data layout in HBM (all fp64 unless noted):
//   state[2][STATE_STRIDE]   ext(7) | pose(11x7) | speed-bias(11x9); two copies: "current" and "trial"
//   invd[2][Ns]              inverse depths in PATTERN-SORTED landmark order, current / trial
//                            (XYZ landmarks: invd[2][3][Ns], the world points coordinate-major; see vio_kernels_xyz.h)
//   pts_i[Ns][2]             host observation of each landmark
//   pts_j[M][2]              target observations, item-major, inside an item k-major: obs_base + k*G + g
//   items[n_items]           ItemDesc: <= G_MAX landmarks sharing (host, targets) — the unit of work of a workgroup
//   pairtab[2][121][PAIR_STRIDE]   per ordered frame pair (h,t): composed rotations of the reprojection chain
//   slab[...]                per-item compact partial sums (deterministic two-pass reduction, no float atomics)
//   lw[...]                  per-landmark Schur row w (Hpl), 1/h_ll, b_l kept for back-substitution
//   vis[VIS_COUNT]           reduced visual system in the 72-dim camera space (the multi-GPU exchange buffer)
//   imu_out[10][IMU_OUT]     J^T*Info*J (30x30), J^T*Info*r (30), r^T*Info*r per IMU edge
//   Hs[171x171], bs[171]     H_pp_schur_ (no lambda) and b_pp_schur_ of Problem::SolveLinearSystem
//   lm                       LmState: lambda, chi2, ni, which state copy is current, counters
#ifndef VIO_TYPES_H
#define VIO_TYPES_H

#include <stdint.h>

#ifndef __HIPCC__       /* (the host-only translation units — vio_plan.cpp, the tests' drivers — are plain C++) */
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif

#define VIO_NF 11
#define VIO_PD 171
#define VIO_PRD 156
#define VIO_CD 72
#define VIO_NCB 12               // camera blocks: ext + 11 poses
#define VIO_NPAIR 78             // 12*13/2 block pairs (P <= Q)
#define VIO_MAXK 10              // a landmark hosted in frame h has at most 10 other frames
#define VIO_MAXNB 12

#define STATE_EXT 0
#define STATE_POSE 7
#define STATE_SB (7 + 77)
#define STATE_STRIDE 184

#define PAIR_A 0                 // ric^T * Rt^T
#define PAIR_B 9                 // A * Rh
#define PAIR_C 18                // B * ric
#define PAIR_D 27                // ric^T * (Rt^T * (Rh*tic + Ph - Pt) - tic)
#define PAIR_EL 30               // ric^T * (Rt^T*Rh - I)
#define PAIR_STRIDE 40
#define CAMTAB_RIC 0             // stored after the 121 pair entries: ric(9), tic(3)
#define CAMTAB_TIC 9
#define PAIRTAB_STRIDE (121 * PAIR_STRIDE + 16)

#define VIS_H 0                              // the 78 blocks (P <= Q) of the symmetric 72x72, 36 doubles each (row-major 6x6), in k_reduce's
                                             // block order: pair (P, Q) at VIS_PAIR(P, Q) * 36 — half of what the shards exchange
#define VIS_PAIR(P, Q) ((P) * VIO_NCB - (P) * ((P) - 1) / 2 + ((Q) - (P)))
#define VIS_BRED (VIO_NPAIR * 36)            // reduced b (after landmark Schur), 72
#define VIS_BDIR (VIS_BRED + VIO_CD)         // direct b (pose part of b_), 72
#define VIS_DIAG (VIS_BDIR + VIO_CD)         // direct diagonal of Hpp (visual part), 72
#define VIS_CHI (VIS_DIAG + VIO_CD)          // sum of RobustChi2 over reprojection edges
#define VIS_STEP (VIS_CHI + 1)               // 2 slots; [1]: landmark part of the previous GN step's gain-ratio denominator (k_reduce)
#define VIS_MAXH (VIS_STEP + 2)              // max |h_ll| — first slot that is NOT summed across shards
#define VIS_COUNT (VIS_MAXH + 1 + 4)         // padded to a multiple of 8
#define VIS_SEND (VIS_MAXH + 1)              // what a shard sends: the slots summed in rank order, then max |h_ll| (max over the ranks)

#define IMU_T 0
#define IMU_G 900
#define IMU_CHI 930
#define IMU_OUT 936

#define STEP_CHI 0               // per-workgroup partials of the trial step: chi2 and gain-ratio scale
#define STEP_SCALE 1

struct ItemDesc {
    int32_t lm_base;             // first landmark (sorted order)
    int32_t G;                   // landmarks in this item
    int32_t K;                   // observations per landmark
    int32_t nb;                  // pattern blocks: [ext] + frames, ascending camera-block id
    int32_t host;                // host frame
    int32_t host_slot;           // pattern-local index of the host block
    int32_t use_ext;             // 1: pattern-local block 0 is the extrinsic
    int32_t obs_base;            // first observation (item-major storage)
    int32_t out_base;            // offset into slab (doubles)
    int32_t lw_base;             // offset into lw (doubles)
    int8_t target[VIO_MAXK];     // target frame of observation k
    int8_t tslot[VIO_MAXK];      // pattern-local block of observation k's target
    int8_t cam_block[VIO_MAXNB]; // camera block id (0 ext, 1+f pose f) of pattern-local block p
    int8_t btype[VIO_MAXNB];     // 0 ext, 1 host, 2 target
    int8_t bk[VIO_MAXNB];        // observation index k of a target block
    int32_t n_rows;              // 6*nbp pair rows + 3*nb vector rows
    int32_t lds_doubles;         // dynamic LDS this item needs
};


// per-item slab layout (doubles): pair blocks [nbp][36] | b_dir[6nb] | b_corr[6nb] | diag_dir[6nb] | chi | maxh
__host__ __device__ inline int item_nbp(int nb) { return nb * (nb + 1) / 2; }
__host__ __device__ inline int item_pair_index(int nb, int p, int q) { return p * nb - p * (p - 1) / 2 + (q - p); }
__host__ __device__ inline int item_out_count(int nb) { return item_nbp(nb) * 36 + 18 * nb + 2; }
// per-landmark storage (doubles) kept for back-substitution: w[6nb] | hinv | bl ; field-major inside an item: [field][g]
__host__ __device__ inline int item_lw_fields(int nb) { return 6 * nb + 2; }

// ---- LDS layout of a k_linearize item (doubles): shared by the kernel's carve-up and the host's item sizing (vio_plan.cpp) ----
// threads of a k_linearize / k_linearize_xyz workgroup.  One workgroup holds a CU (its LDS), so the register budget of a thread is
// 512 / (LIN_THREADS / 256): 128 at 1024 threads (16 waves, 4 per SIMD), 170 at 768 (12 waves, 3 per SIMD).
#ifndef LIN_THREADS
#define LIN_THREADS 1024
#endif
__host__ __device__ inline int lin_rrow(int use_ext) { return use_ext ? 38 : 26; }      // row record stride (doubles): 2 x 6 per block + (z0, z1)
__host__ __device__ inline int lin_raux(int use_ext) { return use_ext ? 15 : 9; }       // per-observation partials (odd stride)
__host__ __device__ inline int lin_plane(int G, int use_ext) { return G * lin_rrow(use_ext) + 6; }
__host__ __device__ inline int lin_lrec(int nb) { return 6 * nb + 7; }                  // landmark record stride (odd): w, 1/h, b_l, h, lambda, GN terms
#define LIN_VS 8            // landmark splits of the vector sums of phase 2
// tiles of phase 2: K direct products (1 tile of 16x16, 3 with the extrinsic) + the lower tiles of the 6nb x 6nb Schur term
__host__ __device__ inline int lin_tiles(int K, int nb, int use_ext) {
    const int ts = (6 * nb + 16) >> 4;         // (one row past the 6 nb columns: the Schur correction of b rides there)
    return K * (use_ext ? 3 : 1) + ts * (ts + 1) / 2;
}
// total dynamic LDS of an item (doubles); must match the carve-up in k_linearize
__host__ __device__ inline int lin_lds_doubles(int G, int K, int nb, int use_ext) {
    int aux = G * K * lin_raux(use_ext), part = lin_tiles(K, nb, use_ext) * 256;
    int shared = aux > part ? aux : part;
    const int stage = (6 * nb + 2) * G;         // the GN / LM head stages the item's Schur rows here (more than the per-observation
    if (stage > shared) shared = stage;         // records of K <= 2 observations hold)
    shared = (shared + 1) & ~1;
    return VIO_MAXK * PAIR_STRIDE + 16 + 3 * (LIN_THREADS / 64) + K * lin_plane(G, use_ext) + G * lin_lrec(nb) + shared;
}


struct LmState {
    double lambda;               // currentLambda_
    double chi;                  // currentChi_
    double ni;                   // ni_
    double last_chi;             // last_chi_ of Problem::Solve
    double chi_try;              // tempChi of the last trial
    double rho;                  // gain ratio of the last trial
    double scale;
    double init_chi;
    int32_t cur;                 // which copy of state/invd/prior is current
    int32_t accepted;            // result of the last IsGoodStepInLM
    int32_t stop;
    int32_t iter;                // outer iterations finished
    int32_t false_cnt;
    int32_t trials;
    int32_t naccepted;
    int32_t need_linearize;      // 1: the reduced system is stale (a step was accepted)
    int32_t finite;
    int32_t max_iter;
    int32_t stop_reason;
    int32_t pending;             // vio_solve's loop: a step has been taken (trial copy written) and waits for its test
    int32_t sys;                 // vio_solve's loop: which set of (lw, Pg, perm, bfull) holds the system linearised at copy `cur`
    int32_t pad_;
    double chi_trace[128];
    double lambda_trace[128];
};

struct DeviceTables {            // everything a kernel needs, passed by value
    const ItemDesc *items;
    int32_t n_items;
    int32_t n_imu_items;         // IMU edges appended to the linearize grid (replicated on every shard: their terms are added after the exchange)
    int32_t Ns;                  // landmarks in sorted order
    int32_t lm_dim;              // 1: inverse depths (k_linearize, k_backsub); 3: XYZ points (k_linearize_xyz, k_backsub_xyz)
    int32_t ext_fixed;
    int32_t loss_type;
    int32_t marg_mode;           // 1: Problem::Marginalize assembly (ext free, no fixed masking)
    double loss_delta;
    double sqrt_info;
    double gravity[3];
    double *state;               // [2][STATE_STRIDE]
    double *invd;                // [2][Ns]
    const double *pts_i;         // [Ns][2]
    const double *pts_j;         // [M][2]
    double *pairtab;             // [2][PAIRTAB_STRIDE]
    double *slab;
    double *lw;
    int64_t lw_set;              // doubles between the two sets of lw (vio_solve's loop linearises a trial state into the other set, so that a
                                 // rejected step leaves the system it came from intact); Pg, perm and bfull have two sets too (fixed strides)
    double *vis;                 // [VIS_COUNT]
    const double *pre;           // [10][PRE_STRIDE] packed pre-integrations + information
    const int32_t *imu_valid;    // [10]
    double *imu_out;             // [10][IMU_OUT]
    double *imu_chi_try;         // [10]
    const int16_t *pair_slot;    // [n_items][78]
    const int8_t *blk_slot;      // [n_items][12]
    const double *Hprior;        // [171x171]
    double *bprior;              // [2][176]
    double *errprior;            // [2][160]
    const double *Jtinv;         // [156x156]
    int32_t has_prior;
    int32_t natural_hs;          // 1: k_assemble also writes H_pp_schur_ in natural order (getters, marginalisation)
    int32_t add_imu_prior;       // 1 on every rank: IMU and prior terms are replicated and added after the exchange
    double *Hs;                  // [171x171]
    double *Pg;                  // permuted, padded, packed lower triangle + rhs row, written by k_assemble
    int32_t *perm;               // [176] pivot order found by k_assemble
    int32_t *rank;               // [176] its inverse (batched launches: k_rank_b writes both, k_assemble_b reads them)
    double *bs;                  // [171]
    double *bfull;               // [171] pose part of b_ (direct + imu + prior), for the gain ratio
    double *diagfull;            // [171] diag(Hessian_) pose part, for lambda_0
    double *dx;                  // [176] pose part of delta_x_
    double *dxl;                 // [Ns]  landmark part of delta_x_
    double *step_part;           // [n_step_blocks][2]
    double *chi_part;            // [n_step_blocks][2] partials of vio_chi2 (kept apart from a pending step test)
    int32_t n_step_blocks;
    int32_t gn_flags;            // GN mode, for the PREVIOUS step: bit 0: k_assemble runs its test (its chi2 is the one this linearisation
                                 // computed); bit 1: k_linearize applies its landmark back-substitution first; bit 2 (k_pose_solve):
                                 // this step's prior update is left to the next k_linearize (b_prior') and k_reduce (err_prior');
                                 // bit 3 (k_backsub): flush of such a step, form b_prior' here; bit 4 (k_linearize, GN loop): the grid's first
                                 // workgroup eliminates the speed-bias chain (and forms the IMU items: no IMU workgroups follow the items)
                                 // bit 5 (k_pose_solve_c, diagnostic: VIO_NO_EARLY_START): the prologue with its barriers instead of the early start
    int32_t cur_hint;            // >= 0: LmState.cur as the host tracks it through GN iterations (kernels skip the dependent load); -1: read lm->cur;
                                 // -2 (vio_solve's loop): the copy to linearise at is lm->cur ^ lm->pending, and bits 0 / 1 of gn_flags
                                 // count only while lm->pending
    int32_t imu_mask;            // bit k: IMU edge k exists (the host's copy of imu_valid: saves the kernels a dependent load)
    int32_t lm_gate;             // device-driven LM loop: 0 run; 2: skip if lm->stop; 3: skip unless lm->need_linearize && !lm->stop
    const int32_t *list_off;     // k_reduce's inverted lists (a batched launch builds its ReduceTables from here)
    const int32_t *list;
    double *step_tot;            // [8] exchange buffer: chi2 of the trial state, gain-ratio scale
    double *sp_part;             // [4] three-launch path: the wave partials of sum dx (lambda dx + b) of the step k_pose_solve_c just took
                                 //     (the pose part of the gain ratio's denominator), read by the next launch's step test
    // Sharded windows (multi-GPU): the exchange is an ALL-GATHER of every shard's vis[0 .. VIS_SEND) resp. step_tot[0 .. 2) into
    // these rank-major buffers, and whoever reads a sum forms it in rank order (d_vis, d_step_tot): every rank computes the
    // identical bits whatever algorithm the collective library picks.  n_shards == 0: unsharded, vis / step_tot are read directly.
    const double *gath;          // [n_shards][VIS_SEND]
    const double *step_gath;     // [n_shards][2]
    int32_t n_shards;
    int32_t solve_order;         // 0: Eigen's pivot order (k_assemble / k_pose_solve); 1: the static chain order (k_assemble_c / k_pose_solve_c, vio_pose_solve_chain.h)
    double *cfi;                 // the GN loop's pre-eliminated speed-bias chain (vio_pose_solve_chain.h: d_chain_pre_item -> k_pose_solve_cs), laid out
                                 // as the solve's LDS image
    double *prior_simg;          // H_prior's entries of the speed-bias rows at their places in the chain image (k_prior_simg), CH_OFF_CC doubles
    int32_t *prior_flags;        // [76 tiles of the chain image | 99 speed-bias rows]: H_prior has a non-zero entry there (k_prior_simg)
    int32_t *prior_list;         // k_prior_compact: [0] n non-zero entries of prior_simg, [1] m speed-bias rows of H_prior with a non-zero, [2 .. 2 + m) those
                                 // rows, [128 .. 128 + n) the entries' positions in the image; their values: prior_cval[0 .. n)
    double *prior_cval;
    const uint32_t *imu_map;     // [10][63][9] where an element of an IMU item's 3 x 3 tile goes in the chain image (p1 | p2 << 16; 0xffff: nowhere)
    LmState *lm;
    unsigned long long *dbg;     // diagnostic builds only (-DVIO_STAMPS): [block][16] s_memtime stamps
};

struct TriTables {               // k_triangulate (FeatureManager::triangulate)
    int64_t n;
    const int32_t *start_frame;  // [n]
    const int64_t *obs_offset;   // [n + 1]
    const double *pts;           // [m][2]
    const double *poses;         // [11][7]
    const double *ext;           // [7]
    double init_depth;
    double *depth;               // [n] in/out
};

// packed pre-integration record (doubles)
#define PRE_SUMDT 0
#define PRE_DP 1
#define PRE_DQ 4
#define PRE_DV 8
#define PRE_BA 11
#define PRE_BG 14
#define PRE_JAC 17               // 15x15
#define PRE_INFO (17 + 225)      // 15x15 information = covariance^-1 (computed on the host at upload)
#define PRE_STRIDE (17 + 450 + 5)
// LDS of an IMU workgroup of k_linearize* (d_imu_item: J 450, information 225, J^T Info 450, r and Info r 32), in doubles: the floor of a
// plan's max_lds_doubles
#define IMU_ITEM_LDS_DOUBLES (450 + 225 + 450 + 32)

#endif
