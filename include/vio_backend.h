/*
 * vio_backend.h — C ABI of the MI355X sliding-window VIO backend.
 *
 * This is the drop-in boundary for the hot path of the reference's hand-written
 * Eigen LM/Schur solver `myslam::backend::Problem` as it is driven by
 * `Estimator::problemSolve / MargOldFrame / MargNewFrame`.
 *
 * Reference shorthands (read-only tree, cited as file:line):
 *   VM/ = workspace/assignments/17-vins-initialization/vins-mono/
 *
 * Three shared libraries export this same surface with different prefixes, so the
 * parity tests call all of them with identical arguments:
 *   vio_*   libvio_hip.so      HIP/gfx950 product path (this header's prototypes)
 *   vioo_*  liboracle.so       plain-C CPU restatement (test infrastructure only)
 *   vior_*  oracle/_ref/...    harness around the compiled reference Problem (tests, this container)
 *
 * Conventions
 *   - plain pointers and sizes; every input is copied at the call, every output is written
 *     into caller-provided buffers; no pointer is retained after return.
 *   - all floating point is IEEE fp64; all matrices are dense ROW-MAJOR.
 *   - pose layout  (x,y,z,qx,qy,qz,qw)            VM/src/estimator.cpp:505-547 (vector2double)
 *     speed-bias   (vx,vy,vz,bax,bay,baz,bgx,bgy,bgz)
 *   - pose-block ordering of the reduced system (dimension VIO_POSE_DIM = 171):
 *       [ ext(6) | pose_0(6) sb_0(9) | ... | pose_10(6) sb_10(9) ]
 *     VM/src/backend/problem.cc:256-285 (SetOrdering) with the vertex creation order of
 *     VM/src/estimator.cpp:915-953.  Landmark k has ordering id 171 + k.
 *   - one context = one HIP stream = one caller thread at a time (the reference backend is not
 *     re-entrant either: VM/src/System.cpp:358-441).
 *   - every function returns a vio_status; nothing throws or aborts across the ABI.
 */
#ifndef VIO_BACKEND_H
#define VIO_BACKEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is what it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define VIO_WINDOW_SIZE 10                      /* VM/include/parameters.h:35 */
#define VIO_NUM_FRAMES (VIO_WINDOW_SIZE + 1)    /* 11 poses + 11 speed-biases */
#define VIO_POSE_DIM (6 + 15 * VIO_NUM_FRAMES)  /* 171, VM/src/estimator.cpp:931,942,952 */
#define VIO_PRIOR_DIM (VIO_POSE_DIM - 15)       /* 156, output of Problem::Marginalize */
#define VIO_CAM_DIM (6 + 6 * VIO_NUM_FRAMES)    /* 72: ext + 11 poses, the columns visual factors touch */

struct vio_ctx;                                 /* opaque; owns all device memory behind the handle */
typedef struct vio_ctx vio_ctx;

typedef enum {
    VIO_OK = 0,
    VIO_ERR_BAD_ARG = -1,
    VIO_ERR_HIP = -2,          /* a HIP runtime call failed; see vio_last_error() */
    VIO_ERR_NOT_FINITE = -3,   /* chi2 or the linear solve produced a non-finite value; from vio_solve: such trials were
                                * rejected as in problem.cc:559, the states are the last accepted ones */
    VIO_ERR_EMPTY = -4,        /* Problem::Solve returns false on an empty graph, problem.cc:172-175 */
    VIO_ERR_UNSUPPORTED = -5,  /* graph shape outside what the window layout can express */
    VIO_ERR_NO_DEVICE = -6
} vio_status;

typedef enum {                 /* VM/include/backend/loss_function.h:23-91 */
    VIO_LOSS_TRIVIAL = 0,
    VIO_LOSS_HUBER = 1,
    VIO_LOSS_CAUCHY = 2,       /* the one Estimator uses, delta = 1.0 (estimator.cpp:905) */
    VIO_LOSS_TUKEY = 3
} vio_loss_type;

typedef enum { VIO_ITEMS_LATENCY = 0, VIO_ITEMS_THROUGHPUT = 1 } vio_item_policy;
/* Elimination order of the damped pose solve (H_pp_schur + lambda I).ldlt().solve(b), VM/src/backend/problem.cc:434-439
 * (vio_set_solve_order; HIP library only):
 *   VIO_ORDER_EIGEN (0)  Eigen's LDLT pivot order (a sort of the diagonal, Cholesky/LDLT.h:317-320), operation for operation;
 *   VIO_ORDER_CHAIN (1)  default: a static order that follows the structure of the system — the 11 speed-bias blocks as a
 *                        block-tridiagonal chain eliminated from both ends, then the dense 72-variable camera block — unpivoted,
 *                        with correctly rounded quotients.  Same answer to the solver's rounding (closer to the exact solution
 *                        than Eigen's own vectors on the golden systems, tests/golden/ldlt_exact.npz); 40 % less time. */
typedef enum { VIO_ORDER_EIGEN = 0, VIO_ORDER_CHAIN = 1 } vio_solve_order;

/* Version of this interface: 3 = the sharded exchange is an all-gather (round 3); 4 = vio_set_solve_order (round 4);
 * 5 = vio_map_observations / vio_commit_observations; 6 = vio_set_imu_all. */
#define VIO_ABI_VERSION 7

typedef enum {
    VIO_MARG_OLD = 0,          /* Estimator::MargOldFrame  estimator.cpp:693-829 */
    VIO_MARG_SECOND_NEW = 1    /* Estimator::MargNewFrame  estimator.cpp:830-901 */
} vio_marg_kind;

typedef struct vio_config {
    int32_t device;            /* HIP device ordinal (ignored by the CPU libraries) */
    int32_t ext_fixed;         /* 1 == vertexExt->SetFixed(), i.e. ESTIMATE_EXTRINSIC == 0 (estimator.cpp:921-926) */
    int32_t loss_type;         /* vio_loss_type applied to every reprojection edge */
    int32_t item_policy;       /* how the landmarks are cut into workgroup items: VIO_ITEMS_LATENCY (0, default) — the fewest rounds of
                                * workgroups on the device's compute units, then the smallest items: what one window alone wants;
                                * VIO_ITEMS_THROUGHPUT (1) — the largest items the LDS holds: fewer, longer workgroups, 17 % less time
                                * per window when many windows share the device (vio_batch_*).  Results differ in the last bits
                                * between the two (another grouping of the same sums). */
    double loss_delta;         /* CauchyLoss(1.0) in the reference */
    double reproj_sqrt_info;   /* s in project_sqrt_info_ = s*I2, reference s = 460/1.5; the edge information is
                                  s^2*I2 (estimator.cpp:42,1012) */
    double gravity[3];         /* global G read by IntegrationBase::evaluate (integration_base.h:178-180) */
    void *stream;              /* optional hipStream_t to enqueue on (NULL: the library creates its own) */
    int32_t shard_rank;        /* landmark shard index of this context (multi-GPU), 0 when unsharded */
    int32_t shard_count;       /* number of shards; IMU + prior terms are replicated: every rank adds them after the exchange */
} vio_config;

/* State of one IntegrationBase as EdgeImu consumes it (VM/include/factor/integration_base.h:160-208,
 * VM/src/backend/edge_imu.cc:13-156).  Offsets inside jacobian/covariance follow StateOrder
 * O_P=0,O_R=3,O_V=6,O_BA=9,O_BG=12 (VM/include/parameters.h:75-82). */
typedef struct vio_preint {
    double sum_dt;
    double delta_p[3];
    double delta_q[4];         /* x,y,z,w */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double jacobian[225];      /* 15x15 row-major */
    double covariance[225];    /* 15x15 row-major; information = covariance^-1 (edge_imu.cc:35) */
} vio_preint;

typedef struct vio_solve_report {
    int32_t iterations;        /* outer LM iterations executed (problem.cc:188) */
    int32_t trials;            /* SolveLinearSystem calls (accepted + rejected) */
    int32_t accepted;          /* accepted steps */
    int32_t stop_reason;       /* 0: iteration cap, 1: chi2 decrease < 1e-5 (problem.cc:239) */
    double initial_chi2;
    double final_chi2;         /* currentChi_ */
    double final_lambda;       /* currentLambda_ */
    double solve_ms;           /* wall clock of the whole call, "problem solve cost" (problem.cc:246) */
    double hessian_ms;         /* device time spent linearising, "makeHessian cost" (problem.cc:247) */
    double chi2_trace[128];    /* currentChi_ at the top of each outer iteration */
    double lambda_trace[128];
} vio_solve_report;

/* ---- lifecycle -------------------------------------------------------------------------- */
vio_status vio_create(const vio_config *cfg, struct vio_ctx **out);
void vio_destroy(struct vio_ctx *ctx);
const char *vio_last_error(const struct vio_ctx *ctx);   /* valid until the next call on ctx */
void vio_default_config(vio_config *cfg);                /* the reference's constants */
/* Change what a Problem decides per graph — ext_fixed (SetFixed), the loss and its delta, the edge information, gravity —
 * on a living context, so that one context (its stream and device buffers) serves every Problem a process builds
 * (problem_hip.cc keeps one).  device, stream and the shard fields must be the ones the context was created with. */
vio_status vio_set_config(struct vio_ctx *ctx, const vio_config *cfg);

/* ---- graph construction: replaces AddVertex/AddEdge in estimator.cpp:909-1034 ----------- */
/* para_Pose[11][7], para_SpeedBias[11][9], para_Ex_Pose[0][7]  (estimator.h:113-119) */
vio_status vio_set_window(struct vio_ctx *ctx, const double *poses, const double *speed_bias,
                          const double *ext);
/* VertexInverseDepth x N, in creation order (estimator.cpp:988-993); N is bounded by device memory only */
vio_status vio_set_landmarks(struct vio_ctx *ctx, int64_t n, const double *inv_depth);
/* EdgeReprojection x M (estimator.cpp:996-1016): edge e connects landmark lm[e], host frame host[e],
 * target frame target[e] and the extrinsic; pts_i/pts_j are the normalised (x,y) of the two
 * observations (z == 1 is implied, feature_manager.h).  All edges of one landmark must share
 * host[e] and pts_i (they do in the reference: one host observation per feature).  A list that is refused (an index out
 * of range) leaves the context without observations. */
vio_status vio_set_observations(struct vio_ctx *ctx, int64_t m, const int32_t *lm,
                                const int32_t *host, const int32_t *target,
                                const double *pts_i_xy, const double *pts_j_xy);
/* The same list, written in place: vio_map_observations hands out the library's own arrays for m edges (the HIP library's are the
 * ones its uploads leave from: pts_j_xy is pinned host memory) for the caller's loop over its tracks (estimator.cpp:975-1016) to fill,
 * vio_commit_observations checks and adopts what was written — one copy of the 44 bytes per edge instead of two.  The arrays are the
 * library's; they stay valid until the next vio_set_observations / vio_map_observations / vio_set_landmarks* / vio_destroy.  Between
 * map and commit the context holds no observation list (any call that needs one fails with VIO_ERR_BAD_ARG).  The ways out of a
 * mapping: vio_commit_observations; vio_map_observations again; vio_set_observations, which drops what was written in place and sets
 * its own list.  vio_set_landmarks* with another landmark count invalidates the mapping (the indices would refer to another landmark
 * set): the commit then fails with VIO_ERR_BAD_ARG and leaves the context without a list.  (VIO_ABI_VERSION 5.) */
vio_status vio_map_observations(struct vio_ctx *ctx, int64_t m, int32_t **lm, int32_t **host, int32_t **target,
                                double **pts_i_xy, double **pts_j_xy);
vio_status vio_commit_observations(struct vio_ctx *ctx);
/* ---- 3-D landmarks: VertexPointXYZ (VM/include/backend/vertex_point_xyz.h:16) observed through EdgeReprojectionXYZ
 *      (VM/src/backend/edge_reprojection.cc:130-180; VM/include/backend/edge_reprojection.h:56-83) instead of inverse
 *      depths in a host frame.  The landmark blocks of Hmm are 3x3; Problem inverts them with the generic
 *      Hmm.block(..).inverse() of problem.cc:421-425 (no lambda on landmarks).  A context holds ONE kind of landmark:
 *      vio_set_landmarks_xyz switches it to this kind (and drops the observation list), vio_set_landmarks switches back.
 *      Everything else of the window (poses, speed-biases, IMU edges, prior, the 171-dim pose ordering with the
 *      extrinsic vertex first) is unchanged; the extrinsic is a constant of these edges (SetTranslationImuFromCamera,
 *      edge_reprojection.cc:142-145), not one of their vertices, so it gets no visual information.
 *      In this mode the landmark arrays of vio_get_delta / vio_get_landmark_system hold 3 (bl, delta) resp. 9 (H_ll,
 *      row-major 3x3) doubles per landmark.  vio_marginalize(VIO_MARG_OLD) is Problem::Marginalize({pose_0, sb_0}) on the
 *      window's graph (generic over the landmark dimension, problem.cc:617-795; the reference's Estimator has no caller
 *      for it): it keeps the edges connected to the marginalised pose (:621) — IMU edge 0 -> 1 and the frame-0 observation
 *      of every landmark seen from there — so every landmark block it eliminates has the rank 2 of a single observation
 *      and is inverted all the same (:697-700).  The visual part of the result is therefore what rounding leaves of a term
 *      that is zero in exact arithmetic, in the reference as here; a block whose elimination meets an
 *      exact zero gives the reference's NaN prior and VIO_ERR_NOT_FINITE. */
/* VertexPointXYZ x N: world coordinates xyz[n][3] */
vio_status vio_set_landmarks_xyz(struct vio_ctx *ctx, int64_t n, const double *xyz);
/* EdgeReprojectionXYZ x M: edge e connects landmark lm[e] and the pose of frame[e]; pts_xy is the normalised
 * observation (obs_, z == 1).  At most one observation of a landmark per frame. */
vio_status vio_set_observations_xyz(struct vio_ctx *ctx, int64_t m, const int32_t *lm, const int32_t *frame,
                                    const double *pts_xy);
vio_status vio_get_landmarks_xyz(struct vio_ctx *ctx, int64_t n, double *xyz);

/* EdgeImu between frames k and k+1, k in [0,10) (estimator.cpp:956-970).  pre == NULL removes the
 * edge (the reference skips it when sum_dt > 10). */
vio_status vio_set_imu(struct vio_ctx *ctx, int32_t k, const vio_preint *pre);
/* All ten edges in one call: pre[k] is the pre-integration between frames k and k + 1, or NULL for no edge — the loop of
 * estimator.cpp:956-970 as one crossing of the boundary.  (VIO_ABI_VERSION 6.) */
vio_status vio_set_imu_all(struct vio_ctx *ctx, const vio_preint *const *pre);
/* SetHessianPrior/SetbPrior/SetErrPrior/SetJtPrior + ExtendHessiansPriorSize(15)
 * (estimator.cpp:1023-1034, problem.cc:82-91).  dim is 0 (no prior yet) or VIO_PRIOR_DIM;
 * H is dim x dim, b and err have dim entries, jt_inv is dim x dim. */
vio_status vio_set_prior(struct vio_ctx *ctx, int32_t dim, const double *H, const double *b,
                         const double *err, const double *jt_inv);

/* ---- the solve: Problem::Solve(iterations)  problem.cc:169-250 ---------------------------- */
vio_status vio_solve(struct vio_ctx *ctx, int32_t iterations, vio_solve_report *report);

/* ---- single steps of the LM loop (the reference's private methods), for parity tests,
 *      the multi-GPU driver and the benchmark ------------------------------------------------- */
vio_status vio_linearize(struct vio_ctx *ctx);                      /* SetOrdering + MakeHessian  problem.cc:303-389 */
vio_status vio_init_lm(struct vio_ctx *ctx, double *chi2, double *lambda); /* ComputeLambdaInitLM  :497-522 */
vio_status vio_solve_linear(struct vio_ctx *ctx, double lambda);    /* SolveLinearSystem          :394-451 */
vio_status vio_update_states(struct vio_ctx *ctx);                  /* UpdateStates               :453-480 */
vio_status vio_rollback_states(struct vio_ctx *ctx);                /* RollbackStates             :482-494 */
/* 0.5*(sum RobustChi2 + ||err_prior||) at the current states (problem.cc:549-556) */
vio_status vio_chi2(struct vio_ctx *ctx, double *chi2);
/* IsGoodStepInLM (problem.cc:541-573) with the context's own currentChi_/currentLambda_/ni_ */
vio_status vio_eval_step(struct vio_ctx *ctx, int32_t *accepted, double *chi2, double *lambda);
/* One fixed-lambda Gauss-Newton iteration, fully enqueued with no host round trip:
 * linearise -> reduce -> IMU + prior -> damped LDLT -> back-substitute -> update -> chi2.
 * This is BASELINE.json's "GN iteration" (SURVEY.md section 8d). */
vio_status vio_gn_iteration(struct vio_ctx *ctx, double lambda);
vio_status vio_synchronize(struct vio_ctx *ctx);
/* The stream the context enqueues on (its own, or vio_config.stream). */
vio_status vio_get_stream(struct vio_ctx *ctx, void **stream);
/* vio_gn_iteration for `count` independent windows at once: one launch per kernel for the whole batch (the grid's second
 * dimension is the window), i.e. `count` pose solves side by side instead of one workgroup on one of the 256 compute units.
 * Every window is an ordinary context (its own graph, states, prior, LmState) and is read back with the ordinary
 * getters; the contexts must share one device and one stream (create the others with vio_config.stream = the first one's
 * vio_get_stream) and hold one kind of landmark (all inverse depths or all XYZ points).  Results are bit-identical to calling
 * vio_gn_iteration on each.
 * Not a reference entry point: the reference solves one window at a time (System::ProcessBackEnd, one estimator). */
vio_status vio_batch_gn_iteration(struct vio_ctx *const *ctxs, int32_t count, double lambda);
/* vio_solve (Problem::Solve, problem.cc:169-250) for `count` independent windows at once, the same way: every kernel of the
 * device-driven LM loop is launched once for the whole batch, every window follows its own LmState (its lambda, its accept /
 * reject decisions, its stop), and the host reads the states once per batch of slots.  Same requirements on the contexts as
 * vio_batch_gn_iteration; results are bit-identical to vio_solve on each.  reports: `count` entries, or NULL.
 * VIO_ERR_NOT_FINITE if any window ended non-finite (its report says which). */
vio_status vio_batch_solve(struct vio_ctx *const *ctxs, int32_t count, int32_t iterations, vio_solve_report *reports);

/* ---- IMU pre-integration (host side, as in the reference: Estimator::processIMU -> IntegrationBase::push_back ->
 *      propagate -> midPointIntegration, integration_base.h:30-36,54-158).  Starts from (acc0, gyr0) — the sample the
 *      IntegrationBase constructor takes — and folds in `count` further samples; noise densities as ACC_N, GYR_N,
 *      ACC_W, GYR_W of the YAML (vio_simulation.yaml:60-63).  No context needed. --------------------------------- */
vio_status vio_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int32_t count,
                            const double *dt, const double *acc, const double *gyr, double acc_n, double gyr_n,
                            double acc_w, double gyr_w, vio_preint *out);

/* ---- triangulation of new tracks: FeatureManager::triangulate  VM/src/feature_manager.cpp:203-257 ----
 * Depth (in its first camera frame) of every track that has none yet (depth[i] <= 0 on entry): the smallest right
 * singular vector v of the 2K x 4 DLT system built from its K observations (:228-240), depth = v[2] / v[3] (:245); a
 * result below 0.1 becomes init_depth (INIT_DEPTH, parameters.cpp:126; :252-255).  Tracks the optimiser would not use
 * (fewer than 2 observations, or start_frame >= VIO_WINDOW_SIZE - 2; :207-208) and tracks that already have a depth
 * (:210-211) are left alone.
 *   start_frame[i]                      frame of the track's first observation (its host frame)
 *   obs_offset[i] .. obs_offset[i+1]    its observations, one per consecutive frame from start_frame on
 *   pts[e][2]                           normalised image point (x, y), z = 1 (the reference normalises the 3-vector, :236)
 *   poses[11][7], ext[7]                p, q(xyzw) of the IMU frames; camera-to-IMU extrinsic
 * One GPU thread per track; depth[] is updated in place on the host. */
vio_status vio_triangulate(struct vio_ctx *ctx, int64_t n_tracks, const int32_t *start_frame, const int64_t *obs_offset,
                           const double *pts, const double *poses, const double *ext, double init_depth, double *depth);

/* ---- marginalisation: Marg{Old,New}Frame + Problem::Marginalize  problem.cc:617-795 ------ */
/* Uses the window/landmarks/observations/IMU/prior currently set.  Outputs the new prior
 * (VIO_PRIOR_DIM): H dim x dim, b, err, jt_inv dim x dim.
 * VIO_ERR_NOT_FINITE: a landmark of the marginalisation graph has a block without an inverse (every edge weighted to zero
 * by the loss; a rank-deficient 3x3 block whose elimination met an exact zero pivot).  The reference's dense
 * Hpm * Hmm^-1 (problem.cc:701-703) is then NaN throughout, its eigen-solvers return NaN spectra, every `> eps` test fails
 * and Marginalize returns true with H_prior_ = 0 and b_prior_, err_prior_, Jt_prior_inv_ all NaN: the outputs hold exactly
 * that, and the status says so. */
vio_status vio_marginalize(struct vio_ctx *ctx, int32_t kind, double *H, double *b, double *err,
                           double *jt_inv);
/* The same in two halves: vio_marginalize == vio_marginalize_begin + vio_marginalize_end.
 * begin returns when the device part is done (graph assembly, landmark Schur complement, the 171x171 read-back) and the dense
 * tail (problem.cc:717-779: two eigen-decompositions and three products, 0.2 - 0.5 ms on a host core) runs on a helper thread of
 * the library; end waits for it and hands the prior out.  In between the context is the caller's as usual: the frame loop calls
 * begin where MargOldFrame stood (estimator.cpp:1088-1092), goes on with slideWindow, the next image, the next window's
 * vio_set_window / landmarks / observations / imu, and calls end where the new prior is first needed — in front of vio_set_prior
 * (estimator.cpp:1023-1034).  A second begin, or vio_destroy, waits for an unfinished tail itself. */
vio_status vio_marginalize_begin(struct vio_ctx *ctx, int32_t kind);
vio_status vio_marginalize_end(struct vio_ctx *ctx, double *H, double *b, double *err, double *jt_inv);
/* What the next solve needs of the graph set so far — the grouping of the landmarks into workgroup items, its tables and the
 * observations on the device (the host side of Problem's AddVertex / AddEdge / SetOrdering, estimator.cpp:909-1016, problem.cc:256-285) —
 * built and uploaded NOW instead of inside the next vio_linearize / vio_solve.  Optional: a frame loop that runs its marginalisation in
 * the background calls it between the vio_set_window / landmarks / observations / imu of the new frame and vio_marginalize_end, so that
 * the planner's 0.06 - 0.2 ms run under the dense tail too; vio_set_prior afterwards only sends the prior.  Results are the same with and
 * without it.  (VIO_ABI_VERSION 7.) */
vio_status vio_prepare(struct vio_ctx *ctx);

/* ---- read back ------------------------------------------------------------------------------ */
vio_status vio_get_window(struct vio_ctx *ctx, double *poses, double *speed_bias, double *ext);
vio_status vio_get_landmarks(struct vio_ctx *ctx, int64_t n, double *inv_depth);
/* b_prior_ (VIO_POSE_DIM entries) and err_prior_ (VIO_PRIOR_DIM) after the first-order updates
 * (estimator.cpp:1040-1049) */
vio_status vio_get_prior(struct vio_ctx *ctx, double *b, double *err);
/* delta_x_: pose part (171) and landmark part (n) of the last SolveLinearSystem */
vio_status vio_get_delta(struct vio_ctx *ctx, double *dx_pose, int64_t n, double *dx_landmarks);
/* H_pp_schur_ WITHOUT the lambda of problem.cc:434-436, and b_pp_schur_ (171x171, 171): of the last vio_linearize /
 * vio_gn_iteration.  VIO_ERR_BAD_ARG when the context holds no linearisation of its current state — after vio_solve in
 * particular, whose last linearisation may be a rejected trial's: call vio_linearize first. */
vio_status vio_get_schur_system(struct vio_ctx *ctx, double *H, double *b);
/* Hmm diagonal and bmm (n entries each), and the pose part of b_ / diag(Hessian_) (171 each) */
vio_status vio_get_landmark_system(struct vio_ctx *ctx, int64_t n, double *hll, double *bl);
vio_status vio_get_pose_gradient(struct vio_ctx *ctx, double *b_pose, double *diag_pose);

/* ---- multi-GPU exchange -------------------------------------------
 * The landmarks of a window are sharded over the ranks; what the ranks exchange per linearisation is every shard's packed
 * partial reduced system (`reduced_system`, `reduced_count` fp64: 24 KB) and, on the stepwise path, two scalars per trial.
 * The exchange is an ALL-GATHER into rank-major receive buffers (vio_gather_buffers: [shard_count][reduced_count] and
 * [shard_count][scalar_count]), and the library itself adds the ranks' slabs in RANK ORDER wherever it reads a sum: every
 * rank performs the same additions in the same order, so all ranks hold bit-identical systems and take bit-identical LM
 * decisions whatever algorithm the collective library picks (an all-reduce would leave the summation order to it). */
/* Device pointer + element count of the send side: this shard's packed fp64 slab (between vio_linearize and
 * vio_solve_linear) and its two step scalars (after vio_update_states: chi2 + gain-ratio scale). */
vio_status vio_exchange_buffers(struct vio_ctx *ctx, void **reduced_system, int64_t *reduced_count,
                                void **step_scalars, int64_t *scalar_count);
/* The receive side: rank r's slab belongs at gathered_system + r * reduced_count, its scalars at gathered_scalars + r * scalar_count. */
vio_status vio_gather_buffers(struct vio_ctx *ctx, void **gathered_system, void **gathered_scalars);
/* Hook called on the host, stream-ordered, wherever the LM loop needs an exchange.  which == 0: all-gather the reduced
 * systems, 1: all-gather the step scalars (both: send buffer -> receive buffers of ALL ranks, rank-major);
 * which == 2 (CPU libraries only): all-reduce MAX of step_scalars[2] in place.  NULL (default) = unsharded. */
typedef int (*vio_exchange_fn)(void *user, int which);
/* The hook's contract changed with VIO_ABI_VERSION 3 (an in-place all-reduce before, the all-gather above since): a caller checks
 * vio_abi_version() >= 3, and the library refuses to run a hook (VIO_ERR_BAD_ARG at the first exchange) until the caller has called
 * vio_gather_buffers or vio_bind_gather_buffers on the context — which a hook of the old contract never does. */
vio_status vio_set_exchange_hook(struct vio_ctx *ctx, vio_exchange_fn fn, void *user);
int32_t vio_abi_version(void);

/* Native exchange: the library all-gathers its exchange buffers itself with RCCL (xGMI inside a node), in stream order
 * on its own stream, with no host callback in the loop.  RCCL is dlopen'ed (librccl.so of the process, e.g. the one
 * PyTorch loaded); the caller only distributes the 128-byte id of rank 0 to all ranks (torch.distributed / MPI /
 * a socket) and every rank calls vio_comm_init with it.  Replaces the hook when both are set.
 *   vio_comm_unique_id   ncclGetUniqueId, rank 0 only: fills id128[128]
 *   vio_comm_init        ncclCommInitRank on the context's device: collective over all `nranks` processes
 *   vio_comm_destroy     ncclCommDestroy (also done by vio_destroy) */
vio_status vio_comm_unique_id(void *id128);
vio_status vio_comm_init(struct vio_ctx *ctx, const void *id128, int32_t rank, int32_t nranks);
vio_status vio_comm_destroy(struct vio_ctx *ctx);
/* The communicator as RCCL reports it: ranks it spans (ncclCommCount) and this rank's index (ncclCommUserRank); 0 / -1 without one. */
vio_status vio_comm_info(struct vio_ctx *ctx, int32_t *nranks_seen, int32_t *rank_seen);

/* Optional: make the library use caller-owned device memory for the exchange buffers (e.g. torch tensors, so that
 * torch.distributed can gather them in place).  reduced must hold >= the count reported by vio_exchange_buffers + 8
 * doubles, scalars >= 8 doubles; gathered_system >= shard_count * reduced_count, gathered_scalars >= shard_count *
 * scalar_count doubles.  NULL restores the library's own buffer. */
vio_status vio_bind_exchange_buffers(struct vio_ctx *ctx, void *reduced_system, void *step_scalars);
vio_status vio_bind_gather_buffers(struct vio_ctx *ctx, void *gathered_system, void *gathered_scalars);

/* ---- measurement (HIP library only) ------------------------------------------------------------ */
typedef enum {
    VIO_K_LINEARIZE = 0, VIO_K_REDUCE = 1, VIO_K_ASSEMBLE = 2, VIO_K_POSE_SOLVE = 3, VIO_K_BACKSUB = 4,
    VIO_K_LM_DECIDE = 5, VIO_K_COUNT = 6
} vio_kernel_id;
/* Wrap every launch of kernel `which` in a hipEvent pair recorded on the context's stream (which < 0: off).  On the
 * leader of a batch the pairs go around that kernel's batched launch in vio_batch_gn_iteration. */
vio_status vio_profile_begin(struct vio_ctx *ctx, int32_t which);
/* The same, but only every `every`-th launch gets its event pair (every >= 1).  An event record drains the stream
 * (about 7 us each on MI355X), so bench.py samples inside its timed region instead of bracketing every launch. */
vio_status vio_profile_begin_sampled(struct vio_ctx *ctx, int32_t which, int32_t every);
/* Synchronise, sum the elapsed times of the recorded pairs, and stop profiling. */
vio_status vio_profile_end(struct vio_ctx *ctx, double *total_ms, int64_t *launches);
const char *vio_kernel_name(int32_t which);
/* Where the host time of the context's last calls went, microseconds (wall clock inside the library):
 *   out8[0..2]  the last activation of a plan: read-back of what the device held newer, plan build + its upload, state upload
 *   out8[3]     the last vio_marginalize: activation + kernels + the 171x171 read-back      out8[4]  its dense host tail
 *   out8[5]     rows of the reduced 156x156 system that were not exactly zero in that tail (the eigen-problem's size)
 *   out8[6]     the marginalisation plan prepared under the last vio_solve (0: nothing to prepare) */
vio_status vio_get_host_timing(struct vio_ctx *ctx, double *out8);

/* ---- elimination order of the pose solve (HIP library only) -------------------------------------- */
/* Choose vio_solve_order for the context's solves from now on (default VIO_ORDER_CHAIN; the environment variable
 * VIO_SOLVE_ORDER=eigen|chain sets the default of new contexts).  A prior whose H couples two speed-bias blocks that are not
 * neighbours — no Problem::Marginalize output does (problem.cc:617-795 reaches the speed-bias of frame 1 only) — is solved in
 * Eigen's order whatever was asked for; vio_get_solve_order reports the order asked for and the one in effect. */
vio_status vio_set_solve_order(struct vio_ctx *ctx, int32_t order);
vio_status vio_get_solve_order(struct vio_ctx *ctx, int32_t *requested, int32_t *effective);
#ifdef VIO_DEBUG_ENTRY_POINTS
/* Diagnostic, NOT part of the product ABI (exported only by a build with -DVIO_DEBUG_ENTRY_POINTS: csrc/diag/libvio_hip_debug.so, which
 * the tests build): x = (H + lambda I)^-1 b by the chain-order kernel alone on a caller-supplied 171 x 171 row-major H (natural order of
 * H_pp_schur_; only its lower triangle is read).  lds_dump: NULL, or room for the factor as the kernel leaves it
 * (tools/chain_solve_model.py documents the layout). */
vio_status vio_debug_chain_solve(struct vio_ctx *ctx, const double *H, const double *b, double lambda, double *x, double *lds_dump);
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* VIO_BACKEND_H */
