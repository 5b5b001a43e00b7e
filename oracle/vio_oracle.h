/*
 * vio_oracle.h — prototypes of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle exports the very same C ABI as include/vio_backend.h under the prefix `vioo_`
 * so that tests call `vio_*` (HIP), `vioo_*` (this restatement) and `vior_*` (the compiled
 * reference, oracle/_ref) with identical arguments.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.
 */
#ifndef VIO_ORACLE_H
#define VIO_ORACLE_H

#define vio_ctx vioo_ctx
#define vio_create vioo_create
#define vio_destroy vioo_destroy
#define vio_last_error vioo_last_error
#define vio_default_config vioo_default_config
#define vio_set_config vioo_set_config
#define vio_set_window vioo_set_window
#define vio_set_landmarks vioo_set_landmarks
#define vio_set_observations vioo_set_observations
#define vio_set_imu vioo_set_imu
#define vio_set_imu_all vioo_set_imu_all
#define vio_set_prior vioo_set_prior
#define vio_solve vioo_solve
#define vio_linearize vioo_linearize
#define vio_prepare vioo_prepare
#define vio_init_lm vioo_init_lm
#define vio_solve_linear vioo_solve_linear
#define vio_update_states vioo_update_states
#define vio_rollback_states vioo_rollback_states
#define vio_chi2 vioo_chi2
#define vio_eval_step vioo_eval_step
#define vio_gn_iteration vioo_gn_iteration
#define vio_synchronize vioo_synchronize
#define vio_marginalize vioo_marginalize
#define vio_marginalize_begin vioo_marginalize_begin
#define vio_marginalize_end vioo_marginalize_end
#define vio_get_window vioo_get_window
#define vio_get_landmarks vioo_get_landmarks
#define vio_get_prior vioo_get_prior
#define vio_get_delta vioo_get_delta
#define vio_get_schur_system vioo_get_schur_system
#define vio_get_landmark_system vioo_get_landmark_system
#define vio_get_pose_gradient vioo_get_pose_gradient
#define vio_exchange_buffers vioo_exchange_buffers
#define vio_set_exchange_hook vioo_set_exchange_hook
#define vio_bind_exchange_buffers vioo_bind_exchange_buffers
#define vio_gather_buffers vioo_gather_buffers
#define vio_bind_gather_buffers vioo_bind_gather_buffers
#define vio_profile_begin vioo_profile_begin
#define vio_profile_end vioo_profile_end
#define vio_kernel_name vioo_kernel_name
#define vio_preintegrate vioo_preintegrate_abi
#define vio_triangulate vioo_triangulate
#define vio_set_landmarks_xyz vioo_set_landmarks_xyz
#define vio_set_observations_xyz vioo_set_observations_xyz
#define vio_map_observations vioo_map_observations
#define vio_commit_observations vioo_commit_observations
#define vio_get_landmarks_xyz vioo_get_landmarks_xyz
#include "../include/vio_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- pieces exposed individually so tests can pin them one by one ------------------------- */

/* EdgeReprojection::ComputeResidual + ComputeJacobians (edge_reprojection.cc:18-109).
 * J_* are row-major: J_lambda 2x1, J_pose_i 2x6, J_pose_j 2x6, J_ext 2x6. Any J pointer may be NULL. */
void vioo_reproj_edge(const double *pose_i, const double *pose_j, const double *ext, double inv_depth,
                      const double *pts_i_xy, const double *pts_j_xy, double *residual,
                      double *J_lambda, double *J_pose_i, double *J_pose_j, double *J_ext);

/* EdgeReprojectionXYZ::ComputeResidual + ComputeJacobians (edge_reprojection.cc:130-180): landmark pw (world xyz)
 * seen from the body pose `pose` through the camera extrinsic `ext`.  J_feature row-major 2x3, J_pose 2x6; may be NULL. */
void vioo_reproj_xyz_edge(const double *pose, const double *ext, const double *pw, const double *obs_xy,
                          double *residual, double *J_feature, double *J_pose);

/* MatXX::inverse() of a 3x3 block as problem.cc:424 runs it: Eigen's dynamic-size path, PartialPivLU (unblocked,
 * LU/PartialPivLU.h) then the two triangular solves of the identity.  Row-major in/out. */
void vioo_inverse3(const double *A, double *Ainv);

/* IntegrationBase::evaluate + EdgeImu::ComputeJacobians (integration_base.h:160-186,
 * edge_imu.cc:38-156). Jacobians row-major 15x6, 15x9, 15x6, 15x9; may be NULL. */
void vioo_imu_edge(const vio_preint *pre, const double *gravity, const double *pose_i,
                   const double *sb_i, const double *pose_j, const double *sb_j, double *residual,
                   double *J_pose_i, double *J_sb_i, double *J_pose_j, double *J_sb_j);

/* covariance.inverse() for a fixed 15x15 (Eigen PartialPivLU path), row-major in/out */
void vioo_inverse15(const double *cov, double *info);

/* LossFunction::Compute (loss_function.cc:9-47): rho[0..2] = rho, rho', rho'' of e2 */
void vioo_loss(int loss_type, double delta, double e2, double *rho);

/* Edge::RobustInfo for a 2-D residual with information s^2*I (edge.cc:48-74): W row-major 2x2 */
void vioo_robust_info2(int loss_type, double delta, double sqrt_info, const double *r, double *drho,
                       double *W);

/* VertexPose::Plus (vertex_pose.cc:7-19) on a 7-vector pose, delta 6 */
void vioo_pose_plus(double *pose, const double *delta);

/* Eigen::LDLT<MatrixXd>(A).solve(b) (Cholesky/LDLT.h:291-400,558-600), A row-major n x n (lower read) */
void vioo_ldlt_solve(int n, const double *A, const double *b, double *x, int *transpositions);

/* Symmetric eigen-decomposition (lower triangle read): eigenvalues ascending, V column k = k-th vector,
 * V row-major n x n. Stands where Problem::Marginalize calls Eigen::SelfAdjointEigenSolver. */
int vioo_symmetric_eigen(int n, const double *A, double *evals, double *V);

/* IntegrationBase constructor + push_back loop (integration_base.h:14-158): mid-point
 * pre-integration of `count` samples (dt[k], acc[k], gyr[k]) starting from (acc0, gyr0). */
void vioo_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg,
                       int count, const double *dt, const double *acc, const double *gyr,
                       double acc_n, double gyr_n, double acc_w, double gyr_w, vio_preint *out);

/* Schur complement of the trailing m2 x m2 block with the eigen pseudo-inverse of problem.cc:747-764 */
void vioo_schur_pinv(int n, int m2, const double *H, const double *b, double *Hp, double *bp);

/* dense pose block of Hessian_ incl. prior (171x171) — oracle only, the HIP path never forms it */
vio_status vioo_get_pose_hessian(struct vioo_ctx *ctx, double *Hpp);

#ifdef __cplusplus
}
#endif
#endif
