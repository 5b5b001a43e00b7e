// host_dense.h — small dense host routines of the product path (C++).
//   inverse15        covariance.inverse() of an IMU edge, done once per upload (edge_imu.cc:35 does it per evaluation)
//   marginalize_tail the dense tail of Problem::Marginalize (problem.cc:717-779): move the marginalised frame
//                    to the bottom, eigen-decomposition pseudo-inverse Schur complement, re-factor the new prior
#ifndef VIO_HOST_DENSE_H
#define VIO_HOST_DENSE_H

namespace vio_host {

void inverse15(const double *cov, double *info);

// Helper threads for the two routines below, behind a function pointer (this file knows no thread pool): run_n(ctx, want, fn, arg) calls
// fn(arg, i, n) for i = 0 .. n - 1 — i = 0 on the caller, n <= want, n = 1 when no helper is free — and returns when all are done.
// nullptr: everything on the caller.  The results do not depend on it: bit for bit the same whatever n turns out to be.
struct Par {
    void *ctx;
    void (*run_n)(void *ctx, int want, void (*fn)(void *arg, int i, int n), void *arg);
    int width;
};

// Symmetric eigen-decomposition (lower triangle is read). evals ascending, V row-major, column k = k-th vector.
bool symmetric_eigen(int n, const double *A, double *evals, double *V, const Par *par = nullptr);
// (the routine as it was until round 6 — the rotations applied to the eigenvector matrix inside the QL loop, one thread: the A/B reference)
bool symmetric_eigen_legacy(int n, const double *A, double *evals, double *V);

// H (171x171 row-major) and b (171) hold H_marg/b_marg AFTER the landmark Schur complement and AFTER the old
// prior has been added.  frame = index of the frame whose pose (6) and speed-bias (9) are marginalised.
// Outputs: Hout 156x156, bout 156, errout 156, jtout 156x156.
// Returns the number of rows of the reduced 156x156 system that were not exactly zero (the size of the eigen-problem solved).
int marginalize_tail(double *H, double *b, int frame, double *Hout, double *bout, double *errout, double *jtout, const Par *par = nullptr);

// IntegrationBase mid-point propagation (integration_base.h:54-158): count samples after (acc0, gyr0).
// out_* : sum_dt, delta_p[3], delta_q[4] (xyzw), delta_v[3], jacobian[225], covariance[225] (row-major)
void preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int count, const double *dt,
                  const double *acc, const double *gyr, double acc_n, double gyr_n, double acc_w, double gyr_w,
                  double *sum_dt, double *delta_p, double *delta_q, double *delta_v, double *jacobian, double *covariance);

}  // namespace vio_host
#endif
