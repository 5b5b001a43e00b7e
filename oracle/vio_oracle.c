/*
 * vio_oracle.c — CPU restatement of the reference's sliding-window backend, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path and the
 * `cpu_baseline` ("port") leg of bench.py.  Nothing in the product path may link, load or
 * call it.
 *
 * Parity status: PINNED for the solver (Problem::{MakeHessian,SolveLinearSystem,UpdateStates,
 * IsGoodStepInLM,Solve,Marginalize}), the reprojection factor, the vertices and the loss
 * functions — checked against the reference's own sources compiled in oracle/_ref (see
 * oracle/Makefile, tests/test_oracle_vs_reference.py, tests/golden/).
 * UNPINNED for the IMU factor arithmetic (IntegrationBase::evaluate/midPointIntegration,
 * EdgeImu::ComputeJacobians): integration_base.h:6 includes <ceres/ceres.h>, which this image
 * lacks, so that translation unit cannot be built here; those functions are restated from the
 * source text and checked by central finite differences and algebraic identities only.
 *
 * What it restates (VM/ = workspace/assignments/17-vins-initialization/vins-mono/):
 *   VM/src/backend/edge_reprojection.cc:18-109   reprojection residual + Jacobians
 *   VM/include/factor/integration_base.h:14-186  mid-point pre-integration, evaluate
 *   VM/src/backend/edge_imu.cc:13-156            IMU residual + Jacobians
 *   VM/src/backend/edge.cc:33-74                 Chi2 / RobustChi2 / RobustInfo
 *   VM/src/backend/loss_function.cc:9-47         Huber / Cauchy / Tukey
 *   VM/src/backend/vertex_pose.cc:7-19           pose Plus (Sophus exp, so3.hpp:393-419)
 *   VM/src/backend/problem.cc:169-573            LM loop, Hessian, Schur, LDLT, update, chi2
 *   VM/src/backend/problem.cc:617-795            Marginalize
 *   VM/src/estimator.cpp:693-1073                graph construction of problemSolve/Marg*Frame
 *
 * The one deliberate difference from the reference: the landmark Schur complement is formed in
 * O(M) from the per-landmark 1x1 blocks instead of dense (171+N)^2 products.  The arithmetic per
 * entry is the same (Hpp - (Hpm*Hmm^-1)*Hmp), only zero terms are skipped.
 */
#define _POSIX_C_SOURCE 200809L
#include "vio_oracle.h"

#include <float.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
/* threads of the all-cores build: the work of one window saturates at about 16 (measured on the GPU box's host: 4.7 ms with
 * 8, 3.6 ms with 16, 4.8 ms with 32 threads at 20 000 landmarks; the private 72x72 accumulators cost 83 KB a thread) */
static int oracle_threads(void) { int n = omp_get_max_threads(); return n < 16 ? n : 16; }
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* layout of the exchange buffer, identical to the HIP library's (csrc/vio_types.h VIS_*) */
#define VIS_H 0
#define VIS_BRED (VIO_CAM_DIM * VIO_CAM_DIM)
#define VIS_BDIR (VIS_BRED + VIO_CAM_DIM)
#define VIS_DIAG (VIS_BDIR + VIO_CAM_DIM)
#define VIS_CHI (VIS_DIAG + VIO_CAM_DIM)
#define VIS_MAXH (VIS_CHI + 3)          /* two slots after VIS_CHI are reserved (the HIP library's GN step scalars) */
#define VIS_N (VIS_MAXH + 1 + 4)

#define NF VIO_NUM_FRAMES
#define PD VIO_POSE_DIM
#define PRD VIO_PRIOR_DIM
#define CD VIO_CAM_DIM

/* ------------------------------------------------------------------------------------------ */
/* small math, written the way Eigen 3.3 evaluates the same expressions                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct { double x, y, z, w; } quat;

static quat q_from_pose(const double *p) { quat q = {p[3], p[4], p[5], p[6]}; return q; }

/* Eigen::QuaternionBase::operator* */
static quat q_mul(quat a, quat b) {
    quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

/* Eigen::QuaternionBase::inverse(): conjugate / squaredNorm */
static quat q_inv(quat q) {
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    quat r = {0, 0, 0, 0};
    if (n2 > 0) { r.x = -q.x / n2; r.y = -q.y / n2; r.z = -q.z / n2; r.w = q.w / n2; }
    return r;
}

/* Eigen::QuaternionBase::_transformVector */
static void q_rot(quat q, const double *v, double *out) {
    double ux = q.y * v[2] - q.z * v[1];
    double uy = q.z * v[0] - q.x * v[2];
    double uz = q.x * v[1] - q.y * v[0];
    ux += ux; uy += uy; uz += uz;
    double cx = q.y * uz - q.z * uy;
    double cy = q.z * ux - q.x * uz;
    double cz = q.x * uy - q.y * ux;
    out[0] = v[0] + q.w * ux + cx;
    out[1] = v[1] + q.w * uy + cy;
    out[2] = v[2] + q.w * uz + cz;
}

/* Eigen::QuaternionBase::toRotationMatrix, row-major 3x3 */
static void q_to_R(quat q, double *R) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static void m3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void m3_T(const double *A, double *T) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}
static void m3_vec(const double *A, const double *v, double *o) {
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
/* Utility::skewSymmetric == Sophus::SO3::hat (utility.h:27-35) */
static void skew(const double *v, double *S) {
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* loss functions: loss_function.cc:9-47, trivial loss loss_function.h:40-50                   */
/* ------------------------------------------------------------------------------------------ */
void vioo_loss(int type, double delta, double e2, double *rho) {
    switch (type) {
    case VIO_LOSS_HUBER: {
        double dsqr = delta * delta;
        if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
        else {
            double sqrte = sqrt(e2);
            rho[0] = 2 * sqrte * delta - dsqr;
            rho[1] = delta / sqrte;
            rho[2] = -0.5 * rho[1] / e2;
        }
        break;
    }
    case VIO_LOSS_CAUCHY: {
        double dsqr = delta * delta;
        double dsqrReci = 1. / dsqr;
        double aux = dsqrReci * e2 + 1.0;
        rho[0] = dsqr * log(aux);
        rho[1] = 1. / aux;
        rho[2] = -dsqrReci * (rho[1] * rho[1]);
        break;
    }
    case VIO_LOSS_TUKEY: {
        double e = sqrt(e2);
        double delta2 = delta * delta;
        if (e <= delta) {
            double aux = e2 / delta2;
            rho[0] = delta2 * (1. - (1. - aux) * (1. - aux) * (1. - aux)) / 3.;
            rho[1] = (1. - aux) * (1. - aux);
            rho[2] = -2. * (1. - aux) / delta2;
        } else { rho[0] = delta2 / 3.; rho[1] = 0; rho[2] = 0; }
        break;
    }
    default: rho[0] = e2; rho[1] = 1; rho[2] = 0; break;
    }
}

/* Edge::RobustInfo (edge.cc:48-74) for information = s^2*I2, sqrt_information = s*I2.
 * With the trivial loss the reference still goes through the lossfunction_ branch only when a
 * loss object is set; VIO_LOSS_TRIVIAL here means "no loss object" (drho = 1, W = information). */
void vioo_robust_info2(int type, double delta, double s, const double *r, double *drho, double *W) {
    double info = s * s;
    if (type == VIO_LOSS_TRIVIAL) {
        *drho = 1.0; W[0] = info; W[1] = 0; W[2] = 0; W[3] = info; return;
    }
    double e2 = r[0] * (info * r[0]) + r[1] * (info * r[1]);   /* Edge::Chi2, edge.cc:33-37 */
    double rho[3];
    vioo_loss(type, delta, e2, rho);
    double w0 = s * r[0], w1 = s * r[1];                        /* sqrt_information_ * residual_ */
    double ri[4] = {rho[1], 0, 0, rho[1]};
    if (rho[1] + 2 * rho[2] * e2 > 0.) {
        double c = 2 * rho[2];
        ri[0] += c * w0 * w0; ri[1] += c * w0 * w1; ri[2] += c * w1 * w0; ri[3] += c * w1 * w1;
    }
    W[0] = ri[0] * info; W[1] = ri[1] * info; W[2] = ri[2] * info; W[3] = ri[3] * info;
    *drho = rho[1];
}

static double robust_chi2_2(int type, double delta, double s, const double *r) {
    double info = s * s;
    double e2 = r[0] * (info * r[0]) + r[1] * (info * r[1]);
    if (type == VIO_LOSS_TRIVIAL) return e2;
    double rho[3];
    vioo_loss(type, delta, e2, rho);
    return rho[0];
}

/* ------------------------------------------------------------------------------------------ */
/* EdgeReprojection: edge_reprojection.cc:18-109                                               */
/* ------------------------------------------------------------------------------------------ */
void vioo_reproj_edge(const double *pose_i, const double *pose_j, const double *ext, double inv_dep_i,
                      const double *pi_xy, const double *pj_xy, double *residual, double *J_l,
                      double *J_i, double *J_j, double *J_e) {
    quat Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j), qic = q_from_pose(ext);
    const double *Pi = pose_i, *Pj = pose_j, *tic = ext;
    double pts_i[3] = {pi_xy[0], pi_xy[1], 1.0};

    double pc_i[3] = {pts_i[0] / inv_dep_i, pts_i[1] / inv_dep_i, pts_i[2] / inv_dep_i};
    double pb_i[3], pw[3], d[3], pb_j[3], e[3], pc_j[3];
    q_rot(qic, pc_i, pb_i);
    for (int k = 0; k < 3; ++k) pb_i[k] += tic[k];
    q_rot(Qi, pb_i, pw);
    for (int k = 0; k < 3; ++k) pw[k] += Pi[k];
    for (int k = 0; k < 3; ++k) d[k] = pw[k] - Pj[k];
    q_rot(q_inv(Qj), d, pb_j);
    for (int k = 0; k < 3; ++k) e[k] = pb_j[k] - tic[k];
    q_rot(q_inv(qic), e, pc_j);

    double dep_j = pc_j[2];
    if (residual) {
        residual[0] = pc_j[0] / dep_j - pj_xy[0];
        residual[1] = pc_j[1] / dep_j - pj_xy[1];
    }
    if (!J_l && !J_i && !J_j && !J_e) return;

    double Ri[9], Rj[9], ric[9], ricT[9], RjT[9];
    q_to_R(Qi, Ri); q_to_R(Qj, Rj); q_to_R(qic, ric);
    m3_T(ric, ricT); m3_T(Rj, RjT);
    double reduce[6] = {1. / dep_j, 0, -pc_j[0] / (dep_j * dep_j),
                        0, 1. / dep_j, -pc_j[1] / (dep_j * dep_j)};
    double A[9];            /* ric^T * Rj^T */
    m3_mul(ricT, RjT, A);
    double ARi[9];          /* ric^T * Rj^T * Ri */
    m3_mul(A, Ri, ARi);

    if (J_i) {
        double H[9], nH[9], right[9];
        skew(pb_i, H);
        for (int k = 0; k < 9; ++k) nH[k] = -H[k];
        m3_mul(ARi, nH, right);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                J_i[6 * r + c] = reduce[3 * r] * A[c] + reduce[3 * r + 1] * A[3 + c] + reduce[3 * r + 2] * A[6 + c];
                J_i[6 * r + 3 + c] = reduce[3 * r] * right[c] + reduce[3 * r + 1] * right[3 + c] + reduce[3 * r + 2] * right[6 + c];
            }
    }
    if (J_j) {
        double H[9], right[9], left[9];
        skew(pb_j, H);
        m3_mul(ricT, H, right);
        for (int k = 0; k < 9; ++k) left[k] = -A[k];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                J_j[6 * r + c] = reduce[3 * r] * left[c] + reduce[3 * r + 1] * left[3 + c] + reduce[3 * r + 2] * left[6 + c];
                J_j[6 * r + 3 + c] = reduce[3 * r] * right[c] + reduce[3 * r + 1] * right[3 + c] + reduce[3 * r + 2] * right[6 + c];
            }
    }
    if (J_l) {
        /* reduce * ric^T * Rj^T * Ri * ric * pts_i * -1.0 / (inv_dep_i * inv_dep_i) */
        double T[9], v[3];
        m3_mul(ARi, ric, T);
        m3_vec(T, pts_i, v);
        for (int r = 0; r < 2; ++r) {
            double s = reduce[3 * r] * v[0] + reduce[3 * r + 1] * v[1] + reduce[3 * r + 2] * v[2];
            J_l[r] = s * -1.0 / (inv_dep_i * inv_dep_i);
        }
    }
    if (J_e) {
        double RjTRi[9], M[9], left[9];
        m3_mul(RjT, Ri, RjTRi);
        for (int k = 0; k < 9; ++k) M[k] = RjTRi[k];
        M[0] -= 1; M[4] -= 1; M[8] -= 1;
        m3_mul(ricT, M, left);
        double tmp_r[9];
        m3_mul(ARi, ric, tmp_r);
        double S1[9], t1[9], v2[3], S2[9], u[3], w[3], x[3], S3[9], right[9];
        skew(pc_i, S1);
        m3_mul(tmp_r, S1, t1);
        m3_vec(tmp_r, pc_i, v2);
        skew(v2, S2);
        /* ric^T * (Rj^T * (Ri * tic + Pi - Pj) - tic) */
        m3_vec(Ri, tic, u);
        for (int k = 0; k < 3; ++k) u[k] = u[k] + Pi[k] - Pj[k];
        m3_vec(RjT, u, w);
        for (int k = 0; k < 3; ++k) w[k] -= tic[k];
        m3_vec(ricT, w, x);
        skew(x, S3);
        for (int k = 0; k < 9; ++k) right[k] = -t1[k] + S2[k] + S3[k];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                J_e[6 * r + c] = reduce[3 * r] * left[c] + reduce[3 * r + 1] * left[3 + c] + reduce[3 * r + 2] * left[6 + c];
                J_e[6 * r + 3 + c] = reduce[3 * r] * right[c] + reduce[3 * r + 1] * right[3 + c] + reduce[3 * r + 2] * right[6 + c];
            }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* EdgeReprojectionXYZ: edge_reprojection.cc:130-180                                            */
/* ------------------------------------------------------------------------------------------ */
void vioo_reproj_xyz_edge(const double *pose, const double *ext, const double *pw, const double *obs_xy,
                          double *residual, double *J_f, double *J_p) {
    quat Qi = q_from_pose(pose), qic = q_from_pose(ext);
    const double *Pi = pose, *tic = ext;
    double d[3], pts_imu[3], e[3], pc[3];
    for (int k = 0; k < 3; ++k) d[k] = pw[k] - Pi[k];
    q_rot(q_inv(Qi), d, pts_imu);                       /* Qi.inverse() * (pts_w - Pi) */
    for (int k = 0; k < 3; ++k) e[k] = pts_imu[k] - tic[k];
    q_rot(q_inv(qic), e, pc);                           /* qic.inverse() * (pts_imu_i - tic) */
    const double dep = pc[2];
    if (residual) {
        residual[0] = pc[0] / dep - obs_xy[0];
        residual[1] = pc[1] / dep - obs_xy[1];
    }
    if (!J_f && !J_p) return;
    double Ri[9], ric[9], RiT[9], ricT[9];
    q_to_R(Qi, Ri); q_to_R(qic, ric);
    m3_T(Ri, RiT); m3_T(ric, ricT);
    const double reduce[6] = {1. / dep, 0, -pc[0] / (dep * dep),
                              0, 1. / dep, -pc[1] / (dep * dep)};
    if (J_p) {
        /* jaco_i = [ric^T * -Ri^T | ric^T * hat(pts_imu_i)], jacobian_pose_i = reduce * jaco_i */
        double nRiT[9], left[9], H[9], right[9];
        for (int k = 0; k < 9; ++k) nRiT[k] = -RiT[k];
        m3_mul(ricT, nRiT, left);
        skew(pts_imu, H);
        m3_mul(ricT, H, right);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) {
                J_p[6 * r + c] = reduce[3 * r] * left[c] + reduce[3 * r + 1] * left[3 + c] + reduce[3 * r + 2] * left[6 + c];
                J_p[6 * r + 3 + c] = reduce[3 * r] * right[c] + reduce[3 * r + 1] * right[3 + c] + reduce[3 * r + 2] * right[6 + c];
            }
    }
    if (J_f) {
        /* jacobian_feature = (reduce * ric^T) * Ri^T */
        double rr[6];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c)
                rr[3 * r + c] = reduce[3 * r] * ricT[c] + reduce[3 * r + 1] * ricT[3 + c] + reduce[3 * r + 2] * ricT[6 + c];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c)
                J_f[3 * r + c] = rr[3 * r] * RiT[c] + rr[3 * r + 1] * RiT[3 + c] + rr[3 * r + 2] * RiT[6 + c];
    }
}

/* Hmm.block(idx, idx, 3, 3).inverse() (problem.cc:424).  The block is a dynamic-size expression, so Eigen takes
 * compute_inverse<.., Dynamic>: PartialPivLU (LU/PartialPivLU.h, unblocked_lu for sizes <= 16: first largest |entry| of the
 * column as pivot, the column divided by it, rank-1 update of the rest) and inverse() = solve(Identity):
 * P * I, then the unit-lower and the upper triangular solves of TriangularSolverMatrix.h (column by column,
 * the diagonal applied as a multiplication by its reciprocal). */
void vioo_inverse3(const double *A, double *Ainv) {
    double lu[9];
    int piv[3];
    for (int k = 0; k < 9; ++k) lu[k] = A[k];
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double big = fabs(lu[3 * k + k]);
        for (int i = k + 1; i < 3; ++i) if (fabs(lu[3 * i + k]) > big) { big = fabs(lu[3 * i + k]); best = i; }
        piv[k] = best;
        if (big != 0.0) {
            if (best != k) for (int j = 0; j < 3; ++j) { double t = lu[3 * k + j]; lu[3 * k + j] = lu[3 * best + j]; lu[3 * best + j] = t; }
            for (int i = k + 1; i < 3; ++i) lu[3 * i + k] /= lu[3 * k + k];
        }
        for (int i = k + 1; i < 3; ++i)
            for (int j = k + 1; j < 3; ++j) lu[3 * i + j] -= lu[3 * i + k] * lu[3 * k + j];
    }
    double X[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 3; ++k)
        if (piv[k] != k) for (int j = 0; j < 3; ++j) { double t = X[3 * k + j]; X[3 * k + j] = X[3 * piv[k] + j]; X[3 * piv[k] + j] = t; }
    for (int j = 0; j < 3; ++j) {
        for (int i = 0; i < 3; ++i) {                   /* unit lower: forward */
            const double b = X[3 * i + j];
            for (int r = i + 1; r < 3; ++r) X[3 * r + j] -= b * lu[3 * r + i];
        }
        for (int i = 2; i >= 0; --i) {                  /* upper: backward, a = 1 / tri(i, i) */
            const double a = 1.0 / lu[3 * i + i];
            const double b = (X[3 * i + j] *= a);
            for (int r = 0; r < i; ++r) X[3 * r + j] -= b * lu[3 * r + i];
        }
    }
    for (int k = 0; k < 9; ++k) Ainv[k] = X[k];
}

/* ------------------------------------------------------------------------------------------ */
/* IMU factor: integration_base.h:160-186 (evaluate), edge_imu.cc:38-156 (Jacobians)           */
/* ------------------------------------------------------------------------------------------ */
#define O_P 0
#define O_R 3
#define O_V 6
#define O_BA 9
#define O_BG 12

static void jac_block(const double *J15, int r0, int c0, double *B) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = J15[15 * (r0 + i) + c0 + j];
}
/* Utility::deltaQ (utility.h:11-24): (1, theta/2), NOT normalised */
static quat delta_q(const double *theta) {
    quat q = {theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0, 1.0};
    return q;
}
/* bottom-right 3x3 of Utility::Qleft(q): w*I + skew(vec)  (utility.h:48-56) */
static void qleft_br(quat q, double *B) {
    double v[3] = {q.x, q.y, q.z}, S[9];
    skew(v, S);
    for (int k = 0; k < 9; ++k) B[k] = S[k];
    B[0] += q.w; B[4] += q.w; B[8] += q.w;
}
/* bottom-right 3x3 of Utility::Qright(p): w*I - skew(vec)  (utility.h:58-66) */
static void qright_br(quat q, double *B) {
    double v[3] = {q.x, q.y, q.z}, S[9];
    skew(v, S);
    for (int k = 0; k < 9; ++k) B[k] = -S[k];
    B[0] += q.w; B[4] += q.w; B[8] += q.w;
}
/* bottom-right 3x3 of Qleft(a)*Qright(b): rows 1..3 of Qleft times cols 1..3 of Qright */
static void qleft_qright_br(quat a, quat b, double *B) {
    /* Qleft(a) = [w, -v^T; v, w I + [v]x],  Qright(b) = [w, -v^T; v, w I - [v]x]
       bottom-right of product = v_a * (-v_b^T) + (w_a I + [v_a]x)(w_b I - [v_b]x) */
    double La[9], Rb[9], P[9];
    qleft_br(a, La); qright_br(b, Rb);
    m3_mul(La, Rb, P);
    double va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[3 * i + j] = va[i] * (-vb[j]) + P[3 * i + j];
}

void vioo_imu_edge(const vio_preint *pre, const double *G, const double *pose_i, const double *sb_i,
                   const double *pose_j, const double *sb_j, double *res, double *Jpi, double *Jsi,
                   double *Jpj, double *Jsj) {
    quat Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
    const double *Pi = pose_i, *Pj = pose_j;
    const double *Vi = sb_i, *Bai = sb_i + 3, *Bgi = sb_i + 6;
    const double *Vj = sb_j, *Baj = sb_j + 3, *Bgj = sb_j + 6;
    double sum_dt = pre->sum_dt;
    quat dq = {pre->delta_q[0], pre->delta_q[1], pre->delta_q[2], pre->delta_q[3]};

    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    jac_block(pre->jacobian, O_P, O_BA, dp_dba);
    jac_block(pre->jacobian, O_P, O_BG, dp_dbg);
    jac_block(pre->jacobian, O_R, O_BG, dq_dbg);
    jac_block(pre->jacobian, O_V, O_BA, dv_dba);
    jac_block(pre->jacobian, O_V, O_BG, dv_dbg);

    double dba[3], dbg[3];
    for (int k = 0; k < 3; ++k) { dba[k] = Bai[k] - pre->linearized_ba[k]; dbg[k] = Bgi[k] - pre->linearized_bg[k]; }
    double th[3];
    m3_vec(dq_dbg, dbg, th);
    quat corrected_dq = q_mul(dq, delta_q(th));
    quat Qi_inv = q_inv(Qi);

    if (res) {
        double a[3], b[3], t[3], u[3];
        m3_vec(dv_dba, dba, a); m3_vec(dv_dbg, dbg, b);
        double cdv[3] = {pre->delta_v[0] + a[0] + b[0], pre->delta_v[1] + a[1] + b[1], pre->delta_v[2] + a[2] + b[2]};
        m3_vec(dp_dba, dba, a); m3_vec(dp_dbg, dbg, b);
        double cdp[3] = {pre->delta_p[0] + a[0] + b[0], pre->delta_p[1] + a[1] + b[1], pre->delta_p[2] + a[2] + b[2]};
        for (int k = 0; k < 3; ++k) t[k] = 0.5 * G[k] * sum_dt * sum_dt + Pj[k] - Pi[k] - Vi[k] * sum_dt;
        q_rot(Qi_inv, t, u);
        for (int k = 0; k < 3; ++k) res[O_P + k] = u[k] - cdp[k];
        quat qe = q_mul(q_inv(corrected_dq), q_mul(Qi_inv, Qj));
        res[O_R + 0] = 2 * qe.x; res[O_R + 1] = 2 * qe.y; res[O_R + 2] = 2 * qe.z;
        for (int k = 0; k < 3; ++k) t[k] = G[k] * sum_dt + Vj[k] - Vi[k];
        q_rot(Qi_inv, t, u);
        for (int k = 0; k < 3; ++k) res[O_V + k] = u[k] - cdv[k];
        for (int k = 0; k < 3; ++k) { res[O_BA + k] = Baj[k] - Bai[k]; res[O_BG + k] = Bgj[k] - Bgi[k]; }
    }

    double RiT[9];              /* Qi.inverse().toRotationMatrix() */
    q_to_R(Qi_inv, RiT);
    if (Jpi) {
        memset(Jpi, 0, sizeof(double) * 15 * 6);
        double t[3], u[3], S[9], B[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpi[6 * (O_P + i) + O_P + j] = -RiT[3 * i + j];
        for (int k = 0; k < 3; ++k) t[k] = 0.5 * G[k] * sum_dt * sum_dt + Pj[k] - Pi[k] - Vi[k] * sum_dt;
        q_rot(Qi_inv, t, u); skew(u, S);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpi[6 * (O_P + i) + O_R + j] = S[3 * i + j];
        qleft_qright_br(q_mul(q_inv(Qj), Qi), corrected_dq, B);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpi[6 * (O_R + i) + O_R + j] = -B[3 * i + j];
        for (int k = 0; k < 3; ++k) t[k] = G[k] * sum_dt + Vj[k] - Vi[k];
        q_rot(Qi_inv, t, u); skew(u, S);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpi[6 * (O_V + i) + O_R + j] = S[3 * i + j];
    }
    if (Jsi) {
        memset(Jsi, 0, sizeof(double) * 15 * 9);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Jsi[9 * (O_P + i) + (O_V - O_V) + j] = -RiT[3 * i + j] * sum_dt;
            Jsi[9 * (O_P + i) + (O_BA - O_V) + j] = -dp_dba[3 * i + j];
            Jsi[9 * (O_P + i) + (O_BG - O_V) + j] = -dp_dbg[3 * i + j];
        }
        /* -Qleft(Qj^-1 * Qi * delta_q).bottomRight * dq_dbg  — delta_q, not corrected_delta_q (edge_imu.cc:107-109) */
        double L[9], nL[9], B[9];
        qleft_br(q_mul(q_mul(q_inv(Qj), Qi), dq), L);
        for (int k = 0; k < 9; ++k) nL[k] = -L[k];
        m3_mul(nL, dq_dbg, B);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            Jsi[9 * (O_R + i) + (O_BG - O_V) + j] = B[3 * i + j];
            Jsi[9 * (O_V + i) + (O_V - O_V) + j] = -RiT[3 * i + j];
            Jsi[9 * (O_V + i) + (O_BA - O_V) + j] = -dv_dba[3 * i + j];
            Jsi[9 * (O_V + i) + (O_BG - O_V) + j] = -dv_dbg[3 * i + j];
        }
        for (int i = 0; i < 3; ++i) {
            Jsi[9 * (O_BA + i) + (O_BA - O_V) + i] = -1.0;
            Jsi[9 * (O_BG + i) + (O_BG - O_V) + i] = -1.0;
        }
    }
    if (Jpj) {
        memset(Jpj, 0, sizeof(double) * 15 * 6);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpj[6 * (O_P + i) + O_P + j] = RiT[3 * i + j];
        double L[9];
        qleft_br(q_mul(q_mul(q_inv(corrected_dq), Qi_inv), Qj), L);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jpj[6 * (O_R + i) + O_R + j] = L[3 * i + j];
    }
    if (Jsj) {
        memset(Jsj, 0, sizeof(double) * 15 * 9);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jsj[9 * (O_V + i) + (O_V - O_V) + j] = RiT[3 * i + j];
        for (int i = 0; i < 3; ++i) {
            Jsj[9 * (O_BA + i) + (O_BA - O_V) + i] = 1.0;
            Jsj[9 * (O_BG + i) + (O_BG - O_V) + i] = 1.0;
        }
    }
}

/* covariance.inverse() on a fixed 15x15: Eigen routes sizes > 4 through PartialPivLU
 * (LU/InverseImpl.h, LU/PartialPivLU.h unblocked_lu for rows <= 16), then solves against I. */
void vioo_inverse15(const double *cov, double *info) {
    enum { n = 15 };
    double lu[n * n];
    int piv[n];
    memcpy(lu, cov, sizeof(lu));
    for (int k = 0; k < n; ++k) {
        int row = k; double big = fabs(lu[n * k + k]);
        for (int i = k + 1; i < n; ++i) if (fabs(lu[n * i + k]) > big) { big = fabs(lu[n * i + k]); row = i; }
        piv[k] = row;
        if (big != 0) {
            if (row != k) for (int j = 0; j < n; ++j) { double t = lu[n * k + j]; lu[n * k + j] = lu[n * row + j]; lu[n * row + j] = t; }
            for (int i = k + 1; i < n; ++i) lu[n * i + k] /= lu[n * k + k];
        }
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) lu[n * i + j] -= lu[n * i + k] * lu[n * k + j];
    }
    for (int c = 0; c < n; ++c) {
        double x[n];
        for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k) if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) x[i] -= lu[n * i + j] * x[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) x[i] -= lu[n * i + j] * x[j];
            x[i] /= lu[n * i + i];
        }
        for (int i = 0; i < n; ++i) info[n * i + c] = x[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* vertices: vertex.cc:28-30, vertex_pose.cc:7-19, Sophus so3.hpp:393-419,683-685              */
/* ------------------------------------------------------------------------------------------ */
void vioo_pose_plus(double *p, const double *d) {
    p[0] += d[0]; p[1] += d[1]; p[2] += d[2];
    double ox = d[3], oy = d[4], oz = d[5];
    double theta_sq = ox * ox + oy * oy + oz * oz;
    double theta = sqrt(theta_sq);
    double half_theta = 0.5 * theta;
    double imag, real;
    if (theta < 1e-10) {
        double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        double s = sin(half_theta);
        imag = s / theta;
        real = cos(half_theta);
    }
    quat e = {imag * ox, imag * oy, imag * oz, real};
    /* SO3Group(const Quaternion&) normalises: Eigen normalize() = coeffs /= norm() */
    double n = sqrt(e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w);
    e.x /= n; e.y /= n; e.z /= n; e.w /= n;
    quat q = {p[3], p[4], p[5], p[6]};
    q = q_mul(q, e);          /* q.normalized() result is discarded in the reference (vertex_pose.cc:12) */
    p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
}

/* ------------------------------------------------------------------------------------------ */
/* Eigen::LDLT<MatrixXd,Lower>: Cholesky/LDLT.h:291-400 (unblocked), :558-600 (solve)          */
/* ------------------------------------------------------------------------------------------ */
void vioo_ldlt_solve(int n, const double *Ain, const double *b, double *x, int *tr_out) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    double *temp = (double *)malloc(sizeof(double) * n);
    int *tr = (int *)malloc(sizeof(int) * n);
    memcpy(A, Ain, sizeof(double) * n * n);
#define M(i, j) A[(size_t)(i) * n + (j)]
    if (n <= 1) {
        tr[0] = 0;
    } else {
        int zero_all = 0;
        for (int k = 0; k < n && !zero_all; ++k) {
            int big = k; double bv = fabs(M(k, k));
            for (int i = k + 1; i < n; ++i) if (fabs(M(i, i)) > bv) { bv = fabs(M(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {
                int s = n - big - 1;
                for (int j = 0; j < k; ++j) { double t = M(k, j); M(k, j) = M(big, j); M(big, j) = t; }
                for (int i = 0; i < s; ++i) { double t = M(n - s + i, k); M(n - s + i, k) = M(n - s + i, big); M(n - s + i, big) = t; }
                { double t = M(k, k); M(k, k) = M(big, big); M(big, big) = t; }
                for (int i = k + 1; i < big; ++i) { double t = M(i, k); M(i, k) = M(big, i); M(big, i) = t; }
            }
            int rs = n - k - 1;
            if (k > 0) {
                for (int j = 0; j < k; ++j) temp[j] = M(j, j) * M(k, j);
                double s = 0;
                for (int j = 0; j < k; ++j) s += M(k, j) * temp[j];
                M(k, k) -= s;
                for (int i = 0; i < rs; ++i) {
                    double t = 0;
                    for (int j = 0; j < k; ++j) t += M(k + 1 + i, j) * temp[j];
                    M(k + 1 + i, k) -= t;
                }
            }
            double akk = M(k, k);
            int valid = fabs(akk) > 0;
            if (k == 0 && !valid) {
                for (int j = 0; j < n; ++j) tr[j] = j;
                zero_all = 1;
                break;
            }
            if (rs > 0 && valid) for (int i = 0; i < rs; ++i) M(k + 1 + i, k) /= akk;
        }
    }
    /* solve: dst = P b; L^-1; D^+; L^-T; P^-1 */
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int k = 0; k < n; ++k) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int j = 0; j < i; ++j) s -= M(i, j) * x[j]; x[i] = s; }
    double tol = 1.0 / DBL_MAX;
    for (int i = 0; i < n; ++i) { if (fabs(M(i, i)) > tol) x[i] /= M(i, i); else x[i] = 0; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int j = i + 1; j < n; ++j) s -= M(j, i) * x[j]; x[i] = s; }
    for (int k = n - 1; k >= 0; --k) if (tr[k] != k) { double t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
#undef M
    if (tr_out) memcpy(tr_out, tr, sizeof(int) * n);
    free(A); free(temp); free(tr);
}

/* ------------------------------------------------------------------------------------------ */
/* symmetric eigen-decomposition: Householder tridiagonalisation + implicit QL (the classical   */
/* tred2/tql2 pair).  Stands where problem.cc:752,766 call Eigen::SelfAdjointEigenSolver.        */
/* ------------------------------------------------------------------------------------------ */
int vioo_symmetric_eigen(int n, const double *Ain, double *d, double *Vout) {
    double *V = (double *)malloc(sizeof(double) * n * n);
    double *e = (double *)malloc(sizeof(double) * n);
#define V_(i, j) V[(size_t)(i) * n + (j)]
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { V_(i, j) = Ain[(size_t)i * n + j]; V_(j, i) = V_(i, j); }
    /* tred2 */
    for (int j = 0; j < n; ++j) d[j] = V_(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = V_(i - 1, j); V_(i, j) = 0.0; V_(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                V_(j, i) = f;
                g = e[j] + V_(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += V_(k, j) * d[k]; e[k] += V_(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) V_(k, j) -= (f * e[k] + g * d[k]);
                d[j] = V_(i - 1, j);
                V_(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        V_(n - 1, i) = V_(i, i);
        V_(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = V_(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += V_(k, i + 1) * V_(k, j);
                for (int k = 0; k <= i; ++k) V_(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) V_(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = V_(n - 1, j); V_(n - 1, j) = 0.0; }
    V_(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    /* tql2 */
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0, eps = ldexp(1.0, -52);
    int ok = 1;
    for (int l = 0; l < n; ++l) {
        double t = fabs(d[l]) + fabs(e[l]);
        if (t > tst1) tst1 = t;
        int m = l;
        while (m < n) { if (fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { ok = 0; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, el1 = e[l + 1], s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = V_(k, i + 1);
                        V_(k, i + 1) = s * V_(k, i) + c * h;
                        V_(k, i) = c * V_(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    /* sort ascending */
    for (int i = 0; i < n - 1; ++i) {
        int k = i; double p = d[i];
        for (int j = i + 1; j < n; ++j) if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; ++j) { double t = V_(j, i); V_(j, i) = V_(j, k); V_(j, k) = t; }
        }
    }
    memcpy(Vout, V, sizeof(double) * n * n);
#undef V_
    free(V); free(e);
    return ok ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------ */
/* FeatureManager::triangulate (VM/src/feature_manager.cpp:203-257), restated.                 */
/* The reference takes the last column of V of Eigen::JacobiSVD of the 2K x 4 matrix svd_A     */
/* (:243).  FeatureManager does not compile here (parameters.h pulls in OpenCV), so this part   */
/* is PARITY UNPINNED against the reference itself; tests/test_triangulate.py pins it against   */
/* numpy.linalg.svd of the same svd_A instead.  Here: eigenvector of the smallest eigenvalue of */
/* svd_A^T svd_A through the tred2/tql2 solver above (the HIP kernel uses Jacobi rotations).    */
/* ------------------------------------------------------------------------------------------ */
vio_status vio_triangulate(struct vioo_ctx *c, int64_t n, const int32_t *start_frame, const int64_t *obs_offset,
                           const double *pts, const double *poses, const double *ext, double init_depth, double *depth) {
    (void)c;
    if (n < 0 || (n > 0 && (!start_frame || !obs_offset || !depth)) || !poses || !ext) return VIO_ERR_BAD_ARG;
    double Rc[VIO_NUM_FRAMES][9], tc[VIO_NUM_FRAMES][3], ric[9];
    quat qe = {ext[3], ext[4], ext[5], ext[6]};
    q_to_R(qe, ric);
    for (int f = 0; f < VIO_NUM_FRAMES; ++f) {
        quat q = {poses[7 * f + 3], poses[7 * f + 4], poses[7 * f + 5], poses[7 * f + 6]};
        double Rf[9], t[3];
        q_to_R(q, Rf);
        m3_mul(Rf, ric, Rc[f]);                                   /* R1 = Rs[j] * ric[0]          (:225) */
        m3_vec(Rf, ext, t);
        for (int k = 0; k < 3; ++k) tc[f][k] = poses[7 * f + k] + t[k];   /* t1 = Ps[j] + Rs[j] * tic[0]  (:224) */
    }
    for (int64_t i = 0; i < n; ++i) {
        const int sf = start_frame[i];
        const int64_t e0 = obs_offset[i];
        const int K = (int)(obs_offset[i + 1] - e0);
        if (K < 0 || sf < 0 || sf + K > VIO_NUM_FRAMES) return VIO_ERR_BAD_ARG;
        if (!(K >= 2 && sf < VIO_WINDOW_SIZE - 2)) continue;      /* :207 */
        if (depth[i] > 0) continue;                               /* :210 */
        double R0T[9], AtA[16], ev[4], V[16];
        m3_T(Rc[sf], R0T);
        memset(AtA, 0, sizeof(AtA));
        for (int j = 0; j < K; ++j) {
            const int f = sf + j;
            double dt[3], t[3], R[9], RT[9], P[3][4];
            for (int k = 0; k < 3; ++k) dt[k] = tc[f][k] - tc[sf][k];
            m3_vec(R0T, dt, t);                                   /* t = R0^T (t1 - t0)           (:230) */
            m3_mul(R0T, Rc[f], R);                                /* R = R0^T R1                  (:231) */
            m3_T(R, RT);
            for (int r = 0; r < 3; ++r) {                         /* P = [R^T | -R^T t]           (:233-234) */
                for (int k = 0; k < 3; ++k) P[r][k] = RT[3 * r + k];
                P[r][3] = -(RT[3 * r] * t[0] + RT[3 * r + 1] * t[1] + RT[3 * r + 2] * t[2]);
            }
            const double x = pts[2 * (e0 + j)], y = pts[2 * (e0 + j) + 1];
            const double nn = sqrt(x * x + y * y + 1.0);          /* f = point.normalized()        (:235) */
            const double f0 = x / nn, f1 = y / nn, f2 = 1.0 / nn;
            double ra[4], rb[4];
            for (int k = 0; k < 4; ++k) { ra[k] = f0 * P[2][k] - f2 * P[0][k]; rb[k] = f1 * P[2][k] - f2 * P[1][k]; }   /* :236-237 */
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) AtA[4 * a + b] += ra[a] * ra[b] + rb[a] * rb[b];
        }
        vioo_symmetric_eigen(4, AtA, ev, V);                      /* ascending: column 0 <-> smallest singular value */
        double dep = V[4 * 2 + 0] / V[4 * 3 + 0];                 /* svd_V[2] / svd_V[3]          (:245) */
        if (dep < 0.1) dep = init_depth;                          /* :252-255 */
        depth[i] = dep;
    }
    return VIO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* IntegrationBase: integration_base.h:14-158                                                  */
/* ------------------------------------------------------------------------------------------ */
static void mat_mul(int m, int k, int n, const double *A, const double *B, double *C) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void set_block3(double *M, int ld, int r0, int c0, const double *B, double scale) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(r0 + i) * ld + c0 + j] = B[3 * i + j] * scale;
}

void vioo_preintegrate(const double *acc_first, const double *gyr_first, const double *ba, const double *bg,
                       int count, const double *dts, const double *accs, const double *gyrs, double ACC_N,
                       double GYR_N, double ACC_W, double GYR_W, vio_preint *out) {
    double acc_0[3], gyr_0[3];
    memcpy(acc_0, acc_first, sizeof(acc_0)); memcpy(gyr_0, gyr_first, sizeof(gyr_0));
    double jac[225], cov[225], noise[18 * 18];
    memset(jac, 0, sizeof(jac)); memset(cov, 0, sizeof(cov)); memset(noise, 0, sizeof(noise));
    for (int i = 0; i < 15; ++i) jac[16 * i] = 1.0;
    for (int i = 0; i < 3; ++i) {
        noise[19 * (0 + i)] = ACC_N * ACC_N; noise[19 * (3 + i)] = GYR_N * GYR_N;
        noise[19 * (6 + i)] = ACC_N * ACC_N; noise[19 * (9 + i)] = GYR_N * GYR_N;
        noise[19 * (12 + i)] = ACC_W * ACC_W; noise[19 * (15 + i)] = GYR_W * GYR_W;
    }
    double sum_dt = 0, dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
    quat dq = {0, 0, 0, 1};
    for (int s = 0; s < count; ++s) {
        double _dt = dts[s];
        const double *acc_1 = accs + 3 * s, *gyr_1 = gyrs + 3 * s;
        double a0[3], a1[3], un_gyr[3], un_acc_0[3], un_acc_1[3], un_acc[3];
        for (int k = 0; k < 3; ++k) { a0[k] = acc_0[k] - ba[k]; a1[k] = acc_1[k] - ba[k]; }
        q_rot(dq, a0, un_acc_0);
        for (int k = 0; k < 3; ++k) un_gyr[k] = 0.5 * (gyr_0[k] + gyr_1[k]) - bg[k];
        quat inc = {un_gyr[0] * _dt / 2, un_gyr[1] * _dt / 2, un_gyr[2] * _dt / 2, 1};
        quat rq = q_mul(dq, inc);
        q_rot(rq, a1, un_acc_1);
        for (int k = 0; k < 3; ++k) un_acc[k] = 0.5 * (un_acc_0[k] + un_acc_1[k]);
        double rp[3], rv[3];
        for (int k = 0; k < 3; ++k) {
            rp[k] = dp[k] + dv[k] * _dt + 0.5 * un_acc[k] * _dt * _dt;
            rv[k] = dv[k] + un_acc[k] * _dt;
        }
        /* jacobian / covariance propagation, integration_base.h:75-127 */
        double R_w_x[9], R_a_0_x[9], R_a_1_x[9], Rd[9], Rr[9];
        skew(un_gyr, R_w_x); skew(a0, R_a_0_x); skew(a1, R_a_1_x);
        q_to_R(dq, Rd); q_to_R(rq, Rr);
        double ImW[9];          /* I - R_w_x*dt */
        for (int k = 0; k < 9; ++k) ImW[k] = -R_w_x[k] * _dt;
        ImW[0] += 1; ImW[4] += 1; ImW[8] += 1;
        double RdA0[9], RrA1[9], RrA1I[9];
        m3_mul(Rd, R_a_0_x, RdA0); m3_mul(Rr, R_a_1_x, RrA1); m3_mul(RrA1, ImW, RrA1I);
        double F[225], Vm[15 * 18];
        memset(F, 0, sizeof(F)); memset(Vm, 0, sizeof(Vm));
        double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, B[9];
        set_block3(F, 15, 0, 0, I3, 1.0);
        for (int k = 0; k < 9; ++k) B[k] = -0.25 * RdA0[k] * _dt * _dt + -0.25 * RrA1I[k] * _dt * _dt;
        set_block3(F, 15, 0, 3, B, 1.0);
        set_block3(F, 15, 0, 6, I3, _dt);
        for (int k = 0; k < 9; ++k) B[k] = -0.25 * (Rd[k] + Rr[k]) * _dt * _dt;
        set_block3(F, 15, 0, 9, B, 1.0);
        for (int k = 0; k < 9; ++k) B[k] = -0.25 * RrA1[k] * _dt * _dt * -_dt;
        set_block3(F, 15, 0, 12, B, 1.0);
        set_block3(F, 15, 3, 3, ImW, 1.0);
        set_block3(F, 15, 3, 12, I3, -1.0 * _dt);
        for (int k = 0; k < 9; ++k) B[k] = -0.5 * RdA0[k] * _dt + -0.5 * RrA1I[k] * _dt;
        set_block3(F, 15, 6, 3, B, 1.0);
        set_block3(F, 15, 6, 6, I3, 1.0);
        for (int k = 0; k < 9; ++k) B[k] = -0.5 * (Rd[k] + Rr[k]) * _dt;
        set_block3(F, 15, 6, 9, B, 1.0);
        for (int k = 0; k < 9; ++k) B[k] = -0.5 * RrA1[k] * _dt * -_dt;
        set_block3(F, 15, 6, 12, B, 1.0);
        set_block3(F, 15, 9, 9, I3, 1.0);
        set_block3(F, 15, 12, 12, I3, 1.0);

        set_block3(Vm, 18, 0, 0, Rd, 0.25 * _dt * _dt);
        for (int k = 0; k < 9; ++k) B[k] = 0.25 * -RrA1[k] * _dt * _dt * 0.5 * _dt;
        set_block3(Vm, 18, 0, 3, B, 1.0);
        set_block3(Vm, 18, 0, 6, Rr, 0.25 * _dt * _dt);
        set_block3(Vm, 18, 0, 9, B, 1.0);
        set_block3(Vm, 18, 3, 3, I3, 0.5 * _dt);
        set_block3(Vm, 18, 3, 9, I3, 0.5 * _dt);
        set_block3(Vm, 18, 6, 0, Rd, 0.5 * _dt);
        for (int k = 0; k < 9; ++k) B[k] = 0.5 * -RrA1[k] * _dt * 0.5 * _dt;
        set_block3(Vm, 18, 6, 3, B, 1.0);
        set_block3(Vm, 18, 6, 6, Rr, 0.5 * _dt);
        set_block3(Vm, 18, 6, 9, B, 1.0);
        set_block3(Vm, 18, 9, 12, I3, _dt);
        set_block3(Vm, 18, 12, 15, I3, _dt);

        double T1[225], T2[225], FT[225], VN[15 * 18], VT[18 * 15], T3[225];
        mat_mul(15, 15, 15, F, jac, T1);
        memcpy(jac, T1, sizeof(jac));
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) FT[15 * i + j] = F[15 * j + i];
        mat_mul(15, 15, 15, F, cov, T1);
        mat_mul(15, 15, 15, T1, FT, T2);
        mat_mul(15, 18, 18, Vm, noise, VN);
        for (int i = 0; i < 18; ++i) for (int j = 0; j < 15; ++j) VT[15 * i + j] = Vm[18 * j + i];
        mat_mul(15, 18, 15, VN, VT, T3);
        for (int k = 0; k < 225; ++k) cov[k] = T2[k] + T3[k];

        memcpy(dp, rp, sizeof(dp)); memcpy(dv, rv, sizeof(dv));
        dq = rq;
        double nq = sqrt(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z + dq.w * dq.w);   /* delta_q.normalize() */
        dq.x /= nq; dq.y /= nq; dq.z /= nq; dq.w /= nq;
        sum_dt += _dt;
        memcpy(acc_0, acc_1, sizeof(acc_0)); memcpy(gyr_0, gyr_1, sizeof(gyr_0));
    }
    out->sum_dt = sum_dt;
    memcpy(out->delta_p, dp, sizeof(dp)); memcpy(out->delta_v, dv, sizeof(dv));
    out->delta_q[0] = dq.x; out->delta_q[1] = dq.y; out->delta_q[2] = dq.z; out->delta_q[3] = dq.w;
    memcpy(out->linearized_ba, ba, sizeof(double) * 3); memcpy(out->linearized_bg, bg, sizeof(double) * 3);
    memcpy(out->jacobian, jac, sizeof(jac)); memcpy(out->covariance, cov, sizeof(cov));
}

/* ------------------------------------------------------------------------------------------ */
/* the Problem                                                                                 */
/* ------------------------------------------------------------------------------------------ */
struct vioo_ctx {
    vio_config cfg;
    char err[256];
    double pose[NF * 7], sb[NF * 9], ext[7];
    double pose_bak[NF * 7], sb_bak[NF * 9], ext_bak[7];
    int64_t N, M;
    int64_t M_mapped;              /* >= 0 between vio_map_observations and vio_commit_observations */
    int lm_dim;                 /* 1: VertexInverseDepth (invd[N]); 3: VertexPointXYZ (invd[N][3] holds the world points,
                                 * target[] the observing frame, pts_j the observation; hll 3x3, bl 3, Hpl 72x3 per landmark) */
    double *invd, *invd_bak;
    int32_t *lm, *host, *target;
    double *pts_i, *pts_j;
    int imu_valid[VIO_WINDOW_SIZE];
    vio_preint pre[VIO_WINDOW_SIZE];
    double imu_info[VIO_WINDOW_SIZE][225];
    /* prior */
    int has_prior;              /* err_prior_.rows() > 0 */
    double Hprior[PD * PD], bprior[PD], bprior_bak[PD];
    double errprior[PRD], errprior_bak[PRD], Jtinv[PRD * PRD];
    /* linearisation */
    int linearized;
    double Hpp[PD * PD], bpp[PD];       /* Hessian_ pose block (+prior), b_ pose part */
    double diagfull[PD];                /* diag(Hessian_) pose part */
    double *hll, *bl, *Hpl;             /* Hmm diagonal, bmm, Hpm column per landmark (CD entries) */
    double Hs[PD * PD], bs[PD];         /* H_pp_schur_ (before lambda), b_pp_schur_ */
    double dx_pose[PD], *dx_lm;
    double lambda, chi, ni;
    double t_hessian_ms;
    /* multi-shard exchange: the reduced visual system in camera space and the step
     * scalars are summed over all shards through the caller's hook; IMU + prior terms are replicated */
    double vis_own[VIS_N], step_own[8];
    double *vis, *step;
    double *gath_own, *step_gath_own;    /* receive side of the all-gather: [shard_count][VIS_MAXH], [shard_count][2] */
    double *gath, *step_gath;
    double Hv_dir[CD * CD];              /* un-Schur'd visual part (local shard only), for vioo_get_pose_hessian */
    vio_exchange_fn hook;
    void *hook_user;
    int nonfinite;               /* a trial chi2 was not finite during the last vio_solve (reported as VIO_ERR_NOT_FINITE) */
    double *mo_H, *mo_jt, mo_b[PRD], mo_err[PRD];      /* vio_marginalize_begin's result, kept for vio_marginalize_end */
    int mo_pending;
    vio_status mo_status;
};

static int cam_to_full(int c) { return c < 6 ? c : 6 + 15 * ((c - 6) / 6) + (c - 6) % 6; }

static double now_ms(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void vio_default_config(vio_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->ext_fixed = 1;                         /* estimate_extrinsic: 0, vio_simulation.yaml:25 */
    cfg->loss_type = VIO_LOSS_CAUCHY;
    cfg->loss_delta = 1.0;
    cfg->reproj_sqrt_info = 460.0 / 1.5;        /* estimator.cpp:42, parameters.cpp:70 */
    cfg->gravity[0] = 0; cfg->gravity[1] = 0; cfg->gravity[2] = 9.81;   /* g_norm, vio_simulation.yaml:79 */
    cfg->stream = NULL;
    cfg->shard_rank = 0; cfg->shard_count = 1;
}

vio_status vio_create(const vio_config *cfg, struct vioo_ctx **out) {
    if (!cfg || !out) return VIO_ERR_BAD_ARG;
    struct vioo_ctx *c = (struct vioo_ctx *)calloc(1, sizeof(*c));
    if (!c) return VIO_ERR_BAD_ARG;
    c->cfg = *cfg;
    if (c->cfg.shard_count < 1) c->cfg.shard_count = 1;
    c->ni = 2; c->lambda = -1;
    c->lm_dim = 1;
    c->M_mapped = -1;
    c->vis = c->vis_own; c->step = c->step_own;
    c->gath = c->gath_own = (double *)calloc((size_t)c->cfg.shard_count * VIS_MAXH, sizeof(double));
    c->step_gath = c->step_gath_own = (double *)calloc((size_t)c->cfg.shard_count * 2, sizeof(double));
    for (int i = 0; i < NF; ++i) c->pose[7 * i + 6] = 1.0;
    c->ext[6] = 1.0;
    *out = c;
    return VIO_OK;
}

vio_status vio_set_config(struct vioo_ctx *c, const vio_config *cfg) {
    if (!c || !cfg) return VIO_ERR_BAD_ARG;
    const int old_count = c->cfg.shard_count;
    c->cfg = *cfg;
    if (c->cfg.shard_count < 1) c->cfg.shard_count = 1;
    if (c->cfg.shard_count != old_count) {
        const int own_g = c->gath == c->gath_own, own_s = c->step_gath == c->step_gath_own;
        free(c->gath_own); free(c->step_gath_own);
        c->gath_own = (double *)calloc((size_t)c->cfg.shard_count * VIS_MAXH, sizeof(double));
        c->step_gath_own = (double *)calloc((size_t)c->cfg.shard_count * 2, sizeof(double));
        if (own_g) c->gath = c->gath_own;
        if (own_s) c->step_gath = c->step_gath_own;
    }
    c->linearized = 0;
    return VIO_OK;
}

void vio_destroy(struct vioo_ctx *c) {
    if (!c) return;
    free(c->invd); free(c->invd_bak); free(c->lm); free(c->host); free(c->target);
    free(c->pts_i); free(c->pts_j); free(c->hll); free(c->bl); free(c->Hpl); free(c->dx_lm);
    free(c->gath_own); free(c->step_gath_own); free(c->mo_H); free(c->mo_jt);
    free(c);
}

const char *vio_last_error(const struct vioo_ctx *c) { return c ? c->err : "null context"; }

vio_status vio_set_window(struct vioo_ctx *c, const double *poses, const double *sb, const double *ext) {
    if (!c || !poses || !sb || !ext) return VIO_ERR_BAD_ARG;
    memcpy(c->pose, poses, sizeof(c->pose)); memcpy(c->sb, sb, sizeof(c->sb)); memcpy(c->ext, ext, sizeof(c->ext));
    c->linearized = 0;
    return VIO_OK;
}

static vio_status set_landmarks_dim(struct vioo_ctx *c, int64_t n, const double *val, int dim) {
    if (!c || n < 0 || (n > 0 && !val)) return VIO_ERR_BAD_ARG;
    free(c->invd); free(c->invd_bak); free(c->hll); free(c->bl); free(c->Hpl); free(c->dx_lm);
    if (dim != c->lm_dim || n != c->N) {
        c->M = 0;                                       /* the observation list refers to the other kind / other indices */
        if (c->M_mapped >= 0) c->M_mapped = -2;         /* ... and so does a mapping: its commit is refused (as the HIP library does) */
    }
    c->N = n;
    c->lm_dim = dim;
    size_t nn = (size_t)(n > 0 ? n : 1);
    c->invd = (double *)malloc(sizeof(double) * nn * dim);
    c->invd_bak = (double *)malloc(sizeof(double) * nn * dim);
    c->hll = (double *)calloc(nn * dim * dim, sizeof(double));
    c->bl = (double *)calloc(nn * dim, sizeof(double));
    c->Hpl = (double *)calloc(nn * CD * dim, sizeof(double));
    c->dx_lm = (double *)calloc(nn * dim, sizeof(double));
    if (n > 0) memcpy(c->invd, val, sizeof(double) * n * dim);
    c->linearized = 0;
    return VIO_OK;
}

vio_status vio_set_landmarks(struct vioo_ctx *c, int64_t n, const double *invd) { return set_landmarks_dim(c, n, invd, 1); }
vio_status vio_set_landmarks_xyz(struct vioo_ctx *c, int64_t n, const double *xyz) { return set_landmarks_dim(c, n, xyz, 3); }

/* EdgeReprojectionXYZ x M: (landmark, observing frame, observation) */
vio_status vio_set_observations_xyz(struct vioo_ctx *c, int64_t m, const int32_t *lm, const int32_t *frame, const double *pts) {
    if (!c || m < 0 || (m > 0 && (!lm || !frame || !pts))) return VIO_ERR_BAD_ARG;
    if (c->lm_dim != 3) { snprintf(c->err, sizeof(c->err), "vio_set_observations_xyz needs vio_set_landmarks_xyz first"); return VIO_ERR_BAD_ARG; }
    c->M_mapped = -1;
    for (int64_t e = 0; e < m; ++e)
        if (lm[e] < 0 || lm[e] >= c->N || frame[e] < 0 || frame[e] >= NF) {
            snprintf(c->err, sizeof(c->err), "observation %lld out of range", (long long)e);
            return VIO_ERR_BAD_ARG;
        }
    free(c->lm); free(c->host); free(c->target); free(c->pts_i); free(c->pts_j);
    c->M = m;
    size_t mm = (size_t)(m > 0 ? m : 1);
    c->lm = (int32_t *)malloc(sizeof(int32_t) * mm); c->host = (int32_t *)calloc(mm, sizeof(int32_t));
    c->target = (int32_t *)malloc(sizeof(int32_t) * mm);
    c->pts_i = (double *)calloc(2 * mm, sizeof(double)); c->pts_j = (double *)malloc(sizeof(double) * 2 * mm);
    if (m > 0) {
        memcpy(c->lm, lm, sizeof(int32_t) * m); memcpy(c->target, frame, sizeof(int32_t) * m);
        memcpy(c->pts_j, pts, sizeof(double) * 2 * m);
    }
    c->linearized = 0;
    return VIO_OK;
}

vio_status vio_set_observations(struct vioo_ctx *c, int64_t m, const int32_t *lm, const int32_t *host,
                                const int32_t *target, const double *pi, const double *pj) {
    if (!c || m < 0 || (m > 0 && (!lm || !host || !target || !pi || !pj))) return VIO_ERR_BAD_ARG;
    if (c->lm_dim == 3) { snprintf(c->err, sizeof(c->err), "the context holds XYZ landmarks: use vio_set_observations_xyz"); return VIO_ERR_BAD_ARG; }
    if (c->M_mapped != -1) { c->M_mapped = -1; c->M = 0; }     /* vio_set_observations ends a mapping (include/vio_backend.h): what was written in place is dropped */
    for (int64_t e = 0; e < m; ++e) {
        if (lm[e] < 0 || lm[e] >= c->N || host[e] < 0 || host[e] >= NF || target[e] < 0 || target[e] >= NF ||
            host[e] == target[e]) {
            snprintf(c->err, sizeof(c->err), "observation %lld out of range", (long long)e);
            c->M = 0;                  /* a refused list leaves the context without one (include/vio_backend.h) */
            c->linearized = 0;
            return VIO_ERR_BAD_ARG;
        }
    }
    free(c->lm); free(c->host); free(c->target); free(c->pts_i); free(c->pts_j);
    c->M = m;
    size_t mm = (size_t)(m > 0 ? m : 1);
    c->lm = (int32_t *)malloc(sizeof(int32_t) * mm); c->host = (int32_t *)malloc(sizeof(int32_t) * mm);
    c->target = (int32_t *)malloc(sizeof(int32_t) * mm);
    c->pts_i = (double *)malloc(sizeof(double) * 2 * mm); c->pts_j = (double *)malloc(sizeof(double) * 2 * mm);
    if (m > 0) {
        memcpy(c->lm, lm, sizeof(int32_t) * m); memcpy(c->host, host, sizeof(int32_t) * m);
        memcpy(c->target, target, sizeof(int32_t) * m);
        memcpy(c->pts_i, pi, sizeof(double) * 2 * m); memcpy(c->pts_j, pj, sizeof(double) * 2 * m);
    }
    c->linearized = 0;
    return VIO_OK;
}

/* vio_map_observations / vio_commit_observations (include/vio_backend.h): the context's arrays written in place */
vio_status vio_map_observations(struct vioo_ctx *c, int64_t m, int32_t **lm, int32_t **host, int32_t **target, double **pi, double **pj) {
    if (!c || m < 0 || !lm || !host || !target || !pi || !pj) return VIO_ERR_BAD_ARG;
    if (c->lm_dim == 3) { snprintf(c->err, sizeof(c->err), "the context holds XYZ landmarks: use vio_set_observations_xyz"); return VIO_ERR_BAD_ARG; }
    free(c->lm); free(c->host); free(c->target); free(c->pts_i); free(c->pts_j);
    size_t mm = (size_t)(m > 0 ? m : 1);
    c->lm = (int32_t *)calloc(mm, sizeof(int32_t)); c->host = (int32_t *)calloc(mm, sizeof(int32_t));
    c->target = (int32_t *)calloc(mm, sizeof(int32_t));
    c->pts_i = (double *)calloc(2 * mm, sizeof(double)); c->pts_j = (double *)calloc(2 * mm, sizeof(double));
    c->M = 0;                      /* no list until the commit */
    c->M_mapped = m;
    c->linearized = 0;
    *lm = c->lm; *host = c->host; *target = c->target; *pi = c->pts_i; *pj = c->pts_j;
    return VIO_OK;
}
vio_status vio_commit_observations(struct vioo_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (c->M_mapped == -2) {
        c->M_mapped = -1;
        snprintf(c->err, sizeof(c->err), "vio_commit_observations: the mapping was invalidated by vio_set_landmarks (another landmark count): map again");
        return VIO_ERR_BAD_ARG;
    }
    if (c->M_mapped < 0) { snprintf(c->err, sizeof(c->err), "vio_commit_observations without vio_map_observations"); return VIO_ERR_BAD_ARG; }
    const int64_t m = c->M_mapped;
    c->M_mapped = -1;
    for (int64_t e = 0; e < m; ++e) {
        if (c->lm[e] < 0 || c->lm[e] >= c->N || c->host[e] < 0 || c->host[e] >= NF || c->target[e] < 0 || c->target[e] >= NF ||
            c->host[e] == c->target[e]) {
            snprintf(c->err, sizeof(c->err), "observation %lld out of range", (long long)e);
            return VIO_ERR_BAD_ARG;
        }
    }
    c->M = m;
    return VIO_OK;
}

vio_status vio_set_imu(struct vioo_ctx *c, int32_t k, const vio_preint *pre) {
    if (!c || k < 0 || k >= VIO_WINDOW_SIZE) return VIO_ERR_BAD_ARG;
    if (!pre) { c->imu_valid[k] = 0; return VIO_OK; }
    c->pre[k] = *pre;
    vioo_inverse15(pre->covariance, c->imu_info[k]);    /* SetInformation(covariance.inverse()), edge_imu.cc:35 */
    c->imu_valid[k] = 1;
    c->linearized = 0;
    return VIO_OK;
}

vio_status vio_set_imu_all(struct vioo_ctx *c, const vio_preint *const *pre) {
    if (!c || !pre) return VIO_ERR_BAD_ARG;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) { vio_status s = vio_set_imu(c, k, pre[k]); if (s != VIO_OK) return s; }
    return VIO_OK;
}

vio_status vio_set_prior(struct vioo_ctx *c, int32_t dim, const double *H, const double *b, const double *err,
                         const double *jt) {
    if (!c || (dim != 0 && dim != PRD)) return VIO_ERR_BAD_ARG;
    memset(c->Hprior, 0, sizeof(c->Hprior)); memset(c->bprior, 0, sizeof(c->bprior));
    memset(c->errprior, 0, sizeof(c->errprior)); memset(c->Jtinv, 0, sizeof(c->Jtinv));
    c->has_prior = 0;
    if (dim == PRD) {
        if (!H || !b || !err || !jt) return VIO_ERR_BAD_ARG;
        /* ExtendHessiansPriorSize(15): zero rows/cols appended (problem.cc:82-91) */
        for (int i = 0; i < PRD; ++i) { memcpy(&c->Hprior[i * PD], &H[i * PRD], sizeof(double) * PRD); c->bprior[i] = b[i]; }
        memcpy(c->errprior, err, sizeof(double) * PRD);
        memcpy(c->Jtinv, jt, sizeof(double) * PRD * PRD);
        c->has_prior = 1;
    }
    c->linearized = 0;
    return VIO_OK;
}

/* add J_a^T * W * J_b (6x6, J row-major 2x6) into the 72x72 camera-space block at (ia, ib), and mirror it
 * (problem.cc:347-355) */
static void add_cam_block(double *H, int ia, int ib, const double *Ja, const double *W, const double *Jb, int same) {
    for (int r = 0; r < 6; ++r) {
        double t0 = Ja[r] * W[0] + Ja[6 + r] * W[2];    /* (Ja^T W) row r */
        double t1 = Ja[r] * W[1] + Ja[6 + r] * W[3];
        for (int cc = 0; cc < 6; ++cc) {
            double h = t0 * Jb[cc] + t1 * Jb[6 + cc];
            H[(ia + r) * CD + ib + cc] += h;
            if (!same) H[(ib + cc) * CD + ia + r] += h;
        }
    }
}

/* Reprojection edges of this shard (MakeHessian's edge sweep, problem.cc:314-360, restricted to the camera
 * columns the visual factors touch) followed by the landmark Schur complement (problem.cc:412-429):
 *   vis.H = Hv - (Hpm*Hmm^-1)*Hmp,  vis.bred = bv - (Hpm*Hmm^-1)*bmm,  vis.bdir = bv,  vis.diag = diag(Hv)
 * marg_mode: Problem::Marginalize's assembly (problem.cc:641-681): no IsFixed test, only the landmarks hosted
 * in frame 0 (estimator.cpp:762-764). */
/* one reprojection edge's share of MakeHessian (problem.cc:303-389): H_ll, b_l and the Hpl row of its landmark, the
 * camera blocks into Hv, the camera part of b into bv, RobustChi2 into *chi */
static void accum_edge(struct vioo_ctx *c, int64_t e, int fixed, double *Hv, double *bv, double *chi) {
    const double s = c->cfg.reproj_sqrt_info, info = s * s;
    int l = c->lm[e], fi = c->host[e], fj = c->target[e];
    double r[2], Jl[2], Ji[12], Jj[12], Je[12], W[4], drho;
    vioo_reproj_edge(&c->pose[7 * fi], &c->pose[7 * fj], c->ext, c->invd[l], &c->pts_i[2 * e], &c->pts_j[2 * e],
                     r, Jl, Ji, Jj, Je);
    vioo_robust_info2(c->cfg.loss_type, c->cfg.loss_delta, s, r, &drho, W);
    *chi += robust_chi2_2(c->cfg.loss_type, c->cfg.loss_delta, s, r);
    int ii = 6 + 6 * fi, ij = 6 + 6 * fj;
    c->hll[l] += (Jl[0] * W[0] + Jl[1] * W[2]) * Jl[0] + (Jl[0] * W[1] + Jl[1] * W[3]) * Jl[1];
    double t0 = Jl[0] * W[0] + Jl[1] * W[2], t1 = Jl[0] * W[1] + Jl[1] * W[3];
    double *w = &c->Hpl[(size_t)l * CD];       /* Hmp row == Hpm column (W symmetric) */
    for (int k = 0; k < 6; ++k) {
        w[ii + k] += t0 * Ji[k] + t1 * Ji[6 + k];
        w[ij + k] += t0 * Jj[k] + t1 * Jj[6 + k];
        if (!fixed) w[k] += t0 * Je[k] + t1 * Je[6 + k];
    }
    /* pose-pose blocks in the edge's vertex order (landmark, pose_i, pose_j, ext), j >= i */
    add_cam_block(Hv, ii, ii, Ji, W, Ji, 1);
    add_cam_block(Hv, ii, ij, Ji, W, Jj, 0);
    if (!fixed) add_cam_block(Hv, ii, 0, Ji, W, Je, 0);
    add_cam_block(Hv, ij, ij, Jj, W, Jj, 1);
    if (!fixed) add_cam_block(Hv, ij, 0, Jj, W, Je, 0);
    if (!fixed) add_cam_block(Hv, 0, 0, Je, W, Je, 1);
    /* b -= drho * J^T * Information * r  (problem.cc:357) */
    double ir0 = info * r[0], ir1 = info * r[1];
    c->bl[l] -= drho * (Jl[0] * ir0 + Jl[1] * ir1);
    for (int k = 0; k < 6; ++k) {
        bv[ii + k] -= drho * (Ji[k] * ir0 + Ji[6 + k] * ir1);
        bv[ij + k] -= drho * (Jj[k] * ir0 + Jj[6 + k] * ir1);
        if (!fixed) bv[k] -= drho * (Je[k] * ir0 + Je[6 + k] * ir1);
    }
}

/* one landmark's Schur terms: S += Hpm Hmm^-1 Hmp, sb += Hpm Hmm^-1 b_l (problem.cc:419-429) */
static void accum_schur(struct vioo_ctx *c, int64_t l, double *S, double *sb, double *maxh, int *degenerate) {
    const double *w = &c->Hpl[(size_t)l * CD];
    *maxh = fmax(*maxh, fabs(c->hll[l]));
    double hinv = 1.0 / c->hll[l];                      /* Hmm_inv (problem.cc:419-425) */
    /* a landmark without information (every edge weighted to zero by the loss): the reference's dense tempH = Hpm *
     * Hmm_inv (problem.cc:427) multiplies zeros by that infinity, and the NaNs fill H_pp_schur; the sparse loop below
     * would skip them, so say it explicitly */
    if (c->hll[l] == 0.0) *degenerate = 1;
    int nzc[CD], nn = 0;
    for (int a = 0; a < CD; ++a) if (w[a] != 0.0) nzc[nn++] = a;
    for (int x = 0; x < nn; ++x) {
        double ta = w[nzc[x]] * hinv;                   /* tempH = Hpm * Hmm_inv (problem.cc:427) */
        for (int y = 0; y < nn; ++y) S[nzc[x] * CD + nzc[y]] += ta * w[nzc[y]];
        sb[nzc[x]] += ta * c->bl[l];
    }
}

/* XYZ landmarks: one EdgeReprojectionXYZ's share of MakeHessian.  Vertex order of the edge: (landmark, pose). */
static void accum_edge_xyz(struct vioo_ctx *c, int64_t e, double *Hv, double *bv, double *chi) {
    const double s = c->cfg.reproj_sqrt_info, info = s * s;
    const int l = c->lm[e], f = c->target[e];
    double r[2], Jf[6], Jp[12], W[4], drho;
    vioo_reproj_xyz_edge(&c->pose[7 * f], c->ext, &c->invd[3 * (size_t)l], &c->pts_j[2 * e], r, Jf, Jp);
    vioo_robust_info2(c->cfg.loss_type, c->cfg.loss_delta, s, r, &drho, W);
    *chi += robust_chi2_2(c->cfg.loss_type, c->cfg.loss_delta, s, r);
    const int ip = 6 + 6 * f;
    double *hl = &c->hll[9 * (size_t)l], *w = &c->Hpl[(size_t)l * CD * 3];
    for (int a = 0; a < 3; ++a) {
        const double t0 = Jf[a] * W[0] + Jf[3 + a] * W[2], t1 = Jf[a] * W[1] + Jf[3 + a] * W[3];     /* (Jf^T W) row a */
        for (int b2 = 0; b2 < 3; ++b2) hl[3 * a + b2] += t0 * Jf[b2] + t1 * Jf[3 + b2];
        for (int k = 0; k < 6; ++k) w[(ip + k) * 3 + a] += t0 * Jp[k] + t1 * Jp[6 + k];              /* Hpm(pose k, landmark a) */
    }
    add_cam_block(Hv, ip, ip, Jp, W, Jp, 1);
    const double ir0 = info * r[0], ir1 = info * r[1];
    for (int a = 0; a < 3; ++a) c->bl[3 * (size_t)l + a] -= drho * (Jf[a] * ir0 + Jf[3 + a] * ir1);
    for (int k = 0; k < 6; ++k) bv[ip + k] -= drho * (Jp[k] * ir0 + Jp[6 + k] * ir1);
}

/* tempH = Hpm * Hmm_inv for one landmark's three columns, then S += tempH * Hmp, sb += tempH * bmm (problem.cc:419-429).
 * `Y` (72 x 3) is returned for the back-substitution's sake only through c->hll: the caller keeps Hmm_inv there. */
static void accum_schur_xyz(struct vioo_ctx *c, int64_t l, double *S, double *sb, double *maxh, int *degenerate) {
    const double *w = &c->Hpl[(size_t)l * CD * 3], *hl = &c->hll[9 * (size_t)l], *bl = &c->bl[3 * (size_t)l];
    for (int a = 0; a < 3; ++a) *maxh = fmax(*maxh, fabs(hl[4 * a]));
    double hinv[9];
    vioo_inverse3(hl, hinv);
    for (int k = 0; k < 9; ++k) if (!isfinite(hinv[k])) *degenerate = 1;
    int nzc[CD], nn = 0;
    for (int a = 0; a < CD; ++a) if (w[3 * a] != 0.0 || w[3 * a + 1] != 0.0 || w[3 * a + 2] != 0.0) nzc[nn++] = a;
    for (int x = 0; x < nn; ++x) {
        const double *wa = &w[3 * nzc[x]];
        double ta[3];
        for (int j = 0; j < 3; ++j) ta[j] = wa[0] * hinv[j] + wa[1] * hinv[3 + j] + wa[2] * hinv[6 + j];
        for (int y = 0; y < nn; ++y) {
            const double *wb = &w[3 * nzc[y]];
            S[nzc[x] * CD + nzc[y]] += ta[0] * wb[0] + ta[1] * wb[1] + ta[2] * wb[2];
        }
        sb[nzc[x]] += ta[0] * bl[0] + ta[1] * bl[1] + ta[2] * bl[2];
    }
}

/* marg_mode: Problem::Marginalize of an XYZ graph (problem.cc:617-715): only the edges connected to the marginalised pose
 * (GetConnectedEdges(margVertexs[0]), :621) — a landmark seen from frame 0 enters with that ONE observation, whatever else
 * observes it — and only the landmarks those edges touch (:627-637).  One EdgeReprojectionXYZ leaves a 3x3 Hmm block of rank
 * 2; the reference inverts it all the same (:697-700, Eigen's PartialPivLU of the dynamic block: the third pivot is what
 * rounding left of an exact zero), and so does this: bug-compatible, and documented as such in DESIGN.md section 2. */
static void linearize_visual_xyz(struct vioo_ctx *c, int marg_mode) {
    double *Hv = c->Hv_dir;
    double bv[CD], sb[CD];
    memset(Hv, 0, sizeof(double) * CD * CD); memset(bv, 0, sizeof(bv)); memset(sb, 0, sizeof(sb));
    const size_t nn = (size_t)(c->N > 0 ? c->N : 1);
    memset(c->hll, 0, sizeof(double) * nn * 9); memset(c->bl, 0, sizeof(double) * nn * 3);
    memset(c->Hpl, 0, sizeof(double) * nn * CD * 3);
    double chi = 0, maxh = 0;
    double *S = (double *)calloc(CD * CD, sizeof(double));
    int degenerate = 0;
    unsigned char *in_graph = (unsigned char *)calloc(nn, 1);       /* the landmarks the graph's edges touch (problem.cc:627-637) */
    for (int64_t e = 0; e < c->M; ++e) {
        if (marg_mode && c->target[e] != 0) continue;
        in_graph[c->lm[e]] = 1;
        accum_edge_xyz(c, e, Hv, bv, &chi);
    }
    for (int64_t l = 0; l < c->N; ++l) if (!marg_mode || in_graph[l]) accum_schur_xyz(c, l, S, sb, &maxh, &degenerate);
    free(in_graph);
    for (int a = 0; a < CD; ++a) {
        for (int b2 = 0; b2 < CD; ++b2) c->vis[VIS_H + a * CD + b2] = Hv[a * CD + b2] - S[a * CD + b2];
        c->vis[VIS_BRED + a] = bv[a] - sb[a];
        c->vis[VIS_BDIR + a] = bv[a];
        c->vis[VIS_DIAG + a] = Hv[a * CD + a];
    }
    if (degenerate)
        for (int a = 0; a < CD; ++a) {
            for (int b2 = 0; b2 < CD; ++b2) c->vis[VIS_H + a * CD + b2] = NAN;
            c->vis[VIS_BRED + a] = NAN;
        }
    c->vis[VIS_CHI] = chi;
    c->vis[VIS_MAXH] = maxh;
    free(S);
}

static void linearize_visual(struct vioo_ctx *c, int marg_mode) {
    if (c->lm_dim == 3) { linearize_visual_xyz(c, marg_mode); return; }
    const int fixed = marg_mode ? 0 : c->cfg.ext_fixed;
    double *Hv = c->Hv_dir;
    double bv[CD];
    memset(Hv, 0, sizeof(double) * CD * CD); memset(bv, 0, sizeof(bv));
#ifndef _OPENMP
    for (int64_t l = 0; l < c->N; ++l) { c->hll[l] = 0; c->bl[l] = 0; }
    memset(c->Hpl, 0, sizeof(double) * (size_t)(c->N > 0 ? c->N : 1) * CD);
#endif
    double chi = 0, maxh = 0;
    double *S = (double *)calloc(CD * CD, sizeof(double));
    double sb[CD];
    int degenerate = 0;
    memset(sb, 0, sizeof(sb));
#ifdef _OPENMP
    /* The all-cores build (liboracle_omp.so: bench.py's second CPU figure, never the parity checker): landmarks are dealt
     * out to the threads in contiguous blocks, a landmark's edges through a CSR list, so the per-landmark sums stay with
     * one thread; the 72x72 accumulators are private and added in thread order.  Same terms, landmark-major order. */
    {
        int64_t *off = (int64_t *)calloc((size_t)c->N + 2, sizeof(int64_t));
        int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(c->M > 0 ? c->M : 1));
        for (int64_t e = 0; e < c->M; ++e) if (!(marg_mode && c->host[e] != 0)) ++off[c->lm[e] + 1];
        for (int64_t l = 0; l < c->N; ++l) off[l + 1] += off[l];
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(c->N + 1));
        memcpy(fill, off, sizeof(int64_t) * (size_t)(c->N + 1));
        for (int64_t e = 0; e < c->M; ++e) if (!(marg_mode && c->host[e] != 0)) idx[fill[c->lm[e]]++] = e;
        const int nt = oracle_threads();
        double *Hp = (double *)calloc((size_t)nt * (2 * CD * CD + 2 * CD + 4), sizeof(double));
        #pragma omp parallel num_threads(nt)
        {
            const int t = omp_get_thread_num();
            double *Hv_t = Hp + (size_t)t * (2 * CD * CD + 2 * CD + 4), *S_t = Hv_t + CD * CD, *bv_t = S_t + CD * CD, *sb_t = bv_t + CD;
            double *sc = sb_t + CD;      /* chi, maxh, degenerate */
            int deg = 0;
            const int64_t l0 = c->N * t / nt, l1 = c->N * (t + 1) / nt;
            for (int64_t l = l0; l < l1; ++l) { c->hll[l] = 0; c->bl[l] = 0; }
            if (l1 > l0) memset(c->Hpl + (size_t)l0 * CD, 0, sizeof(double) * (size_t)(l1 - l0) * CD);
            for (int64_t l = l0; l < l1; ++l) {
                for (int64_t q = off[l]; q < off[l + 1]; ++q) accum_edge(c, idx[q], fixed, Hv_t, bv_t, &sc[0]);
                if (!marg_mode || off[l + 1] > off[l]) accum_schur(c, l, S_t, sb_t, &sc[1], &deg);     /* (a landmark no marginalisation edge touches is not in that graph) */
            }
            sc[2] = deg;
        }
        for (int t = 0; t < nt; ++t) {
            const double *Hv_t = Hp + (size_t)t * (2 * CD * CD + 2 * CD + 4), *S_t = Hv_t + CD * CD, *bv_t = S_t + CD * CD, *sb_t = bv_t + CD;
            const double *sc = sb_t + CD;
            for (int i = 0; i < CD * CD; ++i) { Hv[i] += Hv_t[i]; S[i] += S_t[i]; }
            for (int i = 0; i < CD; ++i) { bv[i] += bv_t[i]; sb[i] += sb_t[i]; }
            chi += sc[0]; maxh = fmax(maxh, sc[1]); if (sc[2] != 0.0) degenerate = 1;
        }
        free(Hp); free(fill); free(idx); free(off);
    }
#else
    /* the marginalisation graph holds the landmarks its edges touch (problem.cc:627-637) — also one whose every edge the loss
     * weights to zero: its Hmm block is 0 and its "inverse" poisons the dense products, in the reference as here */
    unsigned char *in_graph = (unsigned char *)calloc((size_t)(c->N > 0 ? c->N : 1), 1);
    for (int64_t e = 0; e < c->M; ++e) {
        if (marg_mode && c->host[e] != 0) continue;
        in_graph[c->lm[e]] = 1;
        accum_edge(c, e, fixed, Hv, bv, &chi);
    }
    /* Schur terms, landmarks in index order */
    for (int64_t l = 0; l < c->N; ++l) if (!marg_mode || in_graph[l]) accum_schur(c, l, S, sb, &maxh, &degenerate);
    free(in_graph);
#endif
    for (int a = 0; a < CD; ++a) {
        for (int b2 = 0; b2 < CD; ++b2) c->vis[VIS_H + a * CD + b2] = Hv[a * CD + b2] - S[a * CD + b2];
        c->vis[VIS_BRED + a] = bv[a] - sb[a];
        c->vis[VIS_BDIR + a] = bv[a];
        c->vis[VIS_DIAG + a] = Hv[a * CD + a];
    }
    if (degenerate)
        for (int a = 0; a < CD; ++a) {
            for (int b2 = 0; b2 < CD; ++b2) c->vis[VIS_H + a * CD + b2] = NAN;
            c->vis[VIS_BRED + a] = NAN;
        }
    c->vis[VIS_CHI] = chi;
    c->vis[VIS_MAXH] = maxh;
    free(S);
}

/* IMU edges (replicated on every shard): vertex order (pose_i, sb_i, pose_j, sb_j) = 30 contiguous columns from
 * 6+15*i.  Adds J^T Info J into H (upper vertex blocks computed, lower mirrored: problem.cc:347-355) and
 * -J^T Info r into b. */
static void add_imu_terms(struct vioo_ctx *c, int marg_mode, double *H, double *b1, double *b2, double *diag) {
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) {
        if (!c->imu_valid[k]) continue;
        if (marg_mode && k != 0) continue;  /* only pre_integrations[1] (estimator.cpp:735-747) */
        double r[15], Jpi[90], Jsi[135], Jpj[90], Jsj[135], J[15 * 30];
        vioo_imu_edge(&c->pre[k], c->cfg.gravity, &c->pose[7 * k], &c->sb[9 * k], &c->pose[7 * (k + 1)],
                      &c->sb[9 * (k + 1)], r, Jpi, Jsi, Jpj, Jsj);
        for (int i = 0; i < 15; ++i) {
            for (int j = 0; j < 6; ++j) { J[30 * i + j] = Jpi[6 * i + j]; J[30 * i + 15 + j] = Jpj[6 * i + j]; }
            for (int j = 0; j < 9; ++j) { J[30 * i + 6 + j] = Jsi[9 * i + j]; J[30 * i + 21 + j] = Jsj[9 * i + j]; }
        }
        const double *I = c->imu_info[k];
        double JtI[30 * 15], T[30 * 30], Ir[15];
        for (int a = 0; a < 30; ++a)
            for (int j = 0; j < 15; ++j) {
                double sum = 0;
                for (int i = 0; i < 15; ++i) sum += J[30 * i + a] * I[15 * i + j];
                JtI[15 * a + j] = sum;
            }
        for (int a = 0; a < 30; ++a)
            for (int b = 0; b < 30; ++b) {
                double sum = 0;
                for (int j = 0; j < 15; ++j) sum += JtI[15 * a + j] * J[30 * j + b];
                T[30 * a + b] = sum;
            }
        static const int bs[5] = {0, 6, 15, 21, 30};
        int base = 6 + 15 * k;
        for (int bi = 0; bi < 4; ++bi)
            for (int bj = bi; bj < 4; ++bj)
                for (int a = bs[bi]; a < bs[bi + 1]; ++a)
                    for (int b = bs[bj]; b < bs[bj + 1]; ++b) {
                        H[(base + a) * PD + base + b] += T[30 * a + b];
                        if (bi != bj) H[(base + b) * PD + base + a] += T[30 * a + b];
                        if (diag && a == b) diag[base + a] += T[30 * a + b];
                    }
        for (int i = 0; i < 15; ++i) { double sum = 0; for (int j = 0; j < 15; ++j) sum += I[15 * i + j] * r[j]; Ir[i] = sum; }
        for (int a = 0; a < 30; ++a) {
            double sum = 0;
            for (int i = 0; i < 15; ++i) sum += J[30 * i + a] * Ir[i];
            b1[base + a] -= 1.0 * sum;
            if (b2) b2[base + a] -= 1.0 * sum;
        }
    }
}

/* The exchange (include/vio_backend.h): which 0 / 1 = the caller all-gathers every shard's vis[0 .. VIS_MAXH) resp. step[0 .. 2)
 * into the rank-major receive buffers; the sum over the shards is then formed here, in rank order — the same additions in the
 * same order on every rank.  which 2: max of step[2] over the shards, reduced in place by the caller. */
static int run_hook(struct vioo_ctx *c, int which) {
    if (!c->hook) return 0;
    if (c->hook(c->hook_user, which) != 0) return 1;
    const int R = c->cfg.shard_count;
    if (which == 0)
        for (int i = 0; i < VIS_MAXH; ++i) {
            double s = c->gath[i];
            for (int r = 1; r < R; ++r) s += c->gath[(size_t)r * VIS_MAXH + i];
            c->vis[i] = s;
        }
    else if (which == 1)
        for (int i = 0; i < 2; ++i) {
            double s = c->step_gath[i];
            for (int r = 1; r < R; ++r) s += c->step_gath[2 * r + i];
            c->step[i] = s;
        }
    return 0;
}

/* SetOrdering + MakeHessian (problem.cc:256-285,303-389) + the lambda-free part of SolveLinearSystem (:412-429) */
vio_status vio_prepare(struct vioo_ctx *c) { return c ? VIO_OK : VIO_ERR_BAD_ARG; }      /* (the port builds nothing ahead of its solve) */

vio_status vio_linearize(struct vioo_ctx *c) {
    if (c && c->M_mapped != -1) { snprintf(c->err, sizeof(c->err), "vio_map_observations without vio_commit_observations"); return VIO_ERR_BAD_ARG; }
    if (!c) return VIO_ERR_BAD_ARG;
    double t0 = now_ms();
    linearize_visual(c, 0);
    if (run_hook(c, 0) != 0) return VIO_ERR_HIP;        /* sum of vis[0 .. VIS_MAXH) over the shards */
    /* H_pp_schur_ (no lambda) = reduced visual + IMU + prior; b_pp_schur_; pose part of b_; diag(Hessian_) */
    memset(c->Hs, 0, sizeof(c->Hs)); memset(c->bs, 0, sizeof(c->bs)); memset(c->bpp, 0, sizeof(c->bpp));
    double diag[PD];
    memset(diag, 0, sizeof(diag));
    for (int a = 0; a < CD; ++a) {
        int fa = cam_to_full(a);
        for (int b2 = 0; b2 < CD; ++b2) c->Hs[fa * PD + cam_to_full(b2)] = c->vis[VIS_H + a * CD + b2];
        c->bs[fa] = c->vis[VIS_BRED + a];
        c->bpp[fa] = c->vis[VIS_BDIR + a];
        diag[fa] = c->vis[VIS_DIAG + a];
    }
    double *R = (double *)calloc(PD * PD, sizeof(double));      /* IMU + prior part */
    double rb[PD];
    memset(rb, 0, sizeof(rb));
    add_imu_terms(c, 0, R, rb, NULL, NULL);
    c->t_hessian_ms += now_ms() - t0;
    /* prior, with the rows/cols of fixed pose vertices zeroed (problem.cc:365-384) */
    for (int i = 0; i < PD; ++i) {
        int fi = c->cfg.ext_fixed && i < 6;
        for (int j = 0; j < PD; ++j) {
            int fj = c->cfg.ext_fixed && j < 6;
            R[i * PD + j] += (fi || fj) ? 0.0 : c->Hprior[i * PD + j];
        }
        rb[i] += fi ? 0.0 : c->bprior[i];
    }
    for (int i = 0; i < PD; ++i) {
        for (int j = 0; j < PD; ++j) c->Hs[i * PD + j] += R[i * PD + j];
        c->bs[i] += rb[i];
        c->bpp[i] += rb[i];
        diag[i] += R[i * PD + i];
    }
    /* Hpp is only its diagonal plus the local shard's un-Schur'd visual blocks; see vioo_get_pose_hessian */
    memcpy(c->Hpp, R, sizeof(c->Hpp));
    for (int a = 0; a < CD; ++a)
        for (int b2 = 0; b2 < CD; ++b2) c->Hpp[cam_to_full(a) * PD + cam_to_full(b2)] += c->Hv_dir[a * CD + b2];
    memcpy(c->diagfull, diag, sizeof(diag));
    free(R);
    memset(c->dx_pose, 0, sizeof(c->dx_pose));
    for (int64_t l = 0; l < c->N * c->lm_dim; ++l) c->dx_lm[l] = 0;
    c->linearized = 1;
    return VIO_OK;
}

/* sum of RobustChi2 over this shard's reprojection edges */
static double chi2_visual(struct vioo_ctx *c) {
    const double s = c->cfg.reproj_sqrt_info;
    double chi = 0;
    if (c->lm_dim == 3) {
        for (int64_t e = 0; e < c->M; ++e) {
            double r[2];
            vioo_reproj_xyz_edge(&c->pose[7 * c->target[e]], c->ext, &c->invd[3 * (size_t)c->lm[e]], &c->pts_j[2 * e], r, NULL, NULL);
            chi += robust_chi2_2(c->cfg.loss_type, c->cfg.loss_delta, s, r);
        }
        return chi;
    }
#ifdef _OPENMP
    #pragma omp parallel for reduction(+ : chi) schedule(static) num_threads(oracle_threads())
#endif
    for (int64_t e = 0; e < c->M; ++e) {
        double r[2];
        vioo_reproj_edge(&c->pose[7 * c->host[e]], &c->pose[7 * c->target[e]], c->ext, c->invd[c->lm[e]],
                         &c->pts_i[2 * e], &c->pts_j[2 * e], r, NULL, NULL, NULL, NULL);
        chi += robust_chi2_2(c->cfg.loss_type, c->cfg.loss_delta, s, r);
    }
    return chi;
}

/* IMU + prior part of chi2 (replicated) */
static double chi2_imu_prior(struct vioo_ctx *c) {
    double chi = 0;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) {
        if (!c->imu_valid[k]) continue;
        double r[15];
        vioo_imu_edge(&c->pre[k], c->cfg.gravity, &c->pose[7 * k], &c->sb[9 * k], &c->pose[7 * (k + 1)],
                      &c->sb[9 * (k + 1)], r, NULL, NULL, NULL, NULL);
        const double *I = c->imu_info[k];
        double e2 = 0;
        for (int i = 0; i < 15; ++i) { double t = 0; for (int j = 0; j < 15; ++j) t += I[15 * i + j] * r[j]; e2 += r[i] * t; }
        chi += e2;
    }
    if (c->has_prior) {
        double n2 = 0;
        for (int i = 0; i < PRD; ++i) n2 += c->errprior[i] * c->errprior[i];
        chi += sqrt(n2);            /* err_prior_.norm(), NOT squared (problem.cc:505-507,554-556) */
    }
    return chi;
}

/* 0.5*(sum RobustChi2 + ||err_prior||) at the current states; scale_lm rides along in the second slot */
static vio_status chi2_exchange(struct vioo_ctx *c, double scale_lm, double *chi_out, double *scale_out) {
    c->step[0] = chi2_visual(c);
    c->step[1] = scale_lm;
    if (run_hook(c, 1) != 0) return VIO_ERR_HIP;
    *chi_out = 0.5 * (c->step[0] + chi2_imu_prior(c));
    if (scale_out) *scale_out = c->step[1];
    return VIO_OK;
}

vio_status vio_chi2(struct vioo_ctx *c, double *chi2) {
    if (!c || !chi2) return VIO_ERR_BAD_ARG;
    return chi2_exchange(c, 0.0, chi2, NULL);
}

/* ComputeLambdaInitLM, problem.cc:497-522 */
vio_status vio_init_lm(struct vioo_ctx *c, double *chi2, double *lambda) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    c->ni = 2.;
    /* chi2 at the linearisation point: the visual sum travelled with the reduced system */
    c->chi = 0.5 * (c->vis[VIS_CHI] + chi2_imu_prior(c));
    c->step[2] = c->vis[VIS_MAXH];
    if (run_hook(c, 2) != 0) return VIO_ERR_HIP;        /* max of step[2] over the shards */
    double maxd = c->step[2];
    for (int i = 0; i < PD; ++i) maxd = fmax(fabs(c->diagfull[i]), maxd);
    maxd = fmin(5e10, maxd);
    c->lambda = 1e-5 * maxd;
    if (chi2) *chi2 = c->chi;
    if (lambda) *lambda = c->lambda;
    return VIO_OK;
}

/* SolveLinearSystem (SLAM branch), problem.cc:406-449 */
vio_status vio_solve_linear(struct vioo_ctx *c, double lambda) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    double *H = (double *)malloc(sizeof(double) * PD * PD);
    memcpy(H, c->Hs, sizeof(double) * PD * PD);
    for (int i = 0; i < PD; ++i) H[i * PD + i] += lambda;
    vioo_ldlt_solve(PD, H, c->bs, c->dx_pose, NULL);
    free(H);
    if (c->lm_dim == 3) {       /* delta_x_ll = Hmm_inv * (bmm - Hmp * delta_x_pp), problem.cc:445 */
        for (int64_t l = 0; l < c->N; ++l) {
            const double *w = &c->Hpl[(size_t)l * CD * 3];
            double t[3] = {0, 0, 0}, hinv[9], v[3];
            for (int a = 0; a < CD; ++a) {
                if (w[3 * a] == 0.0 && w[3 * a + 1] == 0.0 && w[3 * a + 2] == 0.0) continue;
                const double d = c->dx_pose[cam_to_full(a)];
                for (int j = 0; j < 3; ++j) t[j] += w[3 * a + j] * d;
            }
            vioo_inverse3(&c->hll[9 * (size_t)l], hinv);
            for (int j = 0; j < 3; ++j) v[j] = c->bl[3 * (size_t)l + j] - t[j];
            for (int i = 0; i < 3; ++i) c->dx_lm[3 * (size_t)l + i] = hinv[3 * i] * v[0] + hinv[3 * i + 1] * v[1] + hinv[3 * i + 2] * v[2];
        }
        c->lambda = lambda;
        return VIO_OK;
    }
#ifdef _OPENMP
    #pragma omp parallel for schedule(static) num_threads(oracle_threads())
#endif
    for (int64_t l = 0; l < c->N; ++l) {
        const double *w = &c->Hpl[(size_t)l * CD];
        double t = 0;
        for (int a = 0; a < CD; ++a) if (w[a] != 0.0) t += w[a] * c->dx_pose[cam_to_full(a)];
        c->dx_lm[l] = (1.0 / c->hll[l]) * (c->bl[l] - t);
    }
    c->lambda = lambda;
    return VIO_OK;
}

/* UpdateStates, problem.cc:453-480 */
vio_status vio_update_states(struct vioo_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    memcpy(c->pose_bak, c->pose, sizeof(c->pose)); memcpy(c->sb_bak, c->sb, sizeof(c->sb));
    memcpy(c->ext_bak, c->ext, sizeof(c->ext));
    if (c->N > 0) memcpy(c->invd_bak, c->invd, sizeof(double) * c->N * c->lm_dim);
    vioo_pose_plus(c->ext, &c->dx_pose[0]);
    for (int i = 0; i < NF; ++i) {
        vioo_pose_plus(&c->pose[7 * i], &c->dx_pose[6 + 15 * i]);
        for (int k = 0; k < 9; ++k) c->sb[9 * i + k] += c->dx_pose[12 + 15 * i + k];
    }
    for (int64_t l = 0; l < c->N * c->lm_dim; ++l) c->invd[l] += c->dx_lm[l];       /* Vertex::Plus (vertex.cc:28-30) */
    if (c->has_prior) {
        memcpy(c->bprior_bak, c->bprior, sizeof(c->bprior)); memcpy(c->errprior_bak, c->errprior, sizeof(c->errprior));
        double tmp[PD];
        for (int i = 0; i < PD; ++i) { double s = 0; for (int j = 0; j < PD; ++j) s += c->Hprior[i * PD + j] * c->dx_pose[j]; tmp[i] = s; }
        for (int i = 0; i < PD; ++i) c->bprior[i] -= tmp[i];
        for (int i = 0; i < PRD; ++i) {
            double s = 0;
            for (int j = 0; j < PRD; ++j) s += -c->Jtinv[i * PRD + j] * c->bprior[j];
            c->errprior[i] = s;
        }
    }
    return VIO_OK;
}

/* RollbackStates, problem.cc:482-494 */
vio_status vio_rollback_states(struct vioo_ctx *c) {
    if (!c) return VIO_ERR_BAD_ARG;
    memcpy(c->pose, c->pose_bak, sizeof(c->pose)); memcpy(c->sb, c->sb_bak, sizeof(c->sb));
    memcpy(c->ext, c->ext_bak, sizeof(c->ext));
    if (c->N > 0) memcpy(c->invd, c->invd_bak, sizeof(double) * c->N * c->lm_dim);
    if (c->has_prior) { memcpy(c->bprior, c->bprior_bak, sizeof(c->bprior)); memcpy(c->errprior, c->errprior_bak, sizeof(c->errprior)); }
    return VIO_OK;
}

/* IsGoodStepInLM, problem.cc:541-573 */
vio_status vio_eval_step(struct vioo_ctx *c, int32_t *accepted, double *chi2, double *lambda) {
    if (!c) return VIO_ERR_BAD_ARG;
    double scale_lm = 0, scale = 0;
    for (int64_t l = 0; l < c->N * c->lm_dim; ++l) scale_lm += c->dx_lm[l] * (c->lambda * c->dx_lm[l] + c->bl[l]);
    double tempChi;
    vio_status st = chi2_exchange(c, scale_lm, &tempChi, &scale_lm);
    if (st != VIO_OK) return st;
    for (int i = 0; i < PD; ++i) scale += c->dx_pose[i] * (c->lambda * c->dx_pose[i] + c->bpp[i]);
    scale = 0.5 * (scale + scale_lm);
    scale += 1e-6;
    double rho = (c->chi - tempChi) / scale;
    int ok;
    if (!isfinite(tempChi)) c->nonfinite = 1;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        double scaleFactor = fmax(1. / 3., alpha);
        c->lambda *= scaleFactor;
        c->ni = 2;
        c->chi = tempChi;
        ok = 1;
    } else {
        c->lambda *= c->ni;
        c->ni *= 2;
        ok = 0;
    }
    if (accepted) *accepted = ok;
    if (chi2) *chi2 = c->chi;
    if (lambda) *lambda = c->lambda;
    return VIO_OK;
}

/* Problem::Solve, problem.cc:169-250 */
vio_status vio_solve(struct vioo_ctx *c, int32_t iterations, vio_solve_report *rep) {
    if (c && c->M_mapped != -1) { snprintf(c->err, sizeof(c->err), "vio_map_observations without vio_commit_observations"); return VIO_ERR_BAD_ARG; }
    if (!c) return VIO_ERR_BAD_ARG;
    if (c->M == 0 && c->N == 0) {
        int any = 0;
        for (int k = 0; k < VIO_WINDOW_SIZE; ++k) any |= c->imu_valid[k];
        if (!any) return VIO_ERR_EMPTY;
    }
    double t0 = now_ms();
    c->t_hessian_ms = 0;
    vio_solve_report r;
    memset(&r, 0, sizeof(r));
    c->nonfinite = 0;
    vio_linearize(c);
    vio_init_lm(c, &r.initial_chi2, NULL);
    if (!isfinite(r.initial_chi2)) c->nonfinite = 1;
    int stop = 0, iter = 0;
    double last_chi = 1e20;
    while (!stop && iter < iterations) {
        if (iter < 128) { r.chi2_trace[iter] = c->chi; r.lambda_trace[iter] = c->lambda; }
        int success = 0, false_cnt = 0;
        while (!success && false_cnt < 10) {
            vio_solve_linear(c, c->lambda);
            vio_update_states(c);
            int32_t ok;
            vio_eval_step(c, &ok, NULL, NULL);
            r.trials++;
            if (ok) { vio_linearize(c); false_cnt = 0; success = 1; r.accepted++; }
            else { false_cnt++; vio_rollback_states(c); }
        }
        iter++;
        if (last_chi - c->chi < 1e-5) { stop = 1; r.stop_reason = 1; }
        last_chi = c->chi;
    }
    r.iterations = iter;
    r.final_chi2 = c->chi; r.final_lambda = c->lambda;
    r.solve_ms = now_ms() - t0; r.hessian_ms = c->t_hessian_ms;
    if (rep) *rep = r;
    /* the reference's Solve returns true here: non-finite trials were rejected (problem.cc:559) and the states are the
     * last accepted ones; the ABI says so in its status (include/vio_backend.h) */
    return c->nonfinite ? VIO_ERR_NOT_FINITE : VIO_OK;
}

vio_status vio_gn_iteration(struct vioo_ctx *c, double lambda) {
    if (c && c->M_mapped != -1) { snprintf(c->err, sizeof(c->err), "vio_map_observations without vio_commit_observations"); return VIO_ERR_BAD_ARG; }
    if (!c) return VIO_ERR_BAD_ARG;
    vio_linearize(c);
    vio_solve_linear(c, lambda);
    vio_update_states(c);
    return chi2_exchange(c, 0.0, &c->chi, NULL);
}

vio_status vio_synchronize(struct vioo_ctx *c) { return c ? VIO_OK : VIO_ERR_BAD_ARG; }

/* ------------------------------------------------------------------------------------------ */
/* Marginalize: estimator.cpp:693-901 + problem.cc:617-795                                     */
/* ------------------------------------------------------------------------------------------ */
static void move_to_bottom(double *H, double *b, int n, int idx, int dim) {
    /* rows idx..idx+dim -> bottom, then cols (problem.cc:728-744) */
    double *T = (double *)malloc(sizeof(double) * n * n);
    int rest = n - idx - dim;
    memcpy(T, H, sizeof(double) * n * n);
    for (int i = 0; i < rest; ++i) memcpy(&H[(idx + i) * n], &T[(idx + dim + i) * n], sizeof(double) * n);
    for (int i = 0; i < dim; ++i) memcpy(&H[(n - dim + i) * n], &T[(idx + i) * n], sizeof(double) * n);
    memcpy(T, H, sizeof(double) * n * n);
    for (int r = 0; r < n; ++r) {
        for (int j = 0; j < rest; ++j) H[r * n + idx + j] = T[r * n + idx + dim + j];
        for (int j = 0; j < dim; ++j) H[r * n + n - dim + j] = T[r * n + idx + j];
    }
    double tb[PD];
    memcpy(tb, b, sizeof(double) * n);
    for (int i = 0; i < rest; ++i) b[idx + i] = tb[idx + dim + i];
    for (int i = 0; i < dim; ++i) b[n - dim + i] = tb[idx + i];
    free(T);
}

/* Schur complement of the trailing m2 x m2 block through an eigen-decomposition pseudo-inverse with the 1e-8
 * cut (problem.cc:747-764; the same lines in A/15-vio-backend's TestMarginalize): H is n x n row-major,
 * Hp (n-m2)x(n-m2), bp n-m2.  b may be NULL. */
void vioo_schur_pinv(int n, int m2, const double *H, const double *b, double *Hp, double *bp) {
    const int n2 = n - m2;
    const double eps = 1e-8;
    double *Amm = (double *)calloc((size_t)m2 * m2, sizeof(double)), *ev = (double *)malloc(sizeof(double) * m2);
    double *V = (double *)malloc(sizeof(double) * m2 * m2), *Ainv = (double *)malloc(sizeof(double) * m2 * m2);
    double *tempB = (double *)malloc(sizeof(double) * (n2 > 0 ? n2 : 1) * m2);
    for (int i = 0; i < m2; ++i) for (int j = 0; j < m2; ++j) Amm[i * m2 + j] = 0.5 * (H[(n2 + i) * n + n2 + j] + H[(n2 + j) * n + n2 + i]);
    vioo_symmetric_eigen(m2, Amm, ev, V);
    for (int i = 0; i < m2; ++i) for (int j = 0; j < m2; ++j) {
        double s = 0;
        for (int k = 0; k < m2; ++k) s += V[i * m2 + k] * (ev[k] > eps ? 1.0 / ev[k] : 0.0) * V[j * m2 + k];
        Ainv[i * m2 + j] = s;
    }
    for (int i = 0; i < n2; ++i) for (int j = 0; j < m2; ++j) {
        double s = 0;
        for (int k = 0; k < m2; ++k) s += H[i * n + n2 + k] * Ainv[k * m2 + j];
        tempB[i * m2 + j] = s;
    }
    for (int i = 0; i < n2; ++i) {
        for (int j = 0; j < n2; ++j) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += tempB[i * m2 + k] * H[(n2 + k) * n + j];
            Hp[i * n2 + j] = H[i * n + j] - s;
        }
        if (b && bp) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += tempB[i * m2 + k] * b[n2 + k];
            bp[i] = b[i] - s;
        }
    }
    free(Amm); free(ev); free(V); free(Ainv); free(tempB);
}

vio_status vio_marginalize(struct vioo_ctx *c, int32_t kind, double *Hout, double *bout, double *errout, double *jtout) {
    if (c && c->M_mapped != -1) { snprintf(c->err, sizeof(c->err), "vio_map_observations without vio_commit_observations"); return VIO_ERR_BAD_ARG; }
    if (!c || !Hout || !bout || !errout || !jtout) return VIO_ERR_BAD_ARG;
    if (kind != VIO_MARG_OLD && kind != VIO_MARG_SECOND_NEW) return VIO_ERR_BAD_ARG;
    const int n = PD;
    double *H = (double *)calloc(n * n, sizeof(double));
    double b[PD];
    memset(b, 0, sizeof(b));
    if (kind == VIO_MARG_OLD) {
        /* edges connected to frame 0 (problem.cc:621-681), landmarks Schur-ed out (:685-708) */
        linearize_visual(c, 1);
        if (run_hook(c, 0) != 0) { free(H); return VIO_ERR_HIP; }      /* shards: sum the partial Schur systems */
        for (int a = 0; a < CD; ++a) {
            int fa = cam_to_full(a);
            for (int b2 = 0; b2 < CD; ++b2) H[fa * n + cam_to_full(b2)] = c->vis[VIS_H + a * CD + b2];
            b[fa] = c->vis[VIS_BRED + a];
        }
        add_imu_terms(c, 1, H, b, NULL, NULL);
        c->linearized = 0;      /* hll/bl/Hpl now hold the marginalisation graph's values */
    }
    /* += prior (no fixed-vertex zeroing here, problem.cc:710-715) */
    for (int i = 0; i < n * n; ++i) H[i] += c->Hprior[i];
    for (int i = 0; i < n; ++i) b[i] += c->bprior[i];
    /* move speed-bias then pose of the marginalised frame to the bottom (index-large first, :721-745) */
    int f = (kind == VIO_MARG_OLD) ? 0 : VIO_WINDOW_SIZE - 1;
    move_to_bottom(H, b, n, 12 + 15 * f, 9);
    move_to_bottom(H, b, n, 6 + 15 * f, 6);
    const int m2 = 15, n2 = n - 15;
    double *Hp = (double *)malloc(sizeof(double) * n2 * n2);
    double bp[PRD];
    const double eps = 1e-8;
    vioo_schur_pinv(n, m2, H, b, Hp, bp);
    double *ev2 = (double *)malloc(sizeof(double) * n2);
    double *V2 = (double *)malloc(sizeof(double) * n2 * n2);
    vioo_symmetric_eigen(n2, Hp, ev2, V2);
    /* Jt_prior_inv = S_inv_sqrt.asDiagonal() * V^T ; err = -Jt_prior_inv * b ; H = J^T J with J = S_sqrt * V^T */
    for (int i = 0; i < n2; ++i) {
        double sinv = ev2[i] > eps ? sqrt(1.0 / ev2[i]) : 0.0;
        for (int j = 0; j < n2; ++j) jtout[i * n2 + j] = sinv * V2[j * n2 + i];
    }
    for (int i = 0; i < n2; ++i) {
        double s = 0;
        for (int j = 0; j < n2; ++j) s += -jtout[i * n2 + j] * bp[j];
        errout[i] = s;
    }
    for (int i = 0; i < n2; ++i) for (int j = 0; j < n2; ++j) {
        double s = 0;
        for (int k = 0; k < n2; ++k) {
            double sk = ev2[k] > eps ? ev2[k] : 0.0;
            s += V2[i * n2 + k] * sk * V2[j * n2 + k];      /* (sqrt(S) V^T)^T (sqrt(S) V^T) */
        }
        Hout[i * n2 + j] = fabs(s) > 1e-9 ? s : 0.0;
    }
    memcpy(bout, bp, sizeof(double) * n2);
    free(H); free(Hp); free(ev2); free(V2);
    /* a landmark block without an inverse: the reference's Marginalize returns true with H_prior_ = 0 and b_prior_, err_prior_,
     * Jt_prior_inv_ all NaN (the arithmetic above has just produced that); the status says so */
    for (int i = 0; i < n2; ++i)
        if (!isfinite(bout[i])) {
            snprintf(c->err, sizeof(c->err), "vio_marginalize: a landmark block has no inverse; the prior is the reference's outcome for that case");
            return VIO_ERR_NOT_FINITE;
        }
    return VIO_OK;
}

/* Test infrastructure for the product's host code (tests/test_host_units.py): the dense 171 x 171 system Problem::Marginalize holds after
 * the landmark Schur complement and the old prior (problem.cc:685-715) and BEFORE the marginalised frame moves to the bottom — what the
 * HIP library's device half hands to its host tail (csrc/host_dense.cpp: marginalize_tail). */
vio_status vioo_marg_dense_input(struct vioo_ctx *c, int32_t kind, double *H171, double *b171) {
    if (!c || !H171 || !b171 || c->M_mapped != -1) return VIO_ERR_BAD_ARG;
    const int n = PD;
    memset(H171, 0, sizeof(double) * n * n);
    memset(b171, 0, sizeof(double) * n);
    if (kind == VIO_MARG_OLD) {
        linearize_visual(c, 1);
        for (int a = 0; a < CD; ++a) {
            int fa = cam_to_full(a);
            for (int b2 = 0; b2 < CD; ++b2) H171[fa * n + cam_to_full(b2)] = c->vis[VIS_H + a * CD + b2];
            b171[fa] = c->vis[VIS_BRED + a];
        }
        add_imu_terms(c, 1, H171, b171, NULL, NULL);
        c->linearized = 0;
    }
    for (int i = 0; i < n * n; ++i) H171[i] += c->Hprior[i];
    for (int i = 0; i < n; ++i) b171[i] += c->bprior[i];
    return VIO_OK;
}

/* the two halves of include/vio_backend.h (here: begin computes, end copies; nothing runs in the background) */
vio_status vio_marginalize_begin(struct vioo_ctx *c, int32_t kind) {
    if (c && c->M_mapped != -1) { snprintf(c->err, sizeof(c->err), "vio_map_observations without vio_commit_observations"); return VIO_ERR_BAD_ARG; }
    if (!c) return VIO_ERR_BAD_ARG;
    if (!c->mo_H) { c->mo_H = (double *)malloc(sizeof(double) * PRD * PRD); c->mo_jt = (double *)malloc(sizeof(double) * PRD * PRD); }
    c->mo_status = vio_marginalize(c, kind, c->mo_H, c->mo_b, c->mo_err, c->mo_jt);
    c->mo_pending = (c->mo_status == VIO_OK || c->mo_status == VIO_ERR_NOT_FINITE);
    return c->mo_pending ? VIO_OK : c->mo_status;
}
vio_status vio_marginalize_end(struct vioo_ctx *c, double *H, double *b, double *err, double *jt) {
    if (!c || !H || !b || !err || !jt || !c->mo_pending) return VIO_ERR_BAD_ARG;
    c->mo_pending = 0;
    memcpy(H, c->mo_H, sizeof(double) * PRD * PRD); memcpy(jt, c->mo_jt, sizeof(double) * PRD * PRD);
    memcpy(b, c->mo_b, sizeof(c->mo_b)); memcpy(err, c->mo_err, sizeof(c->mo_err));
    return c->mo_status;
}

/* ------------------------------------------------------------------------------------------ */
/* getters                                                                                     */
/* ------------------------------------------------------------------------------------------ */
vio_status vio_get_window(struct vioo_ctx *c, double *poses, double *sb, double *ext) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (poses) memcpy(poses, c->pose, sizeof(c->pose));
    if (sb) memcpy(sb, c->sb, sizeof(c->sb));
    if (ext) memcpy(ext, c->ext, sizeof(c->ext));
    return VIO_OK;
}
vio_status vio_get_landmarks(struct vioo_ctx *c, int64_t n, double *invd) {
    if (!c || n != c->N || c->lm_dim != 1 || (n > 0 && !invd)) return VIO_ERR_BAD_ARG;
    if (n > 0) memcpy(invd, c->invd, sizeof(double) * n);
    return VIO_OK;
}
vio_status vio_get_landmarks_xyz(struct vioo_ctx *c, int64_t n, double *xyz) {
    if (!c || n != c->N || c->lm_dim != 3 || (n > 0 && !xyz)) return VIO_ERR_BAD_ARG;
    if (n > 0) memcpy(xyz, c->invd, sizeof(double) * 3 * n);
    return VIO_OK;
}
vio_status vio_get_prior(struct vioo_ctx *c, double *b, double *err) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (b) memcpy(b, c->bprior, sizeof(c->bprior));
    if (err) memcpy(err, c->errprior, sizeof(c->errprior));
    return VIO_OK;
}
vio_status vio_get_delta(struct vioo_ctx *c, double *dxp, int64_t n, double *dxl) {
    if (!c || (dxl && n != c->N)) return VIO_ERR_BAD_ARG;
    if (dxp) memcpy(dxp, c->dx_pose, sizeof(c->dx_pose));
    if (dxl && n > 0) memcpy(dxl, c->dx_lm, sizeof(double) * n * c->lm_dim);
    return VIO_OK;
}
vio_status vio_get_schur_system(struct vioo_ctx *c, double *H, double *b) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    if (H) memcpy(H, c->Hs, sizeof(c->Hs));
    if (b) memcpy(b, c->bs, sizeof(c->bs));
    return VIO_OK;
}
vio_status vio_get_landmark_system(struct vioo_ctx *c, int64_t n, double *hll, double *bl) {
    if (!c || n != c->N) return VIO_ERR_BAD_ARG;
    if (hll && n > 0) memcpy(hll, c->hll, sizeof(double) * n * c->lm_dim * c->lm_dim);
    if (bl && n > 0) memcpy(bl, c->bl, sizeof(double) * n * c->lm_dim);
    return VIO_OK;
}
vio_status vio_get_pose_gradient(struct vioo_ctx *c, double *b, double *diag) {
    if (!c || !c->linearized) return VIO_ERR_BAD_ARG;
    if (b) memcpy(b, c->bpp, sizeof(c->bpp));
    if (diag) memcpy(diag, c->diagfull, sizeof(c->diagfull));
    return VIO_OK;
}
vio_status vioo_get_pose_hessian(struct vioo_ctx *c, double *Hpp) {
    if (!c || !c->linearized || !Hpp) return VIO_ERR_BAD_ARG;
    memcpy(Hpp, c->Hpp, sizeof(c->Hpp));
    return VIO_OK;
}
vio_status vio_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int32_t count,
                            const double *dt, const double *acc, const double *gyr, double acc_n, double gyr_n,
                            double acc_w, double gyr_w, vio_preint *out) {
    if (!acc0 || !gyr0 || !ba || !bg || !out || count < 0) return VIO_ERR_BAD_ARG;
    vioo_preintegrate(acc0, gyr0, ba, bg, count, dt, acc, gyr, acc_n, gyr_n, acc_w, gyr_w, out);
    return VIO_OK;
}
vio_status vio_exchange_buffers(struct vioo_ctx *c, void **a, int64_t *na, void **b, int64_t *nb) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (a) *a = c->vis;
    if (na) *na = VIS_MAXH;
    if (b) *b = c->step;
    if (nb) *nb = 2;
    return VIO_OK;
}
vio_status vio_set_exchange_hook(struct vioo_ctx *c, vio_exchange_fn fn, void *user) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->hook = fn; c->hook_user = user;
    return VIO_OK;
}
vio_status vio_gather_buffers(struct vioo_ctx *c, void **gs, void **gc) {
    if (!c) return VIO_ERR_BAD_ARG;
    if (gs) *gs = c->gath;
    if (gc) *gc = c->step_gath;
    return VIO_OK;
}
vio_status vio_bind_gather_buffers(struct vioo_ctx *c, void *gs, void *gc) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->gath = gs ? (double *)gs : c->gath_own;
    c->step_gath = gc ? (double *)gc : c->step_gath_own;
    c->linearized = 0;
    return VIO_OK;
}
vio_status vio_bind_exchange_buffers(struct vioo_ctx *c, void *reduced, void *scalars) {
    if (!c) return VIO_ERR_BAD_ARG;
    c->vis = reduced ? (double *)reduced : c->vis_own;
    c->step = scalars ? (double *)scalars : c->step_own;
    c->linearized = 0;
    return VIO_OK;
}
