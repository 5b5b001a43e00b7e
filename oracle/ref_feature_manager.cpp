/*
 * ref_feature_manager.cpp — drives the REFERENCE's own FeatureManager (VM/src/feature_manager.cpp, compiled where it
 * lies by oracle/Makefile into oracle/_ref/) and the header-only helpers of VM/include/utility/utility.h through a
 * plain C interface, prefix `vior_fm_` / `vior_`.
 *
 * TEST INFRASTRUCTURE ONLY: tests/golden/make_golden_feature_manager.py generates tests/golden/feature_manager.npz and
 * gauge.npz from it, and tests/test_oracle_vs_reference.py compares the oracle with it directly where /root/reference
 * exists.  No reference source text lives in this file: it only *calls* FeatureManager's methods, Utility::R2ypr /
 * ypr2R and Eigen's quaternion conversions the way Estimator::vector2double / double2vector / slideWindow do
 * (VM/src/estimator.cpp:505-643,1143-1241).
 *
 * The two configuration globals the translation unit reads (INIT_DEPTH, MIN_PARALLAX: `extern` in parameters.h:40-41,
 * set by readParameters() from the YAML file, parameters.cpp:126,103) are defined here as the caller would: INIT_DEPTH
 * = 5.0 is the constant of parameters.cpp:126, MIN_PARALLAX = 10 / 460 the value vio_simulation.yaml gives after the
 * division by FOCAL_LENGTH; both can be set through vior_fm_config.
 */
#include <cstring>
#include <list>
#include <map>
#include <vector>

#include <eigen3/Eigen/Dense>

#include "feature_manager.h"

double INIT_DEPTH = 5.0;
double MIN_PARALLAX = 10.0 / 460.0;

namespace {

struct FmHandle {
    Eigen::Matrix3d Rs[WINDOW_SIZE + 1];
    FeatureManager fm;
    FmHandle() : fm(Rs) {
        for (auto &R : Rs) R.setIdentity();
    }
};

typedef Eigen::Matrix<double, 3, 3, Eigen::RowMajor> RowMat3;

}  // namespace

extern "C" {

void vior_fm_config(double init_depth, double min_parallax) {
    INIT_DEPTH = init_depth;
    MIN_PARALLAX = min_parallax;
}

void *vior_fm_create(void) { return new FmHandle(); }
void vior_fm_destroy(void *h) { delete static_cast<FmHandle *>(h); }

/* a track as it sits in f_manager.feature: id, start frame, its points (normalised x, y; z = 1), depth, solve flag */
void vior_fm_add_track(void *h, int feature_id, int start_frame, int n_obs, const double *pts_xy, double estimated_depth,
                       int solve_flag) {
    FmHandle *f = static_cast<FmHandle *>(h);
    FeaturePerId t(feature_id, start_frame);
    for (int k = 0; k < n_obs; ++k) {
        Eigen::Matrix<double, 7, 1> p;
        p << pts_xy[2 * k], pts_xy[2 * k + 1], 1.0, 0.0, 0.0, 0.0, 0.0;
        t.feature_per_frame.push_back(FeaturePerFrame(p, 0.0));
    }
    t.estimated_depth = estimated_depth;
    t.solve_flag = solve_flag;
    f->fm.feature.push_back(t);
}

/* the front-end's hand-over (System.cpp:408-427 -> Estimator::processImage -> addFeatureCheckParallax): one image =
 * n points (id, x, y); returns the keyframe decision */
int vior_fm_add_feature_check_parallax(void *h, int frame_count, int n, const int *ids, const double *pts_xy) {
    FmHandle *f = static_cast<FmHandle *>(h);
    std::map<int, std::vector<std::pair<int, Eigen::Matrix<double, 7, 1>>>> image;
    for (int k = 0; k < n; ++k) {
        Eigen::Matrix<double, 7, 1> p;
        p << pts_xy[2 * k], pts_xy[2 * k + 1], 1.0, 0.0, 0.0, 0.0, 0.0;
        image[ids[k]].emplace_back(0, p);
    }
    return f->fm.addFeatureCheckParallax(frame_count, image, 0.0) ? 1 : 0;
}

int vior_fm_last_track_num(void *h) { return static_cast<FmHandle *>(h)->fm.last_track_num; }
int vior_fm_feature_count(void *h) { return static_cast<FmHandle *>(h)->fm.getFeatureCount(); }

int vior_fm_depth_vector(void *h, double *out) {
    Eigen::VectorXd d = static_cast<FmHandle *>(h)->fm.getDepthVector();
    for (int i = 0; i < d.size(); ++i) out[i] = d(i);
    return (int)d.size();
}

void vior_fm_set_depth(void *h, int n, const double *x) {
    Eigen::VectorXd v = Eigen::Map<const Eigen::VectorXd>(x, n);
    static_cast<FmHandle *>(h)->fm.setDepth(v);
}

void vior_fm_clear_depth(void *h, int n, const double *x) {
    Eigen::VectorXd v = Eigen::Map<const Eigen::VectorXd>(x, n);
    static_cast<FmHandle *>(h)->fm.clearDepth(v);
}

void vior_fm_remove_failures(void *h) { static_cast<FmHandle *>(h)->fm.removeFailures(); }

/* poses = para_Pose rows (p, q xyzw) of the 11 frames, ext = para_Ex_Pose: turned into Ps / Rs / tic / ric the way
 * double2vector does (Quaterniond(w,x,y,z).toRotationMatrix(), estimator.cpp:576,598-602) */
void vior_fm_triangulate(void *h, const double *poses, const double *ext) {
    FmHandle *f = static_cast<FmHandle *>(h);
    Eigen::Vector3d Ps[WINDOW_SIZE + 1], tic[1];
    Eigen::Matrix3d ric[1];
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        const double *p = poses + 7 * i;
        Ps[i] = Eigen::Vector3d(p[0], p[1], p[2]);
        f->Rs[i] = Eigen::Quaterniond(p[6], p[3], p[4], p[5]).toRotationMatrix();
    }
    tic[0] = Eigen::Vector3d(ext[0], ext[1], ext[2]);
    ric[0] = Eigen::Quaterniond(ext[6], ext[3], ext[4], ext[5]).toRotationMatrix();
    f->fm.setRic(ric);
    f->fm.triangulate(Ps, tic, ric);
}

/* 3x3 matrices row-major */
void vior_fm_remove_back_shift_depth(void *h, const double *marg_R, const double *marg_P, const double *new_R,
                                     const double *new_P) {
    Eigen::Matrix3d mR = Eigen::Map<const RowMat3>(marg_R), nR = Eigen::Map<const RowMat3>(new_R);
    Eigen::Vector3d mP(marg_P[0], marg_P[1], marg_P[2]), nP(new_P[0], new_P[1], new_P[2]);
    static_cast<FmHandle *>(h)->fm.removeBackShiftDepth(mR, mP, nR, nP);
}

void vior_fm_remove_back(void *h) { static_cast<FmHandle *>(h)->fm.removeBack(); }
void vior_fm_remove_front(void *h, int frame_count) { static_cast<FmHandle *>(h)->fm.removeFront(frame_count); }

int vior_fm_num_tracks(void *h) { return (int)static_cast<FmHandle *>(h)->fm.feature.size(); }

/* track `idx` in list order; pts_xy may be NULL (sizes only) */
int vior_fm_get_track(void *h, int idx, int *feature_id, int *start_frame, double *estimated_depth, int *solve_flag,
                      double *pts_xy, int cap) {
    FmHandle *f = static_cast<FmHandle *>(h);
    auto it = f->fm.feature.begin();
    std::advance(it, idx);
    *feature_id = it->feature_id;
    *start_frame = it->start_frame;
    *estimated_depth = it->estimated_depth;
    *solve_flag = it->solve_flag;
    const int n = (int)it->feature_per_frame.size();
    if (pts_xy)
        for (int k = 0; k < n && k < cap; ++k) {
            pts_xy[2 * k] = it->feature_per_frame[k].point.x();
            pts_xy[2 * k + 1] = it->feature_per_frame[k].point.y();
        }
    return n;
}

/* ---- the helpers vector2double / double2vector are made of (estimator.cpp:505-643) ---- */
void vior_r2ypr(const double *R, double *ypr) {
    Eigen::Matrix3d M = Eigen::Map<const RowMat3>(R);
    Eigen::Vector3d v = Utility::R2ypr(M);
    ypr[0] = v(0); ypr[1] = v(1); ypr[2] = v(2);
}

void vior_ypr2r(const double *ypr, double *R) {
    Eigen::Matrix3d M = Utility::ypr2R(Eigen::Vector3d(ypr[0], ypr[1], ypr[2]));
    Eigen::Map<RowMat3> out(R);
    out = M;
}

/* Quaterniond(w, x, y, z)[.normalized()].toRotationMatrix(), q given as para_Pose stores it (x, y, z, w) */
void vior_quat_to_rot(const double *q_xyzw, int normalize, double *R) {
    Eigen::Quaterniond q(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]);
    Eigen::Matrix3d M = normalize ? q.normalized().toRotationMatrix() : q.toRotationMatrix();
    Eigen::Map<RowMat3> out(R);
    out = M;
}

/* Quaterniond q{R} (estimator.cpp:512): Eigen's rotation-matrix -> quaternion branch selection */
void vior_rot_to_quat(const double *R, double *q_xyzw) {
    Eigen::Matrix3d M = Eigen::Map<const RowMat3>(R);
    Eigen::Quaterniond q{M};
    q_xyzw[0] = q.x(); q_xyzw[1] = q.y(); q_xyzw[2] = q.z(); q_xyzw[3] = q.w();
}

double vior_normalize_angle(double deg) { return Utility::normalizeAngle(deg); }

}  // extern "C"
