// estimator_backend.cpp — see estimator_backend.h.
#include "estimator_backend.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cmath>
#include <cstring>

namespace vio {

// ---- the small rotation helpers vector2double / double2vector are made of; row-major 3x3 ----
namespace {

const double kPi = 3.14159265358979323846;

// Utility::R2ypr (VM/include/utility/utility.h:68-84), degrees
void R2ypr(const double R[9], double ypr[3]) {
    const double n[3] = {R[0], R[3], R[6]}, o[3] = {R[1], R[4], R[7]}, a[3] = {R[2], R[5], R[8]};
    const double y = std::atan2(n[1], n[0]);
    const double p = std::atan2(-n[2], n[0] * std::cos(y) + n[1] * std::sin(y));
    const double r = std::atan2(a[0] * std::sin(y) - a[1] * std::cos(y), -o[0] * std::sin(y) + o[1] * std::cos(y));
    ypr[0] = y / kPi * 180.0; ypr[1] = p / kPi * 180.0; ypr[2] = r / kPi * 180.0;
}

void matmul3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

void matvec3(const double A[9], const double v[3], double out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}

// Utility::ypr2R (utility.h:86-110): Rz * Ry * Rx, degrees
void ypr2R(const double ypr[3], double R[9]) {
    const double y = ypr[0] / 180.0 * kPi, p = ypr[1] / 180.0 * kPi, r = ypr[2] / 180.0 * kPi;
    const double Rz[9] = {std::cos(y), -std::sin(y), 0, std::sin(y), std::cos(y), 0, 0, 0, 1};
    const double Ry[9] = {std::cos(p), 0, std::sin(p), 0, 1, 0, -std::sin(p), 0, std::cos(p)};
    const double Rx[9] = {1, 0, 0, 0, std::cos(r), -std::sin(r), 0, std::sin(r), std::cos(r)};
    double T[9];
    matmul3(Rz, Ry, T);
    matmul3(T, Rx, R);
}

// Eigen::Quaterniond(w, x, y, z)[.normalized()].toRotationMatrix(); q as para_Pose stores it (x, y, z, w)
void quat2R(const double q[4], bool normalize, double R[9]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    if (normalize) {
        const double n = std::sqrt(x * x + y * y + z * z + w * w);
        x /= n; y /= n; z /= n; w /= n;
    }
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen::Quaterniond q{R}: the branch on the trace, then on the largest diagonal element
void R2quat(const double R[9], double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
        return;
    }
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
}

bool usable(FeaturePerId &f) {          // feature_manager.cpp:146-148 and every other loop over the tracks
    f.used_num = (int)f.feature_per_frame.size();
    return f.used_num >= 2 && f.start_frame < WINDOW_SIZE - 2;
}

}  // namespace

EstimatorBackend::EstimatorBackend(const vio_config &cfg) {
    std::memset(para_Pose, 0, sizeof(para_Pose));
    std::memset(para_SpeedBias, 0, sizeof(para_SpeedBias));
    std::memset(para_Ex_Pose, 0, sizeof(para_Ex_Pose));
    for (auto &p : pre_integrations) p = nullptr;
    std::memset(&last_report, 0, sizeof(last_report));
    std::memset(Ps, 0, sizeof(Ps)); std::memset(Vs, 0, sizeof(Vs)); std::memset(Bas, 0, sizeof(Bas)); std::memset(Bgs, 0, sizeof(Bgs));
    std::memset(tic, 0, sizeof(tic)); std::memset(last_P0, 0, sizeof(last_P0));
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (auto &R : Rs) std::memcpy(R, I3, sizeof(I3));
    std::memcpy(ric[0], I3, sizeof(I3)); std::memcpy(last_R0, I3, sizeof(I3));
    vio_status st = vio_create(&cfg, &ctx_);
    if (st != VIO_OK) { ctx_ = nullptr; err_ = "vio_create failed with status " + std::to_string((int)st); }
}

EstimatorBackend::~EstimatorBackend() { if (ctx_) vio_destroy(ctx_); }

const char *EstimatorBackend::last_error() const { return (ctx_ && err_.empty()) ? vio_last_error(ctx_) : err_.c_str(); }

void EstimatorBackend::vector2double() {                                      // estimator.cpp:505-547
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        for (int k = 0; k < 3; ++k) {
            para_Pose[i][k] = Ps[i][k];
            para_SpeedBias[i][k] = Vs[i][k]; para_SpeedBias[i][3 + k] = Bas[i][k]; para_SpeedBias[i][6 + k] = Bgs[i][k];
        }
        R2quat(Rs[i], &para_Pose[i][3]);
    }
    for (int k = 0; k < 3; ++k) para_Ex_Pose[0][k] = tic[0][k];
    R2quat(ric[0], &para_Ex_Pose[0][3]);
    para_Feature.clear();                                                     // f_manager.getDepthVector(), feature_manager.cpp:184-200
    for (auto &f : feature) if (usable(f)) para_Feature.push_back(1.0 / f.estimated_depth);
}

void EstimatorBackend::double2vector() {                                      // estimator.cpp:549-617
    double origin_R0[3], origin_R00[3], origin_P0[3], R00[9], rot_diff[9];
    R2ypr(Rs[0], origin_R0);
    std::memcpy(origin_P0, Ps[0], sizeof(origin_P0));
    if (failure_occur) {                                                      // :554-559
        R2ypr(last_R0, origin_R0);
        std::memcpy(origin_P0, last_P0, sizeof(origin_P0));
        failure_occur = false;
    }
    quat2R(&para_Pose[0][3], false, R00);
    R2ypr(R00, origin_R00);
    const double ypr[3] = {origin_R0[0] - origin_R00[0], 0.0, 0.0};
    ypr2R(ypr, rot_diff);
    if (std::fabs(std::fabs(origin_R0[1]) - 90) < 1.0 || std::fabs(std::fabs(origin_R00[1]) - 90) < 1.0) {   // euler singular point, :567-575
        const double R00t[9] = {R00[0], R00[3], R00[6], R00[1], R00[4], R00[7], R00[2], R00[5], R00[8]};
        double Rs0[9];
        std::memcpy(Rs0, Rs[0], sizeof(Rs0));
        matmul3(Rs0, R00t, rot_diff);
    }
    for (int i = 0; i <= WINDOW_SIZE; ++i) {                                  // :577-597
        double Rq[9];
        quat2R(&para_Pose[i][3], true, Rq);
        matmul3(rot_diff, Rq, Rs[i]);
        const double d[3] = {para_Pose[i][0] - para_Pose[0][0], para_Pose[i][1] - para_Pose[0][1], para_Pose[i][2] - para_Pose[0][2]};
        double rd[3];
        matvec3(rot_diff, d, rd);
        for (int k = 0; k < 3; ++k) Ps[i][k] = rd[k] + origin_P0[k];
        matvec3(rot_diff, &para_SpeedBias[i][0], Vs[i]);
        for (int k = 0; k < 3; ++k) { Bas[i][k] = para_SpeedBias[i][3 + k]; Bgs[i][k] = para_SpeedBias[i][6 + k]; }
    }
    for (int k = 0; k < 3; ++k) tic[0][k] = para_Ex_Pose[0][k];              // :599-609
    quat2R(&para_Ex_Pose[0][3], false, ric[0]);
    size_t k = 0;                                                             // f_manager.setDepth(dep), :611-614 -> feature_manager.cpp:141-160
    for (auto &f : feature) {
        if (!usable(f) || k >= para_Feature.size()) continue;
        f.estimated_depth = 1.0 / para_Feature[k++];
        f.solve_flag = f.estimated_depth < 0 ? 2 : 1;
    }
}

// What the three graph-building blocks of estimator.cpp do (:909-1034, :699-810, :834-885), once.
bool EstimatorBackend::uploadWindow(bool graph_unchanged) {
    if (!ctx_) return false;
    if (vio_set_window(ctx_, &para_Pose[0][0], &para_SpeedBias[0][0], &para_Ex_Pose[0][0]) != VIO_OK) return false;
    if (graph_unchanged) {
        // MargOldFrame / MargNewFrame straight after problemSolve (estimator.cpp:1083-1114): the same tracks, the landmarks the
        // solve returned, the same pre-integrations — the context holds them; what changed is the states (double2vector's gauge)
        // and the prior's vectors
        // (the depths went through setDepth / getDepthVector, 1 / (1 / x): sent again, the library keeps what is bitwise the same)
        if (vio_set_landmarks(ctx_, (int64_t)para_Feature.size(), para_Feature.data()) != VIO_OK) return false;
        if (!waitMarginalization()) return false;
        if (!Hprior_.empty()) {
            if (vio_set_prior(ctx_, VIO_PRIOR_DIM, Hprior_.data(), bprior_.data(), errprior_.data(), Jprior_inv_.data()) != VIO_OK) return false;
        } else if (vio_set_prior(ctx_, 0, nullptr, nullptr, nullptr, nullptr) != VIO_OK) return false;
        return true;
    }
    // the edges of the selected tracks, written straight into the library's own (pinned) arrays: vio_map_observations, one copy of the
    // 44 bytes per edge instead of two (round 4).  The count first: the loop of estimator.cpp:975-1016 without its body
    size_t n_obs = 0;
    int n_sel = 0;
    for (auto &it_per_id : feature) {
        it_per_id.used_num = (int)it_per_id.feature_per_frame.size();
        if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
        ++n_sel;
        n_obs += it_per_id.feature_per_frame.size() - 1;
    }
    if ((int)para_Feature.size() != n_sel) {                                  // para_Feature is vector2double's (:541-543), one per selected track
        err_ = "para_Feature does not match the feature list: call vector2double() first";
        return false;
    }
    if (vio_set_landmarks(ctx_, (int64_t)para_Feature.size(), para_Feature.data()) != VIO_OK) return false;
    int32_t *lm = nullptr, *host = nullptr, *target = nullptr;
    double *pi = nullptr, *pj = nullptr;
    if (vio_map_observations(ctx_, (int64_t)n_obs, &lm, &host, &target, &pi, &pj) != VIO_OK) return false;
    int feature_index = -1;
    size_t e = 0;
    for (auto &it_per_id : feature) {                                         // estimator.cpp:975-1016
        if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
        ++feature_index;
        const int imu_i = it_per_id.start_frame;
        int imu_j = imu_i - 1;
        const auto &pts_i = it_per_id.feature_per_frame[0];
        for (const auto &it_per_frame : it_per_id.feature_per_frame) {
            imu_j++;
            if (imu_i == imu_j) continue;
            lm[e] = feature_index; host[e] = imu_i; target[e] = imu_j;
            pi[2 * e] = pts_i[0]; pi[2 * e + 1] = pts_i[1];
            pj[2 * e] = it_per_frame[0]; pj[2 * e + 1] = it_per_frame[1];
            ++e;
        }
    }
    if (vio_commit_observations(ctx_) != VIO_OK) return false;
    {                                                                         // estimator.cpp:956-970: the ten edges, one call
        const vio_preint *edges[WINDOW_SIZE];
        for (int i = 0; i < WINDOW_SIZE; ++i) {
            const vio_preint *p = pre_integrations[i + 1];
            edges[i] = (p && p->sum_dt <= 10.0) ? p : nullptr;
        }
        if (vio_set_imu_all(ctx_, edges) != VIO_OK) return false;
    }
    // (the prior of the marginalisation the frame before left running — async_marginalization — is needed from here on: its dense
    // tail has had slideWindow, the front-end and the uploads above to finish under, and the planner's pass over the new graph too)
    if (async_marginalization && vio_prepare(ctx_) != VIO_OK) return false;
    if (!waitMarginalization()) return false;
    if (!Hprior_.empty()) {                                                   // estimator.cpp:1023-1034
        if (vio_set_prior(ctx_, VIO_PRIOR_DIM, Hprior_.data(), bprior_.data(), errprior_.data(), Jprior_inv_.data()) != VIO_OK)
            return false;
    } else if (vio_set_prior(ctx_, 0, nullptr, nullptr, nullptr, nullptr) != VIO_OK) return false;
    return true;
}

bool EstimatorBackend::problemSolve() {
    static const bool timing = std::getenv("VIO_HOST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!uploadWindow()) return false;
    const auto t1 = std::chrono::steady_clock::now();
    if (vio_solve(ctx_, 10, &last_report) != VIO_OK) return false;           // problem.Solve(10), estimator.cpp:1037
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vio host timing] problemSolve: uploadWindow %.0f us, vio_solve %.0f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(),
                     std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    if (!Hprior_.empty()) {                                                   // estimator.cpp:1040-1049
        std::vector<double> b(VIO_POSE_DIM), e(VIO_PRIOR_DIM);
        if (vio_get_prior(ctx_, b.data(), e.data()) != VIO_OK) return false;
        bprior_.assign(b.begin(), b.begin() + VIO_PRIOR_DIM);
        errprior_ = e;
    }
    if (vio_get_window(ctx_, &para_Pose[0][0], &para_SpeedBias[0][0], nullptr) != VIO_OK) return false;   // :1051-1065
    if (vio_get_landmarks(ctx_, (int64_t)para_Feature.size(), para_Feature.data()) != VIO_OK) return false;   // :1067-1072
    return true;
}

// Problem::Marginalize + the four getters (estimator.cpp:821-828, :893-900).  With async_marginalization the device part runs here
// and the dense host tail on the library's helper thread; Hprior_ / bprior_ / errprior_ / Jprior_inv_ are filled by
// waitMarginalization(), which uploadWindow() calls where the next window needs its prior.
bool EstimatorBackend::marginalize(int kind) {
    if (vio_marginalize_begin(ctx_, kind) != VIO_OK) return false;
    marg_pending_ = true;
    return async_marginalization ? true : waitMarginalization();
}

bool EstimatorBackend::waitMarginalization() {
    if (!marg_pending_) return true;
    marg_pending_ = false;
    Hprior_.assign((size_t)VIO_PRIOR_DIM * VIO_PRIOR_DIM, 0.0); Jprior_inv_.assign((size_t)VIO_PRIOR_DIM * VIO_PRIOR_DIM, 0.0);
    bprior_.assign(VIO_PRIOR_DIM, 0.0); errprior_.assign(VIO_PRIOR_DIM, 0.0);
    return vio_marginalize_end(ctx_, Hprior_.data(), bprior_.data(), errprior_.data(), Jprior_inv_.data()) == VIO_OK;
}

bool EstimatorBackend::MargOldFrame() {
    const bool same = graph_uploaded_;
    graph_uploaded_ = false;
    if (!uploadWindow(same)) return false;
    return marginalize(VIO_MARG_OLD);
}

bool EstimatorBackend::MargNewFrame() {
    const bool same = graph_uploaded_;
    graph_uploaded_ = false;
    if (!uploadWindow(same)) return false;
    return marginalize(VIO_MARG_SECOND_NEW);
}

void EstimatorBackend::backendOptimization(MarginalizationFlag marginalization_flag) {
    vector2double();                                    // estimator.cpp:1079-1083
    graph_uploaded_ = false;
    if (!problemSolve()) return;
    double2vector();
    // (vector2double re-reads the depths double2vector has just written: para_Feature is what the context already holds)
    if (marginalization_flag == MARGIN_OLD) {           // :1088-1092
        vector2double();
        graph_uploaded_ = true;
        MargOldFrame();
    } else if (!Hprior_.empty()) {                      // :1107-1114 (problemSolve has collected a marginalisation that was still running)
        vector2double();
        graph_uploaded_ = true;
        MargNewFrame();
    }
    graph_uploaded_ = false;
}

}  // namespace vio
