// estimator_backend.cpp — see estimator_backend.h.
#include "estimator_backend.h"

#include <cstring>

namespace vio {

EstimatorBackend::EstimatorBackend(const vio_config &cfg) {
    std::memset(para_Pose, 0, sizeof(para_Pose));
    std::memset(para_SpeedBias, 0, sizeof(para_SpeedBias));
    std::memset(para_Ex_Pose, 0, sizeof(para_Ex_Pose));
    for (auto &p : pre_integrations) p = nullptr;
    std::memset(&last_report, 0, sizeof(last_report));
    vio_status st = vio_create(&cfg, &ctx_);
    if (st != VIO_OK) { ctx_ = nullptr; err_ = "vio_create failed with status " + std::to_string((int)st); }
}

EstimatorBackend::~EstimatorBackend() { if (ctx_) vio_destroy(ctx_); }

const char *EstimatorBackend::last_error() const { return ctx_ ? vio_last_error(ctx_) : err_.c_str(); }

// What the three graph-building blocks of estimator.cpp do (:909-1034, :699-810, :834-885), once.
bool EstimatorBackend::uploadWindow() {
    if (!ctx_) return false;
    if (vio_set_window(ctx_, &para_Pose[0][0], &para_SpeedBias[0][0], &para_Ex_Pose[0][0]) != VIO_OK) return false;
    std::vector<int32_t> lm, host, target;
    std::vector<double> pi, pj;
    int feature_index = -1;
    for (auto &it_per_id : feature) {                                         // estimator.cpp:975-1016
        it_per_id.used_num = (int)it_per_id.feature_per_frame.size();
        if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
        ++feature_index;
        const int imu_i = it_per_id.start_frame;
        int imu_j = imu_i - 1;
        const auto &pts_i = it_per_id.feature_per_frame[0];
        for (const auto &it_per_frame : it_per_id.feature_per_frame) {
            imu_j++;
            if (imu_i == imu_j) continue;
            lm.push_back(feature_index); host.push_back(imu_i); target.push_back(imu_j);
            pi.push_back(pts_i[0]); pi.push_back(pts_i[1]);
            pj.push_back(it_per_frame[0]); pj.push_back(it_per_frame[1]);
        }
    }
    para_Feature.resize(feature_index + 1);
    {
        int k = 0;                                                            // vector2double, estimator.cpp:541-543
        for (const auto &f : feature)
            if (f.used_num >= 2 && f.start_frame < WINDOW_SIZE - 2) para_Feature[k++] = f.inv_depth;
    }
    if (vio_set_landmarks(ctx_, (int64_t)para_Feature.size(), para_Feature.data()) != VIO_OK) return false;
    if (vio_set_observations(ctx_, (int64_t)lm.size(), lm.data(), host.data(), target.data(), pi.data(), pj.data()) != VIO_OK)
        return false;
    for (int i = 0; i < WINDOW_SIZE; ++i) {                                   // estimator.cpp:956-970
        const vio_preint *p = pre_integrations[i + 1];
        if (vio_set_imu(ctx_, i, (p && p->sum_dt <= 10.0) ? p : nullptr) != VIO_OK) return false;
    }
    if (!Hprior_.empty()) {                                                   // estimator.cpp:1023-1034
        if (vio_set_prior(ctx_, VIO_PRIOR_DIM, Hprior_.data(), bprior_.data(), errprior_.data(), Jprior_inv_.data()) != VIO_OK)
            return false;
    } else if (vio_set_prior(ctx_, 0, nullptr, nullptr, nullptr, nullptr) != VIO_OK) return false;
    return true;
}

bool EstimatorBackend::problemSolve() {
    if (!uploadWindow()) return false;
    if (vio_solve(ctx_, 10, &last_report) != VIO_OK) return false;           // problem.Solve(10), estimator.cpp:1037
    if (!Hprior_.empty()) {                                                   // estimator.cpp:1040-1049
        std::vector<double> b(VIO_POSE_DIM), e(VIO_PRIOR_DIM);
        if (vio_get_prior(ctx_, b.data(), e.data()) != VIO_OK) return false;
        bprior_.assign(b.begin(), b.begin() + VIO_PRIOR_DIM);
        errprior_ = e;
    }
    if (vio_get_window(ctx_, &para_Pose[0][0], &para_SpeedBias[0][0], nullptr) != VIO_OK) return false;   // :1051-1065
    if (vio_get_landmarks(ctx_, (int64_t)para_Feature.size(), para_Feature.data()) != VIO_OK) return false;   // :1067-1072
    int k = 0;                                                                // double2vector -> f_manager.setDepth
    for (auto &f : feature)
        if (f.used_num >= 2 && f.start_frame < WINDOW_SIZE - 2) f.inv_depth = para_Feature[k++];
    return true;
}

static bool marginalize(vio_ctx *ctx, int kind, std::vector<double> &H, std::vector<double> &b, std::vector<double> &e,
                        std::vector<double> &J) {
    H.assign((size_t)VIO_PRIOR_DIM * VIO_PRIOR_DIM, 0.0); J.assign((size_t)VIO_PRIOR_DIM * VIO_PRIOR_DIM, 0.0);
    b.assign(VIO_PRIOR_DIM, 0.0); e.assign(VIO_PRIOR_DIM, 0.0);
    return vio_marginalize(ctx, kind, H.data(), b.data(), e.data(), J.data()) == VIO_OK;
}

bool EstimatorBackend::MargOldFrame() {
    if (!uploadWindow()) return false;
    return marginalize(ctx_, VIO_MARG_OLD, Hprior_, bprior_, errprior_, Jprior_inv_);      // estimator.cpp:821-828
}

bool EstimatorBackend::MargNewFrame() {
    if (!uploadWindow()) return false;
    return marginalize(ctx_, VIO_MARG_SECOND_NEW, Hprior_, bprior_, errprior_, Jprior_inv_);   // estimator.cpp:893-900
}

void EstimatorBackend::backendOptimization(MarginalizationFlag marginalization_flag) {
    problemSolve();                                     // vector2double / double2vector stay with the caller
    if (marginalization_flag == MARGIN_OLD) MargOldFrame();
    else if (!Hprior_.empty()) MargNewFrame();          // estimator.cpp:1107-1114
}

}  // namespace vio
