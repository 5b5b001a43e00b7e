// feature_manager.cpp — see feature_manager.h.  Each method cites the reference lines whose behaviour it keeps.
#include "feature_manager.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace vio {

bool FeatureManager::addObservation(int feature_id, int frame_count, double x, double y) {
    // :67-89: look the id up; unknown -> new FeaturePerId starting at frame_count, known -> append
    auto it = std::find_if(feature.begin(), feature.end(), [&](const FeaturePerId &f) { return f.feature_id == feature_id; });
    if (it == feature.end()) {
        FeaturePerId f;
        f.feature_id = feature_id;
        f.start_frame = frame_count;
        f.feature_per_frame.push_back({x, y});
        feature.push_back(f);
        return false;
    }
    it->feature_per_frame.push_back({x, y});
    return true;
}

bool FeatureManager::addFeatureCheckParallax(int frame_count, int n, const int *ids, const double *pts_xy) {   // :55-115
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ids[a] < ids[b]; });
    double parallax_sum = 0;
    int parallax_num = 0;
    last_track_num = 0;
    for (int k : order)
        if (addObservation(ids[k], frame_count, pts_xy[2 * k], pts_xy[2 * k + 1])) ++last_track_num;
    if (frame_count < 2 || last_track_num < 20) return true;
    for (const auto &f : feature)
        if (f.start_frame <= frame_count - 2 && f.start_frame + (int)f.feature_per_frame.size() - 1 >= frame_count - 1) {
            parallax_sum += compensatedParallax2(f, frame_count);
            ++parallax_num;
        }
    if (parallax_num == 0) return true;
    return parallax_sum / parallax_num >= MIN_PARALLAX;
}

double FeatureManager::compensatedParallax2(const FeaturePerId &f, int frame_count) {   // :352-388 (z == 1: no rotation compensation)
    const auto &pi = f.feature_per_frame[frame_count - 2 - f.start_frame];
    const auto &pj = f.feature_per_frame[frame_count - 1 - f.start_frame];
    const double du = pi[0] - pj[0], dv = pi[1] - pj[1];
    return std::max(0.0, std::sqrt(du * du + dv * dv));
}

int FeatureManager::getFeatureCount() {             // :37-52
    int cnt = 0;
    for (auto &f : feature) cnt += usable(f) ? 1 : 0;
    return cnt;
}

std::vector<double> FeatureManager::getDepthVector() {      // :184-200: inverse depths of the usable tracks, in order
    std::vector<double> dep;
    for (auto &f : feature) if (usable(f)) dep.push_back(1.0 / f.estimated_depth);
    return dep;
}

void FeatureManager::setDepth(const std::vector<double> &x) {   // :141-160
    size_t k = 0;
    for (auto &f : feature) {
        if (!usable(f)) continue;
        f.estimated_depth = 1.0 / x[k++];
        f.solve_flag = f.estimated_depth < 0 ? 2 : 1;
    }
}

void FeatureManager::clearDepth(const std::vector<double> &x) {  // :173-182
    size_t k = 0;
    for (auto &f : feature) if (usable(f)) f.estimated_depth = 1.0 / x[k++];
}

void FeatureManager::removeFailures() {             // :162-171
    feature.erase(std::remove_if(feature.begin(), feature.end(), [](const FeaturePerId &f) { return f.solve_flag == 2; }), feature.end());
}

bool FeatureManager::triangulate(vio_ctx *ctx, const double poses[][7], const double ext[7]) {    // :203-257
    // CSR of all tracks; vio_triangulate applies the rules of :207-211 itself and leaves the other depths alone
    std::vector<int32_t> start(feature.size());
    std::vector<int64_t> off(feature.size() + 1, 0);
    std::vector<double> pts, depth(feature.size());
    for (size_t i = 0; i < feature.size(); ++i) {
        FeaturePerId &f = feature[i];
        f.used_num = (int)f.feature_per_frame.size();
        start[i] = f.start_frame;
        off[i + 1] = off[i] + (int64_t)f.feature_per_frame.size();
        for (const auto &p : f.feature_per_frame) { pts.push_back(p[0]); pts.push_back(p[1]); }
        depth[i] = f.estimated_depth;
    }
    if (vio_triangulate(ctx, (int64_t)feature.size(), start.data(), off.data(), pts.data(), &poses[0][0], ext, INIT_DEPTH, depth.data()) != VIO_OK)
        return false;
    for (size_t i = 0; i < feature.size(); ++i) feature[i].estimated_depth = depth[i];
    return true;
}

void FeatureManager::removeBackShiftDepth(const double marg_R[9], const double marg_P[3], const double new_R[9], const double new_P[3]) {   // :276-312
    std::vector<FeaturePerId> kept;
    kept.reserve(feature.size());
    for (auto &f : feature) {
        if (f.start_frame != 0) { --f.start_frame; kept.push_back(f); continue; }
        const double uv[3] = {f.feature_per_frame[0][0], f.feature_per_frame[0][1], 1.0};
        f.feature_per_frame.erase(f.feature_per_frame.begin());
        if (f.feature_per_frame.size() < 2) continue;                       // the track dies with its host frame
        double pi[3], w[3], pj[3];
        for (int k = 0; k < 3; ++k) pi[k] = uv[k] * f.estimated_depth;
        for (int r = 0; r < 3; ++r) w[r] = marg_R[3 * r] * pi[0] + marg_R[3 * r + 1] * pi[1] + marg_R[3 * r + 2] * pi[2] + marg_P[r];
        for (int r = 0; r < 3; ++r)                                         // new_R^T (w - new_P)
            pj[r] = new_R[r] * (w[0] - new_P[0]) + new_R[3 + r] * (w[1] - new_P[1]) + new_R[6 + r] * (w[2] - new_P[2]);
        f.estimated_depth = pj[2] > 0 ? pj[2] : INIT_DEPTH;
        kept.push_back(f);
    }
    feature.swap(kept);
}

void FeatureManager::removeBack() {                 // :314-329
    std::vector<FeaturePerId> kept;
    kept.reserve(feature.size());
    for (auto &f : feature) {
        if (f.start_frame != 0) { --f.start_frame; kept.push_back(f); continue; }
        f.feature_per_frame.erase(f.feature_per_frame.begin());
        if (!f.feature_per_frame.empty()) kept.push_back(f);
    }
    feature.swap(kept);
}

void FeatureManager::removeFront(int frame_count) { // :331-350
    std::vector<FeaturePerId> kept;
    kept.reserve(feature.size());
    for (auto &f : feature) {
        if (f.start_frame == frame_count) { --f.start_frame; kept.push_back(f); continue; }
        const int j = WINDOW_SIZE - 1 - f.start_frame;
        if (f.endFrame() < frame_count - 1) { kept.push_back(f); continue; }
        f.feature_per_frame.erase(f.feature_per_frame.begin() + j);
        if (!f.feature_per_frame.empty()) kept.push_back(f);
    }
    feature.swap(kept);
}

}  // namespace vio
