// feature_manager.h — C++ host mirror of the FeatureManager methods on either side of the solve:
//   getFeatureCount        VM/src/feature_manager.cpp:37-52
//   getDepthVector         :184-200      setDepth :141-160      clearDepth :173-182      removeFailures :162-171
//   triangulate            :203-257      (the per-track SVDs run on the GPU through vio_triangulate)
//   removeBackShiftDepth   :276-312      removeBack :314-329    removeFront :331-350
// over the same FeaturePerId records EstimatorBackend::feature holds.  Same names, same rules (used_num >= 2 and
// start_frame < WINDOW_SIZE - 2 decide whether a track is a landmark of the window), plain arrays instead of Eigen.
//   addFeatureCheckParallax :55-115     compensatedParallax2 :352-388   (the hand-over from the front-end: one image's
//                                       points appended to the tracks + the keyframe decision)
// addObservation appends one tracked point the way :62-90 does.
#ifndef VIO_FEATURE_MANAGER_H
#define VIO_FEATURE_MANAGER_H

#include <vector>

#include "estimator_backend.h"

namespace vio {

constexpr double INIT_DEPTH = 5.0;      // parameters.cpp:126

class FeatureManager {
public:
    explicit FeatureManager(std::vector<FeaturePerId> &tracks) : feature(tracks) {}
    std::vector<FeaturePerId> &feature;                 // f_manager.feature (a std::list in the reference; order = insertion)

    double MIN_PARALLAX = 10.0 / 460.0;                 // keyframe_parallax / FOCAL_LENGTH (parameters.cpp:103, vio_simulation.yaml)
    int last_track_num = 0;

    // a point (normalised x, y) of feature `feature_id` seen in frame `frame_count`: new track or one more frame;
    // returns true when the feature was already tracked
    bool addObservation(int feature_id, int frame_count, double x, double y);
    // one image: n points (id, x, y), taken in ascending id order as the reference's std::map does; true = the second
    // newest frame is a keyframe (marginalise the oldest), false = not (marginalise the second newest)
    bool addFeatureCheckParallax(int frame_count, int n, const int *ids, const double *pts_xy);
    int getFeatureCount();
    std::vector<double> getDepthVector();
    void setDepth(const std::vector<double> &x);
    void clearDepth(const std::vector<double> &x);
    void removeFailures();
    // Ps / Rs as para_Pose rows (p, q xyzw), tic/ric as para_Ex_Pose; returns false and keeps the depths on an ABI error
    bool triangulate(vio_ctx *ctx, const double poses[][7], const double ext[7]);
    void removeBackShiftDepth(const double marg_R[9], const double marg_P[3], const double new_R[9], const double new_P[3]);
    void removeBack();
    void removeFront(int frame_count);

private:
    static double compensatedParallax2(const FeaturePerId &it_per_id, int frame_count);
    static bool usable(FeaturePerId &f) {               // :146-148 and every other loop of the file
        f.used_num = (int)f.feature_per_frame.size();
        return f.used_num >= 2 && f.start_frame < WINDOW_SIZE - 2;
    }
};

}  // namespace vio
#endif
