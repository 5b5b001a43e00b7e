// estimator_backend.h — C++ host mirror of the three Estimator methods that talk to the backend
//  and of the two that convert its state for them (8a1).  Same names, same data members, same
// call order as
//   Estimator::vector2double         VM/src/estimator.cpp:505-547
//   Estimator::double2vector         VM/src/estimator.cpp:549-617   (the relocalisation tail :620-643 is out of scope)
//   Estimator::problemSolve          VM/src/estimator.cpp:902-1073
//   Estimator::MargOldFrame          VM/src/estimator.cpp:693-829
//   Estimator::MargNewFrame          VM/src/estimator.cpp:830-901
//   Estimator::backendOptimization   VM/src/estimator.cpp:1075-1141
// but over plain arrays instead of Eigen objects, and with the graph handed to the C ABI of include/vio_backend.h
// instead of being built from heap-allocated Vertex/Edge objects.  A maintainer of the reference replaces the
// bodies of those methods with calls to this class (or copies its ~100 lines into estimator.cpp).
#ifndef VIO_ESTIMATOR_BACKEND_H
#define VIO_ESTIMATOR_BACKEND_H

#include <array>
#include <string>
#include <vector>

#include "../../include/vio_backend.h"

namespace vio {

constexpr int WINDOW_SIZE = VIO_WINDOW_SIZE;   // parameters.h:35

// FeaturePerId / FeaturePerFrame (VM/include/feature_manager.h): one track = consecutive frames from start_frame
struct FeaturePerId {
    int start_frame = 0;
    std::vector<std::array<double, 2>> feature_per_frame;   // normalised (x, y), z == 1
    int used_num = 0;
    int feature_id = 0;
    double estimated_depth = -1.0;                           // FeaturePerId::estimated_depth (feature_manager.h:62,74): <= 0 = not triangulated yet
    int solve_flag = 0;                                      // 0 haven't solved yet, 1 solved, 2 solve failed (feature_manager.h:63)
    int endFrame() const { return start_frame + (int)feature_per_frame.size() - 1; }
};

enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };   // estimator.h:52-56

class EstimatorBackend {
public:
    explicit EstimatorBackend(const vio_config &cfg);
    ~EstimatorBackend();
    EstimatorBackend(const EstimatorBackend &) = delete;
    EstimatorBackend &operator=(const EstimatorBackend &) = delete;

    // ---- the members Estimator keeps (estimator.h:66-69,113-119) ----
    double Ps[WINDOW_SIZE + 1][3], Vs[WINDOW_SIZE + 1][3], Bas[WINDOW_SIZE + 1][3], Bgs[WINDOW_SIZE + 1][3];
    double Rs[WINDOW_SIZE + 1][9];                    // row-major 3x3
    double tic[1][3], ric[1][9];
    bool failure_occur = false;                       // estimator.h:129-131: re-anchor to last_R0 / last_P0 once
    double last_R0[9], last_P0[3];
    double para_Pose[WINDOW_SIZE + 1][7];
    double para_SpeedBias[WINDOW_SIZE + 1][9];
    double para_Ex_Pose[1][7];
    std::vector<double> para_Feature;                 // [feature_index] (the reference caps this at NUM_OF_F = 1000)
    std::vector<FeaturePerId> feature;                // f_manager.feature
    const vio_preint *pre_integrations[WINDOW_SIZE + 1];   // [j] = between frames j-1 and j; nullptr or sum_dt > 10 skips the edge
    std::vector<double> Hprior_, bprior_, errprior_, Jprior_inv_;   // empty until the first marginalisation
    vio_solve_report last_report;

    // ---- the methods ----
    void vector2double();                             // Ps/Rs/Vs/Bas/Bgs/tic/ric + the depth vector -> para_*
    void double2vector();                             // para_* -> Ps/Rs/..., yaw and position re-anchored to the pre-solve frame 0; setDepth
    bool problemSolve();
    bool MargOldFrame();
    bool MargNewFrame();
    void backendOptimization(MarginalizationFlag marginalization_flag);   // vector2double .. double2vector (estimator.cpp:1075-1141)
    // Marg*Frame return once the device part of Problem::Marginalize is done and leave its dense tail (two eigen-decompositions,
    // 0.2 - 0.5 ms) to a helper thread of the library (vio_marginalize_begin / _end): the prior members are valid after
    // waitMarginalization(), which the next problemSolve calls itself where it hands the prior over.  Off: as the reference.
    bool async_marginalization = false;
    bool waitMarginalization();
    const char *last_error() const;

private:
    // graph_unchanged: landmarks, observations and IMU factors are the ones of the upload before (backendOptimization's second
    // upload, for the marginalisation of the frame it has just solved): only the states and the prior are sent
    bool uploadWindow(bool graph_unchanged = false);
    bool marginalize(int kind);
    bool marg_pending_ = false;
    bool graph_uploaded_ = false;                     // set by problemSolve inside backendOptimization, used by the Marg*Frame that follows
public:
    vio_ctx *context() { return ctx_; }               // for FeatureManager::triangulate
private:
    vio_ctx *ctx_ = nullptr;
    std::string err_;
};

}  // namespace vio
#endif
