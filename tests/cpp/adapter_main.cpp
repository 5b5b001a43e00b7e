// Test driver for the C++ host mirror (EstimatorBackend + FeatureManager): reads a flat binary file written by the
// pytest, runs one of three programs, writes the results back.       usage: adapter_main <in> <out> [mode]
//
// mode 0  problemSolve() + MargOldFrame() on given para_* arrays
//   in : int64 n_tracks; per track: int32 start, int32 n_obs, double estimated_depth, n_obs x (x,y);
//        77 + 99 + 7 doubles (para_Pose, para_SpeedBias, para_Ex_Pose); 10 x vio_preint; int32 has_prior;
//        [156*156 + 156 + 156 + 156*156 doubles]
//   out: para_Pose 77, para_SpeedBias 99, int64 nf, para_Feature nf, Hprior 156*156, bprior 156, errprior 156,
//        (iterations, final_chi2, final_lambda) as doubles
// mode 1  the frame chain of Estimator::processImage (estimator.cpp:157-166): f_manager.triangulate ->
//         backendOptimization(MARGIN_OLD) [vector2double, problemSolve, double2vector, vector2double, MargOldFrame] ->
//         f_manager.removeFailures
//   in : tracks as above (estimated_depth <= 0: not triangulated yet); Ps 33, Rs 99, Vs 33, Bas 33, Bgs 33, tic 3, ric 9;
//        10 x vio_preint
//   out: Ps 33, Rs 99, Vs 33, Bas 33, Bgs 33; int64 n_tracks; per track: int32 id, int32 solve_flag, double estimated_depth;
//        Hprior 156*156, bprior 156; (iterations, final_chi2) as doubles
// mode 2  double2vector() then vector2double() alone (no device): in: Rs[0] 9, Ps[0] 3, para_Pose 77, para_SpeedBias 99;
//         out: Rs 99, Ps 33, Vs 33, para_Pose 77
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../visual-inertial-odometry_amd/host/feature_manager.h"

template <typename T>
static bool rd(FILE *f, T *p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

static bool read_tracks(FILE *f, std::vector<vio::FeaturePerId> &tracks) {
    int64_t nt;
    if (!rd(f, &nt, 1)) return false;
    tracks.resize(nt);
    int id = 0;
    for (auto &t : tracks) {
        int32_t start, nobs;
        if (!rd(f, &start, 1) || !rd(f, &nobs, 1) || !rd(f, &t.estimated_depth, 1)) return false;
        t.start_frame = start;
        t.feature_id = id++;
        t.feature_per_frame.resize(nobs);
        for (auto &p : t.feature_per_frame) if (!rd(f, p.data(), 2)) return false;
    }
    return true;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int mode = argc > 3 ? std::atoi(argv[3]) : 0;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    vio_config cfg;
    vio_default_config(&cfg);
    vio::EstimatorBackend est(cfg);
    if (std::getenv("VIO_ASYNC_MARG")) est.async_marginalization = true;      // Marg*Frame leave the dense tail running (vio_marginalize_begin / _end)
    std::vector<vio_preint> pre(10);

    if (mode == 2) {
        if (!rd(f, est.Rs[0], 9) || !rd(f, est.Ps[0], 3) || !rd(f, &est.para_Pose[0][0], 77) || !rd(f, &est.para_SpeedBias[0][0], 99)) return 4;
        std::fclose(f);
        est.para_Ex_Pose[0][6] = 1.0;
        est.double2vector();
        FILE *o = std::fopen(argv[2], "wb");
        std::fwrite(&est.Rs[0][0], 8, 99, o);
        std::fwrite(&est.Ps[0][0], 8, 33, o);
        std::fwrite(&est.Vs[0][0], 8, 33, o);
        est.vector2double();
        std::fwrite(&est.para_Pose[0][0], 8, 77, o);
        std::fclose(o);
        return 0;
    }

    if (!read_tracks(f, est.feature)) return 4;
    if (mode == 0) {
        if (!rd(f, &est.para_Pose[0][0], 77) || !rd(f, &est.para_SpeedBias[0][0], 99) || !rd(f, &est.para_Ex_Pose[0][0], 7)) return 4;
    } else {
        if (!rd(f, &est.Ps[0][0], 33) || !rd(f, &est.Rs[0][0], 99) || !rd(f, &est.Vs[0][0], 33) || !rd(f, &est.Bas[0][0], 33) ||
            !rd(f, &est.Bgs[0][0], 33) || !rd(f, &est.tic[0][0], 3) || !rd(f, &est.ric[0][0], 9)) return 4;
    }
    if (!rd(f, pre.data(), 10)) return 4;
    for (int j = 1; j <= 10; ++j) est.pre_integrations[j] = &pre[j - 1];

    if (mode == 0) {
        int32_t has_prior;
        if (!rd(f, &has_prior, 1)) return 4;
        if (has_prior) {
            est.Hprior_.resize(156 * 156); est.bprior_.resize(156); est.errprior_.resize(156); est.Jprior_inv_.resize(156 * 156);
            if (!rd(f, est.Hprior_.data(), 156 * 156) || !rd(f, est.bprior_.data(), 156) || !rd(f, est.errprior_.data(), 156) ||
                !rd(f, est.Jprior_inv_.data(), 156 * 156)) return 4;
        }
        std::fclose(f);
        for (auto &t : est.feature) {                       // the depth part of vector2double (the para_Pose arrays are given)
            t.used_num = (int)t.feature_per_frame.size();
            if (t.used_num >= 2 && t.start_frame < vio::WINDOW_SIZE - 2) est.para_Feature.push_back(1.0 / t.estimated_depth);
        }
        if (const char *reps_s = std::getenv("VIO_TIME_REPS")) {
            // timing of the frame's backend part as an integrator's C++ sees it (tools/bench_cpp_frame.py): the same inputs again
            // and again, problemSolve() (upload, Solve(10), read-back) and MargOldFrame(), wall clock per call
            const int reps = std::max(1, std::atoi(reps_s));
            double keep_pose[77], keep_sb[99], keep_ex[7];
            std::memcpy(keep_pose, &est.para_Pose[0][0], sizeof(keep_pose)); std::memcpy(keep_sb, &est.para_SpeedBias[0][0], sizeof(keep_sb));
            std::memcpy(keep_ex, &est.para_Ex_Pose[0][0], sizeof(keep_ex));
            const std::vector<double> keep_f = est.para_Feature, kH = est.Hprior_, kb = est.bprior_, ke = est.errprior_, kJ = est.Jprior_inv_;
            double t_solve = 0, t_marg = 0;
            for (int r = 0; r <= reps; ++r) {
                std::memcpy(&est.para_Pose[0][0], keep_pose, sizeof(keep_pose)); std::memcpy(&est.para_SpeedBias[0][0], keep_sb, sizeof(keep_sb));
                std::memcpy(&est.para_Ex_Pose[0][0], keep_ex, sizeof(keep_ex));
                est.waitMarginalization();          // (async_marginalization: the tail of the pass before, outside the timed calls — where a frame loop has its front-end)
                est.para_Feature = keep_f; est.Hprior_ = kH; est.bprior_ = kb; est.errprior_ = ke; est.Jprior_inv_ = kJ;
                // a frame never repeats the one before (the library keeps the plans of a graph it already holds): one observation moves by 1e-13
                if (!est.feature.empty() && est.feature[0].feature_per_frame.size() > 1) est.feature[0].feature_per_frame[1][0] += (r & 1) ? 1e-13 : -1e-13;
                const auto t0 = std::chrono::steady_clock::now();
                if (!est.problemSolve()) { std::fprintf(stderr, "backend failed: %s\n", est.last_error()); return 5; }
                const auto t1 = std::chrono::steady_clock::now();
                if (!est.MargOldFrame()) { std::fprintf(stderr, "backend failed: %s\n", est.last_error()); return 5; }
                const auto t2 = std::chrono::steady_clock::now();
                if (r == 0) continue;           // first pass: allocations
                t_solve += std::chrono::duration<double, std::milli>(t1 - t0).count();
                t_marg += std::chrono::duration<double, std::milli>(t2 - t1).count();
            }
            std::printf("cpp_frame problemSolve_ms %.4f MargOldFrame_ms %.4f frame_ms %.4f reps %d landmarks %zu iterations %d\n", t_solve / reps, t_marg / reps,
                        (t_solve + t_marg) / reps, reps, est.para_Feature.size(), est.last_report.iterations);
            return 0;
        }
        if (!est.problemSolve() || !est.MargOldFrame() || !est.waitMarginalization()) { std::fprintf(stderr, "backend failed: %s\n", est.last_error()); return 5; }
        FILE *o = std::fopen(argv[2], "wb");
        std::fwrite(&est.para_Pose[0][0], 8, 77, o);
        std::fwrite(&est.para_SpeedBias[0][0], 8, 99, o);
        int64_t nf = (int64_t)est.para_Feature.size();
        std::fwrite(&nf, 8, 1, o);
        std::fwrite(est.para_Feature.data(), 8, nf, o);
        std::fwrite(est.Hprior_.data(), 8, 156 * 156, o);
        std::fwrite(est.bprior_.data(), 8, 156, o);
        std::fwrite(est.errprior_.data(), 8, 156, o);
        double info[3] = {(double)est.last_report.iterations, est.last_report.final_chi2, est.last_report.final_lambda};
        std::fwrite(info, 8, 3, o);
        std::fclose(o);
        return 0;
    }

    if (const char *reps_s = std::getenv("VIO_TIME_REPS")) {
        // the frame's backend part exactly as Estimator::backendOptimization runs it (vector2double, problemSolve, double2vector,
        // vector2double, MargOldFrame), timed as a whole, the same inputs restored before every pass; an optional prior follows
        // the pre-integrations in the input file
        const int reps = std::max(1, std::atoi(reps_s));
        int32_t has_prior = 0;
        std::vector<double> kH, kb, ke, kJ;
        if (rd(f, &has_prior, 1) && has_prior) {
            kH.resize(156 * 156); kb.resize(156); ke.resize(156); kJ.resize(156 * 156);
            if (!rd(f, kH.data(), kH.size()) || !rd(f, kb.data(), 156) || !rd(f, ke.data(), 156) || !rd(f, kJ.data(), kJ.size())) return 4;
        }
        std::fclose(f);
        double kP[33], kR[99], kV[33], kBa[33], kBg[33];
        std::memcpy(kP, &est.Ps[0][0], sizeof(kP)); std::memcpy(kR, &est.Rs[0][0], sizeof(kR)); std::memcpy(kV, &est.Vs[0][0], sizeof(kV));
        std::memcpy(kBa, &est.Bas[0][0], sizeof(kBa)); std::memcpy(kBg, &est.Bgs[0][0], sizeof(kBg));
        std::vector<double> kdepth;
        for (auto &t : est.feature) kdepth.push_back(t.estimated_depth);
        double total = 0;
        for (int r = 0; r <= reps; ++r) {
            std::memcpy(&est.Ps[0][0], kP, sizeof(kP)); std::memcpy(&est.Rs[0][0], kR, sizeof(kR)); std::memcpy(&est.Vs[0][0], kV, sizeof(kV));
            std::memcpy(&est.Bas[0][0], kBa, sizeof(kBa)); std::memcpy(&est.Bgs[0][0], kBg, sizeof(kBg));
            size_t q = 0;
            for (auto &t : est.feature) { t.estimated_depth = kdepth[q++]; t.solve_flag = 0; }
            // (with the marginalisation left running the prior of a pass is what the pass before computed, as in a stream: nothing to restore)
            if (!est.async_marginalization || r == 0) { est.Hprior_ = kH; est.bprior_ = kb; est.errprior_ = ke; est.Jprior_inv_ = kJ; }
            if (!est.feature.empty() && est.feature[0].feature_per_frame.size() > 1) est.feature[0].feature_per_frame[1][0] += (r & 1) ? 1e-13 : -1e-13;     // (a new frame)
            const auto t0 = std::chrono::steady_clock::now();
            est.backendOptimization(vio::MARGIN_OLD);
            const auto t1 = std::chrono::steady_clock::now();
            if (r == reps) est.waitMarginalization();
            if (!est.async_marginalization && est.Hprior_.size() != 156 * 156) { std::fprintf(stderr, "backend failed: %s\n", est.last_error()); return 5; }
            if (r) total += std::chrono::duration<double, std::milli>(t1 - t0).count();
        }
        std::printf("cpp_frame backendOptimization_ms %.4f reps %d tracks %zu iterations %d\n", total / reps, reps, est.feature.size(), est.last_report.iterations);
        return 0;
    }
    std::fclose(f);
    vio::FeatureManager f_manager(est.feature);
    est.vector2double();                                    // the mirror's triangulate takes Ps / Rs as para_Pose rows
    if (!f_manager.triangulate(est.context(), est.para_Pose, est.para_Ex_Pose[0])) {
        std::fprintf(stderr, "triangulate failed: %s\n", est.last_error());
        return 6;
    }
    est.backendOptimization(vio::MARGIN_OLD);
    est.waitMarginalization();
    if (est.Hprior_.size() != 156 * 156) { std::fprintf(stderr, "backend failed: %s\n", est.last_error()); return 5; }
    f_manager.removeFailures();
    FILE *o = std::fopen(argv[2], "wb");
    std::fwrite(&est.Ps[0][0], 8, 33, o);
    std::fwrite(&est.Rs[0][0], 8, 99, o);
    std::fwrite(&est.Vs[0][0], 8, 33, o);
    std::fwrite(&est.Bas[0][0], 8, 33, o);
    std::fwrite(&est.Bgs[0][0], 8, 33, o);
    int64_t n = (int64_t)est.feature.size();
    std::fwrite(&n, 8, 1, o);
    for (auto &t : est.feature) {
        int32_t h[2] = {t.feature_id, t.solve_flag};
        std::fwrite(h, 4, 2, o);
        std::fwrite(&t.estimated_depth, 8, 1, o);
    }
    std::fwrite(est.Hprior_.data(), 8, 156 * 156, o);
    std::fwrite(est.bprior_.data(), 8, 156, o);
    double info[2] = {(double)est.last_report.iterations, est.last_report.final_chi2};
    std::fwrite(info, 8, 2, o);
    std::fclose(o);
    return 0;
}
