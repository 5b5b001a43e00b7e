// problem_hip.cc — drop-in replacement for VM/src/backend/problem.cc of the reference.
//
// The reference's include/backend/problem.h stays as it is: this file implements the public members of
// myslam::backend::Problem that Estimator uses — AddVertex, AddEdge, ExtendHessiansPriorSize, Solve, Marginalize (the
// vector form), the destructor's id reset — on top of the C ABI of include/vio_backend.h.  AddVertex / AddEdge only
// record the graph (the containers the header declares); Solve and Marginalize flatten it into the window layout
//   [ext | (pose, speed-bias) x 11 | inverse depths or XYZ points],  one row per EdgeReprojection / EdgeReprojectionXYZ,
//   one record per EdgeImu,
// hand it to the backend, and write the results back into the Vertex objects and the prior members, so that an
// unmodified estimator.cpp (problemSolve :902-1073, MargOldFrame :693-829, MargNewFrame :830-901) links against it.
// A maintainer swaps this file for problem.cc in VM/CMakeLists.txt and links libvio_hip.so.
//
// Two things the reference's headers keep private and the flat layout needs: the observations of an
// EdgeReprojection (pts_i_, pts_j_, edge_reprojection.h) and the delta of a loss function (loss_function.h); and the
// IntegrationBase of an EdgeImu (edge_imu.h).  The first two are read through `#define private public` around the
// reference's own headers (no layout change); the third through vio_shim_edge_imu(), defined below for the reference's
// EdgeImu when VIO_SHIM_WITH_EDGE_IMU is set (its header needs Ceres: the reference tree has it, this repo's image
// does not) and by the test harness otherwise.
//
// One backend context serves every Problem of the process (a function-local static, re-configured per graph with
// vio_set_config): the reference builds a fresh Problem per call — one for Solve, another for Marginalize, each with its own
// graph (MargOldFrame keeps the landmarks hosted in frame 0 only) — so two uploads per frame are the reference's own
// construction; what is not repeated is the stream and the ~30 device allocations behind the handle.
//
// Only the SLAM problem type over the window layout is supported; anything else fails loudly (Solve returns false
// after printing the backend's message).  RemoveVertex/RemoveEdge/GetOutlierEdges/TestComputePrior, which Estimator
// never calls, are not provided.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

// every standard / Eigen header the reference's headers pull in comes first: the two defines below must not reach them
#include <eigen3/Eigen/Dense>
#include <Eigen/Dense>

#define private public
#define protected public
#include "backend/problem.h"
#include "backend/vertex_pose.h"
#include "backend/vertex_speedbias.h"
#include "backend/vertex_inverse_depth.h"
#include "backend/vertex_point_xyz.h"
#include "backend/edge_reprojection.h"
#include "backend/loss_function.h"
#ifdef VIO_SHIM_WITH_EDGE_IMU
#include "backend/edge_imu.h"
#include "parameters.h"          // the global G that IntegrationBase::evaluate reads (parameters.h:46, integration_base.h:178-180)
#endif
#undef private
#undef protected

#include "vio_backend.h"

// The C ABI's symbol prefix: vio_ for the HIP library, vioo_ for the CPU oracle (tests), set with -DVIO_SHIM_PREFIX=vioo_
#ifndef VIO_SHIM_PREFIX
#define VIO_SHIM_PREFIX vio_
#endif
#define VIO_CAT2(a, b) a##b
#define VIO_CAT(a, b) VIO_CAT2(a, b)
#define ABI(name) VIO_CAT(VIO_SHIM_PREFIX, name)
extern "C" {
vio_status ABI(create)(const vio_config *, struct vio_ctx **);
void ABI(destroy)(struct vio_ctx *);
const char *ABI(last_error)(const struct vio_ctx *);
void ABI(default_config)(vio_config *);
vio_status ABI(set_config)(struct vio_ctx *, const vio_config *);
vio_status ABI(set_landmarks_xyz)(struct vio_ctx *, int64_t, const double *);
vio_status ABI(set_observations_xyz)(struct vio_ctx *, int64_t, const int32_t *, const int32_t *, const double *);
vio_status ABI(get_landmarks_xyz)(struct vio_ctx *, int64_t, double *);
vio_status ABI(set_window)(struct vio_ctx *, const double *, const double *, const double *);
vio_status ABI(set_landmarks)(struct vio_ctx *, int64_t, const double *);
vio_status ABI(set_observations)(struct vio_ctx *, int64_t, const int32_t *, const int32_t *, const int32_t *, const double *, const double *);
vio_status ABI(set_imu)(struct vio_ctx *, int32_t, const vio_preint *);
vio_status ABI(set_prior)(struct vio_ctx *, int32_t, const double *, const double *, const double *, const double *);
vio_status ABI(solve)(struct vio_ctx *, int32_t, vio_solve_report *);
vio_status ABI(marginalize)(struct vio_ctx *, int32_t, double *, double *, double *, double *);
vio_status ABI(get_window)(struct vio_ctx *, double *, double *, double *);
vio_status ABI(get_landmarks)(struct vio_ctx *, int64_t, double *);
vio_status ABI(get_prior)(struct vio_ctx *, double *, double *);
}

// The pre-integration an "EdgeImu" carries, and the gravity / device the backend should use (globals G of parameters.h).
// Returns false when the edge is not an IMU edge it knows.
extern "C" bool vio_shim_edge_imu(myslam::backend::Edge *edge, vio_preint *out);
extern "C" void vio_shim_config(vio_config *cfg);          // optional override of the defaults (weak)

#ifdef VIO_SHIM_WITH_EDGE_IMU
extern "C" bool vio_shim_edge_imu(myslam::backend::Edge *edge, vio_preint *out) {
    auto *e = dynamic_cast<myslam::backend::EdgeImu *>(edge);
    if (!e || !e->pre_integration_) return false;
    const IntegrationBase &p = *e->pre_integration_;
    out->sum_dt = p.sum_dt;
    for (int k = 0; k < 3; ++k) {
        out->delta_p[k] = p.delta_p[k]; out->delta_v[k] = p.delta_v[k];
        out->linearized_ba[k] = p.linearized_ba[k]; out->linearized_bg[k] = p.linearized_bg[k];
    }
    out->delta_q[0] = p.delta_q.x(); out->delta_q[1] = p.delta_q.y(); out->delta_q[2] = p.delta_q.z(); out->delta_q[3] = p.delta_q.w();
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) { out->jacobian[15 * i + j] = p.jacobian(i, j); out->covariance[15 * i + j] = p.covariance(i, j); }
    return true;
}
#endif
extern "C" __attribute__((weak)) void vio_shim_config(vio_config *) {}

namespace myslam {
namespace backend {

namespace {

const int NF = VIO_NUM_FRAMES, PD = VIO_POSE_DIM, PRD = VIO_PRIOR_DIM;

bool is_pose_type(const std::shared_ptr<Vertex> &v) {
    const std::string t = v->TypeInfo();
    return t == "VertexPose" || t == "VertexSpeedBias";
}

// What Solve and Marginalize hand to the backend
struct FlatWindow {
    std::shared_ptr<Vertex> ext;
    std::vector<std::shared_ptr<Vertex>> pose, sb, lm;          // frames in id order, landmarks in id order
    std::vector<int32_t> o_lm, o_host, o_target;                // XYZ: o_target = the observing frame, o_pj = the observation
    std::vector<double> o_pi, o_pj;
    bool xyz = false;                                           // VertexPointXYZ landmarks observed through EdgeReprojectionXYZ
    double xyz_ext[7];                                          // their camera extrinsic: a constant of those edges (qic, tic)
    vio_config cfg;
    bool ok = false;
    std::string why;
};

int index_of(const std::vector<std::shared_ptr<Vertex>> &v, const std::shared_ptr<Vertex> &x) {
    for (size_t i = 0; i < v.size(); ++i) if (v[i]->Id() == x->Id()) return (int)i;
    return -1;
}

int loss_of(LossFunction *lf, double *delta) {
    *delta = 1.0;
    if (!lf || dynamic_cast<TrivalLoss *>(lf)) return VIO_LOSS_TRIVIAL;
    if (auto *h = dynamic_cast<HuberLoss *>(lf)) { *delta = h->delta_; return VIO_LOSS_HUBER; }
    if (auto *c = dynamic_cast<CauchyLoss *>(lf)) { *delta = c->delta_; return VIO_LOSS_CAUCHY; }
    if (auto *t = dynamic_cast<TukeyLoss *>(lf)) { *delta = t->delta_; return VIO_LOSS_TUKEY; }
    return -1;
}

// verticies_ is ordered by id (std::map): the Estimator creates ext, then (pose_i, speed-bias_i) for i = 0..10, then the
// inverse depths (estimator.cpp:915-953,988-993).  The extrinsic is the vertex the reprojection edges name fourth; with no
// such edge (MargNewFrame) it is the pose vertex created first.
FlatWindow flatten(Problem &p) {
    FlatWindow w;
    ABI(default_config)(&w.cfg);
#ifdef VIO_SHIM_WITH_EDGE_IMU
    for (int k = 0; k < 3; ++k) w.cfg.gravity[k] = G[k];
#endif
    vio_shim_config(&w.cfg);
    std::vector<std::shared_ptr<Edge>> edges;
    for (auto &kv : p.edges_) edges.push_back(kv.second);
    std::sort(edges.begin(), edges.end(), [](const std::shared_ptr<Edge> &a, const std::shared_ptr<Edge> &b) { return a->Id() < b->Id(); });
    for (auto &e : edges)
        if (e->TypeInfo() == "EdgeReprojection" && e->verticies_.size() == 4) { w.ext = e->verticies_[3]; break; }
    for (auto &kv : p.verticies_) {
        const std::shared_ptr<Vertex> &v = kv.second;
        const std::string t = v->TypeInfo();
        if (t == "VertexPose") {
            if (!w.ext) w.ext = v;
            if (v->Id() != w.ext->Id()) w.pose.push_back(v);
        } else if (t == "VertexSpeedBias") w.sb.push_back(v);
        else if (t == "VertexInverseDepth") { if (w.xyz) { w.why = "a window holds one kind of landmark"; return w; } w.lm.push_back(v); }
        else if (t == "VertexPointXYZ") { if (!w.xyz && !w.lm.empty()) { w.why = "a window holds one kind of landmark"; return w; } w.xyz = true; w.lm.push_back(v); }
        else { w.why = "vertex type outside the window layout: " + t; return w; }
    }
    if (!w.ext || (int)w.pose.size() != NF || (int)w.sb.size() != NF) { w.why = "expected 1 extrinsic + 11 (pose, speed-bias) pairs"; return w; }
    w.cfg.ext_fixed = w.ext->IsFixed() ? 1 : 0;
    bool have_loss = false;
    bool have_ext = false;
    for (auto &e : edges) {
        if (e->TypeInfo() == "EdgeReprojectionXYZ") {
            auto *re = static_cast<EdgeReprojectionXYZ *>(e.get());
            const int l = index_of(w.lm, e->verticies_[0]), f = index_of(w.pose, e->verticies_[1]);
            if (!w.xyz || l < 0 || f < 0) { w.why = "XYZ reprojection edge with vertices outside the window"; return w; }
            const double ex[7] = {re->tic[0], re->tic[1], re->tic[2], re->qic.x(), re->qic.y(), re->qic.z(), re->qic.w()};
            if (have_ext && std::memcmp(ex, w.xyz_ext, sizeof(ex)) != 0) { w.why = "XYZ edges differ in their camera extrinsic"; return w; }
            std::memcpy(w.xyz_ext, ex, sizeof(ex));
            have_ext = true;
            w.o_lm.push_back(l); w.o_target.push_back(f);
            w.o_pj.push_back(re->obs_[0] / re->obs_[2]); w.o_pj.push_back(re->obs_[1] / re->obs_[2]);
        } else if (e->TypeInfo() != "EdgeReprojection") continue;
        else {
        const int l = index_of(w.lm, e->verticies_[0]), h = index_of(w.pose, e->verticies_[1]), t = index_of(w.pose, e->verticies_[2]);
        if (l < 0 || h < 0 || t < 0 || e->verticies_[3]->Id() != w.ext->Id()) { w.why = "reprojection edge with vertices outside the window"; return w; }
        auto *re = static_cast<EdgeReprojection *>(e.get());
        w.o_lm.push_back(l); w.o_host.push_back(h); w.o_target.push_back(t);
        w.o_pi.push_back(re->pts_i_[0] / re->pts_i_[2]); w.o_pi.push_back(re->pts_i_[1] / re->pts_i_[2]);
        w.o_pj.push_back(re->pts_j_[0] / re->pts_j_[2]); w.o_pj.push_back(re->pts_j_[1] / re->pts_j_[2]);
        }
        const MatXX info = e->Information();
        double delta;
        const int loss = loss_of(e->lossfunction_, &delta);
        const double s = std::sqrt(info(0, 0));
        if (loss < 0 || info(0, 1) != 0.0 || info(1, 0) != 0.0 || info(1, 1) != info(0, 0)) { w.why = "edge information / loss outside the layout (one isotropic information, one loss for all edges)"; return w; }
        if (have_loss && (loss != w.cfg.loss_type || delta != w.cfg.loss_delta || s != w.cfg.reproj_sqrt_info)) { w.why = "edges differ in loss or information"; return w; }
        w.cfg.loss_type = loss; w.cfg.loss_delta = delta; w.cfg.reproj_sqrt_info = s;
        have_loss = true;
    }
    w.ok = true;
    return w;
}

struct Ctx {
    struct vio_ctx *h = nullptr;
};

// The one backend context of the process, created at the first graph and re-configured for every later one.  It is never
// destroyed: a static destructor would call into the HIP runtime (stream synchronise, frees) at process exit, in an order
// relative to the runtime's own teardown that depends on how the host application was linked; the process's exit releases
// the device.  Like the reference's Problem (process-global vertex / edge id counters, VM/src/backend/vertex.cc:7,
// edge.cc:11; one caller thread under m_estimator, VM/src/System.cpp:358-441), Problems must not be solved or marginalised
// from several threads at once: they share this context.
Ctx &shared_ctx() {
    static Ctx *c = new Ctx();
    return *c;
}

// window + landmarks + observations + IMU edges + prior into the backend context
bool upload(Problem &p, const FlatWindow &w, Ctx &c) {
    if (!c.h) {
        if (ABI(create)(&w.cfg, &c.h) != VIO_OK) { c.h = nullptr; std::cerr << "vio_create failed" << std::endl; return false; }
    } else if (ABI(set_config)(c.h, &w.cfg) != VIO_OK) { std::cerr << "vio_set_config: " << ABI(last_error)(c.h) << std::endl; return false; }
    double poses[NF * 7], sbs[NF * 9], ext[7];
    for (int i = 0; i < NF; ++i) {
        for (int k = 0; k < 7; ++k) poses[7 * i + k] = w.pose[i]->parameters_[k];
        for (int k = 0; k < 9; ++k) sbs[9 * i + k] = w.sb[i]->parameters_[k];
    }
    for (int k = 0; k < 7; ++k) ext[k] = w.xyz && !w.o_lm.empty() ? w.xyz_ext[k] : w.ext->parameters_[k];
    const int dim = w.xyz ? 3 : 1;
    std::vector<double> lmv(w.lm.size() * dim);
    for (size_t l = 0; l < w.lm.size(); ++l) for (int k = 0; k < dim; ++k) lmv[dim * l + k] = w.lm[l]->parameters_[k];
    bool ok = ABI(set_window)(c.h, poses, sbs, ext) == VIO_OK;
    if (w.xyz) ok = ok && ABI(set_landmarks_xyz)(c.h, (int64_t)w.lm.size(), lmv.data()) == VIO_OK
                       && ABI(set_observations_xyz)(c.h, (int64_t)w.o_lm.size(), w.o_lm.data(), w.o_target.data(), w.o_pj.data()) == VIO_OK;
    else ok = ok && ABI(set_landmarks)(c.h, (int64_t)lmv.size(), lmv.data()) == VIO_OK
                 && ABI(set_observations)(c.h, (int64_t)w.o_lm.size(), w.o_lm.data(), w.o_host.data(), w.o_target.data(), w.o_pi.data(), w.o_pj.data()) == VIO_OK;
    for (int k = 0; k < VIO_WINDOW_SIZE; ++k) ok = ok && ABI(set_imu)(c.h, k, nullptr) == VIO_OK;      // the context outlives the graph: start from no IMU edges
    for (auto &kv : p.edges_) {
        Edge *e = kv.second.get();
        if (e->TypeInfo() != "EdgeImu") continue;
        vio_preint pre;
        const int k = index_of(w.pose, e->verticies_[0]);
        if (!vio_shim_edge_imu(e, &pre) || k < 0 || k >= VIO_WINDOW_SIZE || index_of(w.pose, e->verticies_[2]) != k + 1) {
            std::cerr << "IMU edge outside the window layout" << std::endl;
            return false;
        }
        ok = ok && ABI(set_imu)(c.h, k, &pre) == VIO_OK;
    }
    // the prior as the caller installed it: SetHessianPrior(156) + ExtendHessiansPriorSize(15), or nothing yet
    if (p.err_prior_.rows() > 0) {
        if (p.H_prior_.rows() != PD || p.err_prior_.rows() != PRD || p.Jt_prior_inv_.rows() != PRD || p.b_prior_.rows() != PD) {
            std::cerr << "prior of unexpected size " << p.H_prior_.rows() << "/" << p.err_prior_.rows() << std::endl;
            return false;
        }
        std::vector<double> H((size_t)PRD * PRD), J((size_t)PRD * PRD), b(PRD), err(PRD);
        for (int i = 0; i < PRD; ++i) {
            b[i] = p.b_prior_[i]; err[i] = p.err_prior_[i];
            for (int j = 0; j < PRD; ++j) { H[(size_t)i * PRD + j] = p.H_prior_(i, j); J[(size_t)i * PRD + j] = p.Jt_prior_inv_(i, j); }
        }
        ok = ok && ABI(set_prior)(c.h, PRD, H.data(), b.data(), err.data(), J.data()) == VIO_OK;
    } else ok = ok && ABI(set_prior)(c.h, 0, nullptr, nullptr, nullptr, nullptr) == VIO_OK;
    if (!ok) std::cerr << "backend rejected the graph: " << ABI(last_error)(c.h) << std::endl;
    return ok;
}

}  // namespace

Problem::Problem(ProblemType problemType) : problemType_(problemType) {}

Problem::~Problem() { global_vertex_id = 0; }      // the reference restarts vertex ids with every Problem (problem.cc:38-41)

// Pose-type vertices grow the prior by their local dimension (problem.cc:51-55,71-81): a fresh Problem holds a 171 x 171
// zero prior before SetHessianPrior overwrites it.
bool Problem::AddVertex(std::shared_ptr<Vertex> vertex) {
    if (!verticies_.emplace(vertex->Id(), vertex).second) return false;
    if (problemType_ == ProblemType::SLAM_PROBLEM && is_pose_type(vertex)) ExtendHessiansPriorSize(vertex->LocalDimension());
    return true;
}

bool Problem::AddEdge(std::shared_ptr<Edge> edge) {
    if (!edges_.emplace(edge->Id(), edge).second) return false;
    for (auto &v : edge->Verticies()) vertexToEdge_.emplace(v->Id(), edge);
    return true;
}

void Problem::ExtendHessiansPriorSize(int dim) {
    const int n = (int)H_prior_.rows() + dim;
    MatXX H = MatXX::Zero(n, n);
    VecX b = VecX::Zero(n);
    H.topLeftCorner(H_prior_.rows(), H_prior_.cols()) = H_prior_;
    b.head(b_prior_.rows()) = b_prior_;
    H_prior_ = H;
    b_prior_ = b;
}

bool Problem::Solve(int iterations) {
    if (edges_.empty() || verticies_.empty()) {
        std::cerr << "\nCannot solve problem without edges or verticies" << std::endl;
        return false;
    }
    if (problemType_ != ProblemType::SLAM_PROBLEM) { std::cerr << "problem_hip.cc: only SLAM_PROBLEM is supported" << std::endl; return false; }
    FlatWindow w = flatten(*this);
    if (!w.ok) { std::cerr << "problem_hip.cc: " << w.why << std::endl; return false; }
    Ctx &c = shared_ctx();
    if (!upload(*this, w, c)) return false;
    vio_solve_report rep;
    const vio_status st = ABI(solve)(c.h, iterations, &rep);
    if (st != VIO_OK && st != VIO_ERR_NOT_FINITE) { std::cerr << "vio_solve: " << ABI(last_error)(c.h) << std::endl; return false; }
    double poses[NF * 7], sbs[NF * 9], ext[7];
    const int dim = w.xyz ? 3 : 1;
    std::vector<double> lmv(w.lm.size() * dim);
    if (ABI(get_window)(c.h, poses, sbs, ext) != VIO_OK) return false;
    if ((w.xyz ? ABI(get_landmarks_xyz)(c.h, (int64_t)w.lm.size(), lmv.data()) : ABI(get_landmarks)(c.h, (int64_t)lmv.size(), lmv.data())) != VIO_OK) return false;
    for (int i = 0; i < NF; ++i) {
        for (int k = 0; k < 7; ++k) w.pose[i]->parameters_[k] = poses[7 * i + k];
        for (int k = 0; k < 9; ++k) w.sb[i]->parameters_[k] = sbs[9 * i + k];
    }
    if (!w.xyz) for (int k = 0; k < 7; ++k) w.ext->parameters_[k] = ext[k];      // (XYZ edges do not name the extrinsic vertex: it gets no update)
    for (size_t l = 0; l < w.lm.size(); ++l) for (int k = 0; k < dim; ++k) w.lm[l]->parameters_[k] = lmv[dim * l + k];
    if (err_prior_.rows() > 0) {                    // b_prior_ / err_prior_ after the first-order updates (estimator.cpp:1040-1049)
        double b[VIO_POSE_DIM], err[VIO_PRIOR_DIM];
        if (ABI(get_prior)(c.h, b, err) != VIO_OK) return false;
        for (int i = 0; i < PD; ++i) b_prior_[i] = b[i];
        for (int i = 0; i < PRD; ++i) err_prior_[i] = err[i];
    }
    std::cout << "problem solve cost: " << rep.solve_ms << " ms" << std::endl;
    std::cout << "   makeHessian cost: " << rep.hessian_ms << " ms" << std::endl;
    return true;
}

// Marginalize(margVertexs, pose_dim) as MargOldFrame / MargNewFrame call it: margVertexs = (pose_k, speed-bias_k) of the
// oldest (k = 0) or the second-newest (k = 9) frame (estimator.cpp:810-812,883-885)
bool Problem::Marginalize(const std::vector<std::shared_ptr<Vertex>> margVertexs, int /*pose_dim*/) {
    FlatWindow w = flatten(*this);
    if (!w.ok) { std::cerr << "problem_hip.cc: " << w.why << std::endl; return false; }
    const int k = margVertexs.empty() ? -1 : index_of(w.pose, margVertexs[0]);
    if (k != 0 && k != VIO_WINDOW_SIZE - 1) { std::cerr << "problem_hip.cc: Marginalize of frame " << k << " is not a window operation" << std::endl; return false; }
    Ctx &c = shared_ctx();
    if (!upload(*this, w, c)) return false;
    std::vector<double> H((size_t)PRD * PRD), J((size_t)PRD * PRD), b(PRD), err(PRD);
    // (VIO_ERR_NOT_FINITE: a landmark block without an inverse — the backend has filled the outputs with what the reference's
    // Marginalize leaves in that case, H_prior_ = 0 and NaN elsewhere, and the reference returns true: so does this)
    const vio_status mst = ABI(marginalize)(c.h, k == 0 ? VIO_MARG_OLD : VIO_MARG_SECOND_NEW, H.data(), b.data(), err.data(), J.data());
    if (mst != VIO_OK && mst != VIO_ERR_NOT_FINITE) {
        std::cerr << "vio_marginalize: " << ABI(last_error)(c.h) << std::endl;
        return false;
    }
    H_prior_ = MatXX::Zero(PRD, PRD); Jt_prior_inv_ = MatXX::Zero(PRD, PRD);
    b_prior_ = VecX::Zero(PRD); err_prior_ = VecX::Zero(PRD);
    for (int i = 0; i < PRD; ++i) {
        b_prior_[i] = b[i]; err_prior_[i] = err[i];
        for (int j = 0; j < PRD; ++j) { H_prior_(i, j) = H[(size_t)i * PRD + j]; Jt_prior_inv_(i, j) = J[(size_t)i * PRD + j]; }
    }
    return true;
}

}  // namespace backend
}  // namespace myslam
