#!/usr/bin/env python3
"""Generates the XYZ-landmark fixtures tests/golden/window_xyz_*.npz, marg0_xyz_*.npz, reproj_xyz_edges.npz and inverse3.npz from the
COMPILED REFERENCE (oracle/_ref/libvio_ref.so): VertexPointXYZ (vertex_point_xyz.h) + EdgeReprojectionXYZ
(edge_reprojection.cc:130-180) inside the reference's own Problem, graphs built by oracle/ref_harness.cpp.

Run only in the container that mounts /root/reference:   python tests/golden/make_golden_xyz.py
Inputs and the reference's outputs only; no reference source travels with them.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
import vio_testutil as tu  # noqa: E402

vio = load_package()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
ref = vio.VioLib(os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so"), "vior_")
dll = ref.dll
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-44s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def window_case(name, w, ext_fixed=1, loss=None, keep_matrix=False, solve=True, marg_new=False):
    d = tu.window_to_arrays(w)
    kw = dict(ext_fixed=ext_fixed)
    d["cfg_ext_fixed"] = np.int32(ext_fixed)
    if loss is not None:
        kw["loss_type"] = loss
        d["cfg_loss_type"] = np.int32(loss)
    ctx = ref.context(**kw)
    ctx.load(w)
    step = tu.run_stepwise(ctx)
    if not keep_matrix:
        step.pop("Hs")
    d.update({"step_" + k: v for k, v in step.items()})
    if solve:
        ctx2 = ref.context(**kw)
        ctx2.load(w)
        sol, _ = tu.run_solve(ctx2, 10)
        d.update({"solve_" + k: v for k, v in sol.items()})
        if marg_new:        # MargNewFrame is edge-free: defined for any kind of landmark (estimator.cpp:830-901)
            ws = w.copy()
            ws.poses, ws.speed_bias, ws.ext, ws.xyz = sol["posesF"], sol["sbF"], sol["extF"], sol["invdF"]
            ws.prior = dict(w.prior)
            ws.prior["b"] = sol["bpriorF"][:156].copy()
            ws.prior["err"] = sol["errpriorF"].copy()
            ctx3 = ref.context(**kw)
            ctx3.load(ws)
            m = ctx3.marginalize(vio.MARG_SECOND_NEW)
            d.update({"marg1_" + k: v for k, v in m.items()})
            d.update(tu.window_to_arrays(ws, prefix="marg1_in_"))
    save(name, **d)
    return d


window_case("window_xyz_n50_s51", vio.synth.make_window_xyz(50, seed=51), keep_matrix=True)
window_case("window_xyz_n300_s52_ragged_extfree", vio.synth.make_window_xyz(300, seed=52, ragged=True), ext_fixed=0)
# a prior: the reference's MargOldFrame output of an inverse-depth window one frame earlier (tests/golden/window_n50_s42.npz)
zA = np.load(os.path.join(HERE, "window_n50_s42.npz"))
wP = vio.synth.make_window_xyz(300, seed=53, t0=1.1)
wP.prior = {k: zA["marg0_" + k] for k in tu.PRIOR_FIELDS}
window_case("window_xyz_n300_s53_prior", wP, marg_new=True)
wL = vio.synth.make_window_xyz(200, seed=54, outlier_fraction=0.05, pos_noise=0.001, rot_noise=0.0002, pixel_noise=0.25 / 460.0,
                               xyz_noise=0.003)
for loss, nm in ((vio.LOSS_TUKEY, "tukey"), (vio.LOSS_TRIVIAL, "trivial")):
    window_case("window_xyz_n200_s54_" + nm, wL, loss=loss, solve=False)
# Problem::Marginalize on an XYZ graph (generic over the landmark dimension, problem.cc:617-795; no caller in Estimator): only the
# edges connected to pose 0 enter (:621), so every landmark seen from frame 0 has ONE observation and a 3x3 block of rank 2, which
# the reference inverts all the same (:697-700).  What comes out is the marginalisation of the graph WITHOUT reprojection edges
# (a single view of a free point says nothing about the pose) plus whatever rounding leaves of H_pl H_ll^-1 H_lp - H_pp(direct) —
# noise of the size of the visual information itself — or NaN when an elimination meets an exact zero pivot (about one landmark
# in eight).  Fixtures: two finite cases with the reference's own edge-free result beside them (the size of its noise), and one
# whose frame-0 observation of a landmark is an outlier beyond Tukey's delta: weight exactly 0, H_ll exactly 0, NaN by construction.
def edge_free(w):
    w0 = w.copy()
    w0.xyz, w0.xyz_gt = np.zeros((0, 3)), np.zeros((0, 3))
    w0.lm, w0.frame, w0.pts = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2))
    w0.n_landmarks = w0.n_observations = 0
    return w0


def marg0_case(name, w, **kw):
    d = tu.window_to_arrays(w)
    for k, v in kw.items():
        d["cfg_" + k] = np.int32(v) if isinstance(v, int) else np.float64(v)
    ctx = ref.context(**kw)
    ctx.load(w)
    m = ctx.marginalize(vio.MARG_OLD, allow_nonfinite=True)
    d.update({"marg0_" + k: v for k, v in m.items()})
    ctx = ref.context(**kw)
    ctx.load(edge_free(w))
    e = ctx.marginalize(vio.MARG_OLD)
    d.update({"marg0_free_" + k: e[k] for k in ("H", "b")})
    d["marg0_finite"] = np.int32(np.isfinite(m["b"]).all())
    save(name, **d)
    return m, e


prior_A = {k: zA["marg0_" + k] for k in tu.PRIOR_FIELDS}
n_fin = 0
for seed in range(61, 100):
    w = vio.synth.make_window_xyz(20, seed=seed, t0=1.1)
    w.prior = prior_A
    c_ = ref.context()
    c_.load(w)
    if not np.isfinite(c_.marginalize(vio.MARG_OLD, allow_nonfinite=True)["b"]).all():
        continue
    m, e = marg0_case("marg0_xyz_n20_s%d" % seed, w)
    print("   frame-0 landmarks %d, |H - H_edge_free|max %.3g (prior max %.3g)" % (len(np.unique(w.lm[w.frame == 0])), np.abs(m["H"] - e["H"]).max(), np.abs(e["H"]).max()))
    n_fin += 1
    if n_fin == 4:
        break
wN = vio.synth.make_window_xyz(40, seed=71, t0=1.1)
wN.prior = prior_A
first0 = int(np.flatnonzero(wN.frame == 0)[0])
wN.pts[first0] += np.array([0.3, -0.2])            # e = s |r| ~ 100 >> delta = 1: Tukey's rho' is exactly 0 out there
m, _ = marg0_case("marg0_xyz_n40_s71_tukey_nan", wN, loss_type=vio.LOSS_TUKEY)
assert not np.isfinite(m["b"]).any() and (m["H"] == 0).all()

# delta_x only at N = 2000 (the dense reference needs (171 + 6000)^2 doubles = 305 MB here)
wF = vio.synth.make_window_xyz(2000, seed=42)
ctx = ref.context()
ctx.load(wF)
ctx.linearize()
chi0, lam0 = ctx.init_lm()
ctx.solve_linear(lam0)
dxp, dxl = ctx.get_delta()
d = tu.window_to_arrays(wF)
d.update(step_chi0=np.float64(chi0), step_lambda0=np.float64(lam0), step_dx_pose=dxp, step_dx_lm=dxl)
save("window_xyz_n2000_s42_dx", **d)

# ---- per-function vectors
w = vio.synth.make_window_xyz(64, seed=9, ragged=True)
M = w.lm.size
res, Jf, Jp = np.zeros((M, 2)), np.zeros((M, 6)), np.zeros((M, 12))
f = dll.vior_reproj_xyz_edge
f.restype = None
pose_e, pw_e = w.poses[w.frame], w.xyz[w.lm]
for e in range(M):
    f(dp(np.ascontiguousarray(pose_e[e])), dp(np.ascontiguousarray(w.ext)), dp(np.ascontiguousarray(pw_e[e])),
      dp(np.ascontiguousarray(w.pts[e])), dp(res[e]), dp(Jf[e]), dp(Jp[e]))
save("reproj_xyz_edges", pose=pose_e, ext=w.ext, pw=pw_e, obs=w.pts, residual=res, J_feature=Jf, J_pose=Jp)

# Hmm.block().inverse() on 3x3 blocks: the H_ll of the N = 50 window, and matrices that exercise every pivot choice
z50 = np.load(os.path.join(HERE, "window_xyz_n50_s51.npz"))
rng = np.random.RandomState(3)
mats = [h for h in z50["step_hll"]]
for _ in range(30):
    a = rng.normal(size=(3, 3))
    mats.append(a @ a.T + 1e-3 * np.eye(3))
mats += [np.array([[1e-3, 2.0, 0.5], [2.0, 1.0, 0.3], [0.5, 0.3, 4.0]]),        # pivot 0 <- row 1
         np.array([[1e-3, 0.2, 5.0], [0.2, 1.0, 0.3], [5.0, 0.3, 4.0]]),        # pivot 0 <- row 2
         np.array([[4.0, 1.0, 1.0], [1.0, 0.26, 3.0], [1.0, 3.0, 2.0]]),        # pivot 1 <- row 2
         np.diag([3.0, 0.5, 7.0])]
mats = np.stack(mats)
inv = np.zeros_like(mats)
fi = dll.vior_inverse3
fi.restype = None
for k in range(mats.shape[0]):
    fi(dp(np.ascontiguousarray(mats[k])), dp(inv[k]))
save("inverse3", A=mats, Ainv=inv)
print("done")
