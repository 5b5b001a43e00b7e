#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libvio_ref.so).

Run only in the container that mounts /root/reference:   python tests/golden/make_golden.py
Every file holds the inputs (flat arrays, exactly what the C ABI takes) and the reference's outputs for them.
The fixtures are data; no reference source travels with them.
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
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def window_case(name, w, ext_fixed=1, keep_matrix=True, marg=(), solve=True):
    d = tu.window_to_arrays(w)
    d["cfg_ext_fixed"] = np.int32(ext_fixed)
    ctx = ref.context(ext_fixed=ext_fixed)
    ctx.load(w)
    step = tu.run_stepwise(ctx)
    if not keep_matrix:
        step.pop("Hs")
    d.update({"step_" + k: v for k, v in step.items()})
    if solve:
        ctx2 = ref.context(ext_fixed=ext_fixed)
        ctx2.load(w)
        sol, _ = tu.run_solve(ctx2, 10)
        d.update({"solve_" + k: v for k, v in sol.items()})
        ws = w.copy()
        ws.poses, ws.speed_bias, ws.ext, ws.inv_depth = sol["posesF"], sol["sbF"], sol["extF"], sol["invdF"]
        if w.prior is not None:     # estimator.cpp:1040-1049: b/err prior come back updated, H/Jt stay
            ws.prior = dict(w.prior)
            ws.prior["b"] = sol["bpriorF"][:156].copy()
            ws.prior["err"] = sol["errpriorF"].copy()
        for kind in marg:
            ctx3 = ref.context(ext_fixed=ext_fixed)
            ctx3.load(ws)
            m = ctx3.marginalize(kind)
            d.update({"marg%d_%s" % (kind, k): v for k, v in m.items()})
            d.update(tu.window_to_arrays(ws, prefix="marg%d_in_" % kind))
    save(name, **d)
    return d


# ---- whole-window cases ---------------------------------------------------------------------------------
wA = vio.synth.make_window(50, seed=42)
dA = window_case("window_n50_s42", wA, marg=(vio.MARG_OLD,))
wB = vio.synth.make_window(300, seed=43)
window_case("window_n300_s43", wB, keep_matrix=False)
wC = vio.synth.make_window(120, seed=44, ragged=True)
window_case("window_n120_s44_ragged_extfree", wC, ext_fixed=0, marg=(vio.MARG_OLD,))
# a window that carries a prior: the prior is the reference's own MargOldFrame output of the preceding window
prior = {k: dA["marg0_" + k] for k in tu.PRIOR_FIELDS}
wD = vio.synth.make_window(300, seed=45, t0=1.1)
wD.prior = prior
window_case("window_n300_s45_prior", wD, marg=(vio.MARG_OLD, vio.MARG_SECOND_NEW))
# small initial errors: with Tukey(1.0) every edge whose whitened residual exceeds 1 gets weight 0, and a landmark
# whose edges all have weight 0 makes the reference divide by h_ll = 0
wE = vio.synth.make_window(200, seed=46, outlier_fraction=0.05, pos_noise=0.001, rot_noise=0.0002, depth_noise=0.003,
                          pixel_noise=0.25 / 460.0)
for loss, nm in ((vio.LOSS_HUBER, "huber"), (vio.LOSS_TUKEY, "tukey"), (vio.LOSS_TRIVIAL, "trivial")):
    d = tu.window_to_arrays(wE)
    ctx = ref.context(loss_type=loss)
    ctx.load(wE)
    step = tu.run_stepwise(ctx)
    step.pop("Hs")
    d.update({"step_" + k: v for k, v in step.items()})
    d["cfg_loss_type"] = np.int32(loss)
    save("window_n200_s46_" + nm, **d)
# delta_x only at N = 2000 (the dense reference needs 38 MB here)
wF = vio.synth.make_window(2000, seed=42)
ctx = ref.context()
ctx.load(wF)
ctx.linearize()
chi0, lam0 = ctx.init_lm()
ctx.solve_linear(lam0)
dxp, dxl = ctx.get_delta()
d = tu.window_to_arrays(wF)
d.update(step_chi0=np.float64(chi0), step_lambda0=np.float64(lam0), step_dx_pose=dxp, step_dx_lm=dxl)
save("window_n2000_s42_dx", **d)

# ---- per-function vectors -------------------------------------------------------------------------------
rng = np.random.RandomState(7)
n = 64
w = vio.synth.make_window(n, seed=9, ragged=True)
pi_, pj_, ext_ = w.poses[w.host], w.poses[w.target], np.tile(w.ext, (w.lm.size, 1))
res, Jl, Ji, Jj, Je = (np.zeros((w.lm.size, k)) for k in (2, 2, 12, 12, 12))
f = dll.vior_reproj_edge
f.restype = None
for e in range(w.lm.size):
    a, b, c = np.ascontiguousarray(pi_[e]), np.ascontiguousarray(pj_[e]), np.ascontiguousarray(ext_[e])
    p, q = np.ascontiguousarray(w.pts_i[e]), np.ascontiguousarray(w.pts_j[e])
    f(dp(a), dp(b), dp(c), C.c_double(w.inv_depth[w.lm[e]]), dp(p), dp(q), dp(res[e]), dp(Jl[e]), dp(Ji[e]), dp(Jj[e]), dp(Je[e]))
save("reproj_edges", pose_i=pi_, pose_j=pj_, ext=ext_, inv_depth=w.inv_depth[w.lm], pts_i=w.pts_i, pts_j=w.pts_j,
     residual=res, J_lambda=Jl, J_pose_i=Ji, J_pose_j=Jj, J_ext=Je)

e2 = np.concatenate([[0.0, 1e-12, 0.5, 1.0, 1.0 + 1e-9, 2.0], rng.uniform(0, 30, 58)])
loss = {}
fl = dll.vior_loss
fl.restype = None
for t, nm in ((1, "huber"), (2, "cauchy"), (3, "tukey")):
    for delta in (1.0, 2.5):
        out = np.zeros((e2.size, 3))
        for i, v in enumerate(e2):
            fl(C.c_int(t), C.c_double(delta), C.c_double(v), dp(out[i]))
        loss["%s_%g" % (nm, delta)] = out
rr = rng.normal(0, 0.004, size=(64, 2))
rr[0] = 0
fr = dll.vior_robust_info2
fr.restype = None
rob = {}
for t, nm in ((0, "trivial"), (1, "huber"), (2, "cauchy"), (3, "tukey")):
    W, dr = np.zeros((64, 4)), np.zeros(64)
    for i in range(64):
        d_ = C.c_double()
        fr(C.c_int(t), C.c_double(1.0), C.c_double(460 / 1.5), dp(np.ascontiguousarray(rr[i])), C.byref(d_), dp(W[i]))
        dr[i] = d_.value
    rob["W_" + nm], rob["drho_" + nm] = W, dr
save("loss_and_robust", e2=e2, residuals=rr, **loss, **rob)

poses = w.poses[rng.randint(0, 11, 32)].copy()
deltas = rng.normal(0, 0.05, size=(32, 6))
deltas[0] = 0
deltas[1, 3:] = 1e-12
deltas[2, 3:] = [3.0, -2.0, 1.0]
outp = poses.copy()
fp = dll.vior_pose_plus
fp.restype = None
for i in range(32):
    fp(dp(outp[i]), dp(np.ascontiguousarray(deltas[i])))
save("pose_plus", poses=poses, deltas=deltas, result=outp)

Hs, bs, lam0 = dA["step_Hs"], dA["step_bs"], float(dA["step_lambda0"])
fs = dll.vior_ldlt_solve
fs.restype = None
cases = {}
for i, lam in enumerate((lam0, 1e3, 1.0)):
    A = Hs + lam * np.eye(171)
    x, tr = np.zeros(171), np.zeros(171, dtype=np.int32)
    fs(C.c_int(171), dp(np.ascontiguousarray(A)), dp(np.ascontiguousarray(bs)), dp(x), tr.ctypes.data_as(C.POINTER(C.c_int)))
    cases["lambda_%d" % i], cases["x_%d" % i], cases["tr_%d" % i] = np.float64(lam), x, tr
Mr = rng.normal(size=(24, 24))
Asmall = Mr @ Mr.T + np.diag(rng.uniform(0, 5, 24))
Asmall[5, :] = 0
Asmall[:, 5] = 0      # an exactly zero pivot row, as the fixed extrinsic produces
bsmall = rng.normal(size=24)
bsmall[5] = 0
x, tr = np.zeros(24), np.zeros(24, dtype=np.int32)
fs(C.c_int(24), dp(Asmall), dp(bsmall), dp(x), tr.ctypes.data_as(C.POINTER(C.c_int)))
save("ldlt", Hs=Hs, bs=bs, A_small=Asmall, b_small=bsmall, x_small=x, tr_small=tr, **cases)

fe = dll.vior_symmetric_eigen
ev, V = np.zeros(24), np.zeros((24, 24))
fe(C.c_int(24), dp(Asmall), dp(ev), dp(V))
Hp = dA["marg0_H"]
ev2, V2 = np.zeros(156), np.zeros((156, 156))
fe(C.c_int(156), dp(np.ascontiguousarray(Hp)), dp(ev2), dp(V2))
save("symmetric_eigen", A_small=Asmall, evals_small=ev, A_prior=Hp, evals_prior=ev2)

fi = dll.vior_inverse15
fi.restype = None
covs = np.stack([np.asarray(p["covariance"]).reshape(15, 15) for p in wA.preint])
infos = np.zeros_like(covs)
for k in range(covs.shape[0]):
    fi(dp(np.ascontiguousarray(covs[k])), dp(infos[k]))
save("inverse15", cov=covs, info=infos)
print("done")
