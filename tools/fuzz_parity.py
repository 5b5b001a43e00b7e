#!/usr/bin/env python3
"""Randomised parity sweep (diagnostic; the fixed cases live in tests/): HIP against the oracle on windows of random size,
track structure, loss, extrinsic mode and prior; one stepwise LM step, a few GN iterations and Solve(10) each.
  python tools/fuzz_parity.py [cases] [seed]          (VIO_FUZZ_ONLY=<case>: that case alone, in detail; VIO_DEFAULT_ITEM_POLICY=1: throughput-policy contexts)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402
import vio_testutil as tu  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = int(os.environ.get("VIO_FUZZ_ONLY", "-1"))          # one case of the sequence, with more detail
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
w0 = vio.synth.make_window(300, seed=41, t0=0.9)
c0 = orc.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
worst = {"dx": 0.0, "state": 0.0, "gn": 0.0, "marg": 0.0}
for case in range(cases):
    n = int(rng.choice([1, 7, 33, 200, 777, 2500, 6000, 11000]))
    ragged = bool(rng.randint(2))
    k_obs = int(rng.randint(1, 11))
    ext_fixed = int(rng.randint(2))
    loss = int(rng.choice([0, 2, 2, 2, 3]))
    with_prior = bool(rng.randint(2))
    kw = dict(pos_noise=0.001, rot_noise=0.0002, depth_noise=0.003, pixel_noise=0.25 / 460, outlier_fraction=0.05) if loss == 3 else {}
    w = vio.synth.make_window(n, seed=1000 + case, ragged=ragged, obs_per_landmark=k_obs, **kw)
    if with_prior:
        w.prior = prior
    drop_obs, drop_imu = rng.rand() < 0.3, rng.rand() < 0.3
    if drop_obs and w.n_observations > 4:          # some landmarks lose observations
        keep = rng.rand(w.n_observations) > 0.2
        _, first = np.unique(w.lm, return_index=True)
        keep[first] = True                         # (a landmark without any edge is rejected at the boundary: VIO_ERR_UNSUPPORTED)
        w.lm, w.host, w.target = w.lm[keep].copy(), w.host[keep].copy(), w.target[keep].copy()
        w.pts_i, w.pts_j = w.pts_i[keep].copy(), w.pts_j[keep].copy()
        w.n_observations = int(keep.sum())
    if drop_imu:                                   # estimator.cpp:959-960 skips an IMU edge with sum_dt > 10
        w.preint = list(w.preint)
        for k in rng.choice(10, size=int(rng.randint(1, 4)), replace=False):
            w.preint[int(k)] = None
    if only >= 0 and case != only:
        continue
    ch, co = hip.context(ext_fixed=ext_fixed, loss_type=loss), orc.context(ext_fixed=ext_fixed, loss_type=loss)
    ch.load(w); co.load(w)
    if only >= 0 and not os.environ.get("VIO_FUZZ_NODETAIL"):                                   # one case in detail: the system, then the GN loop iteration by iteration
        ch.linearize(); co.linearize()
        (Hh, bh), (Ho, bo) = ch.get_schur_system(), co.get_schur_system()
        print("system: H %.2e b %.2e (relative to the largest entry)" % (np.abs(Hh - Ho).max() / np.abs(Ho).max(), np.abs(bh - bo).max() / np.abs(bo).max()))
        _, lam_d = ch.init_lm(); co.init_lm()
        for it in range(4):
            ch.gn_iteration(lam_d); co.gn_iteration(lam_d)
            print("GN iteration %d: poses %.2e landmarks %.2e chi2 %.9g / %.9g" % (it, np.abs(ch.get_window()[0] - co.get_window()[0]).max(),
                  np.abs(ch.get_landmarks() - co.get_landmarks()).max(), ch.chi2(), co.chi2()))
        ch.load(w); co.load(w)
    a, b = tu.run_stepwise(ch), tu.run_stepwise(co)
    if not np.isfinite(b["dx_pose"]).all():       # a landmark without information: NaN in the reference, the oracle and here
        okd = not np.isfinite(a["dx_pose"]).all()
        try:
            ch.load(w); ch.solve(10); okd = False
        except vio.VioError:
            pass
        print("%s case %2d: n=%5d ragged=%d K=%2d ext_fixed=%d loss=%d prior=%d | degenerate (non-finite in both, error status from vio_solve)"
              % ("ok  " if okd else "FAIL", case, n, ragged, k_obs, ext_fixed, loss, with_prior))
        continue
    dx = max(np.abs(a["dx_pose"] - b["dx_pose"]).max(), np.abs(a["dx_lm"] - b["dx_lm"]).max() if n else 0.0)
    ok = dx <= 1e-8 and int(a["accepted"]) == int(b["accepted"]) and a["lambda0"] == b["lambda0"]
    ch.load(w); co.load(w)
    ch.linearize(); _, lam = ch.init_lm()
    for _ in range(3):
        ch.gn_iteration(lam); co.gn_iteration(lam)
        if only >= 0:
            print("   main path GN: nan hip %d nan oracle %d lam %g" % (np.isnan(ch.get_window()[0]).sum(), np.isnan(co.get_window()[0]).sum(), lam))
    ph_, po_ = ch.get_window()[0], co.get_window()[0]
    if np.isnan(po_).any() and np.array_equal(np.isnan(ph_), np.isnan(po_)):
        gn = 0.0          # the fixed-lambda loop left the basin on this (tiny, degenerate) window: NaN in the oracle and here, in the same entries
    else:
        gn = np.abs(ph_ - po_).max()
    ok = ok and gn <= 1e-7
    ch.load(w); co.load(w)
    def try_solve(c):
        try:
            return c.solve(10), None
        except vio.VioError as exc:
            return None, str(exc)
    (rh, eh), (ro, eo) = try_solve(ch), try_solve(co)
    if eh or eo:                                   # a landmark lost its last weighted edge on the way: both must say so
        okd = bool(eh) and bool(eo) and ("NOT_FINITE" in eh) and ("NOT_FINITE" in eo)
        okd = okd and np.abs(ch.get_window()[0] - co.get_window()[0]).max() <= 1e-5      # the last accepted states
        print("%s case %2d: n=%5d ragged=%d K=%2d ext_fixed=%d loss=%d prior=%d | solve not finite: hip %r oracle %r"
              % ("ok  " if okd else "FAIL", case, n, ragged, k_obs, ext_fixed, loss, with_prior, eh, eo))
        continue
    st = np.abs(ch.get_window()[0] - co.get_window()[0]).max()
    ok = ok and rh.iterations == ro.iterations and st <= 1e-5
    # MargOldFrame on the same solved state (the oracle's): the invariants of tests/test_oracle_golden.py::check_prior
    # (entry-wise the result is ill-posed: eigenvalue cut at 1e-8, see there)
    ws = w.copy()
    ws.poses, ws.speed_bias, ws.ext = co.get_window()
    ws.inv_depth = co.get_landmarks()
    ch.load(ws); co.load(ws)
    mh, mo = ch.marginalize(vio.MARG_OLD), co.marginalize(vio.MARG_OLD)
    mg = np.abs(mh["H"] - mo["H"]).max() / max(np.abs(mo["H"]).max(), 1e-300)
    # (check_prior's finer invariants are tuned to the golden windows: on arbitrary ones the reference itself misses them,
    # e.g. |H P H - H| = 204 against a bound of 236 on seed 1023, the oracle 757; here: entries and spectrum only)
    ev_h, ev_o = np.linalg.eigvalsh(0.5 * (mh["H"] + mh["H"].T)), np.linalg.eigvalsh(0.5 * (mo["H"] + mo["H"].T))
    # judged only where the problem is posed: with few edges hosted in frame 0 and no prior the marginalised block is the
    # IMU factor with its gauge freedom, eigenvalues sit on the 1e-8 cut and oracle and reference themselves differ by O(1)
    # (one observation per landmark: every landmark block rests on a single edge and the Schur complement cancels all but rounding — 2.2e-4 of the
    #  largest entry between oracle and HIP on seeds 9102 / 1234, n = 6000: not a posed comparison either)
    posed = (with_prior or int((ws.host == 0).sum()) >= 300) and (with_prior or k_obs >= 2)
    okm = np.isfinite(mh["H"]).all() and (not posed or (mg <= 2e-4 and np.abs(ev_h - ev_o).max() <= 2e-4 * ev_o.max()))
    ok = ok and okm
    if not posed:
        mg = 0.0
    if with_prior:                                 # MargNewFrame: the prior alone, frame 9 out (estimator.cpp:830-901)
        nh, no_ = ch.marginalize(vio.MARG_SECOND_NEW), co.marginalize(vio.MARG_SECOND_NEW)
        ok = ok and np.abs(nh["H"] - no_["H"]).max() <= 2e-4 * max(np.abs(no_["H"]).max(), 1e-300)
    if case % 3 == 0:                              # the sharded kernel sequence on one rank (RCCL sums are identities): bit for bit
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sb = vio.sharded.ShardedBackend(hip, w, 0, 1, dist=None, torch_device="cuda", force_hook=True, exchange="native",
                                        ctx_kwargs=dict(ext_fixed=ext_fixed, loss_type=loss))
        plain = hip.context(ext_fixed=ext_fixed, loss_type=loss); plain.load(w)
        rs, rp = try_solve(sb.ctx), try_solve(plain)
        same = (rs[1] is None) == (rp[1] is None) and np.array_equal(sb.ctx.get_window()[0], plain.get_window()[0]) \
            and np.array_equal(sb.ctx.get_landmarks(), plain.get_landmarks())
        sb.ctx.load(w); plain.load(w)
        for _ in range(3):
            sb.gn_iteration(lam); plain.gn_iteration(lam)
        same = same and np.array_equal(sb.ctx.get_window()[0], plain.get_window()[0]) and sb.ctx.chi2() == plain.chi2()
        sb.ctx.comm_destroy()
        if not same:
            ok = False
            print("     sharded-on-one-rank differs from unsharded")
    worst["dx"], worst["state"], worst["gn"], worst["marg"] = max(worst["dx"], dx), max(worst["state"], st), max(worst["gn"], gn), max(worst["marg"], mg)
    print("%s case %2d: n=%5d ragged=%d K=%2d ext_fixed=%d loss=%d prior=%d drop=%d%d | dx %.1e gn %.1e solve %.1e marg %.1e iters %d/%d"
          % ("ok  " if ok else "FAIL", case, n, ragged, k_obs, ext_fixed, loss, with_prior, drop_obs, drop_imu, dx, gn, st, mg, rh.iterations, ro.iterations))
print("worst:", worst)
