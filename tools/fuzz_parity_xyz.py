#!/usr/bin/env python3
"""Randomised parity sweep for XYZ-landmark windows (diagnostic; the fixed cases live in tests/test_xyz_landmarks.py): HIP
against the oracle on windows of random size, pattern structure, loss, extrinsic flag and prior; one stepwise LM step, three
GN iterations and Solve(10) (the batched pass of XYZ windows is in tests/test_gpu_batch.py).
  python tools/fuzz_parity_xyz.py [cases] [seed]"""
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
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # one case, with the per-iteration trace of Solve(10)
w0 = vio.synth.make_window(300, seed=41, t0=0.9)
c0 = orc.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
worst = {"dx": 0.0, "state": 0.0, "gn": 0.0}
bad = 0
for case in range(cases):
    n = int(rng.choice([1, 7, 33, 200, 777, 2500, 6000, 11000]))
    ragged = bool(rng.randint(2))
    # Well-posed windows only.  A point seen twice from neighbouring frames (or once, after Tukey has zeroed its other
    # edges) has a 3x3 block of rank 2 up to rounding: its inverse is 1e16 or infinite by the last bit, in the reference as
    # here, the point runs off to 1e14 within the Cauchy loss's flat tail and no two implementations agree (oracle against
    # the compiled reference: poses 1e-4 apart on such windows).  So: at least 3 observations, no Tukey.
    k_obs = int(rng.randint(2, 11))            # + the host observation: 3 .. 11 observations per landmark
    ext_fixed = int(rng.randint(2))
    loss = int(rng.choice([0, 2, 2, 2]))
    with_prior = bool(rng.randint(2))
    kw = dict(pos_noise=0.001, rot_noise=0.0002, pixel_noise=0.25 / 460, outlier_fraction=0.05, xyz_noise=0.003) if loss == 3 else {}
    w = vio.synth.make_window_xyz(n, seed=2000 + case, ragged=ragged, obs_per_landmark=k_obs, **kw)
    if with_prior:
        w.prior = prior
    if ragged:                                         # ragged tracks: drop the landmarks left with fewer than 3 observations
        cnt = np.bincount(w.lm, minlength=w.n_landmarks)
        good = cnt >= 3
        if not good.all() and good.any():
            remap = np.cumsum(good) - 1
            keep = good[w.lm]
            w.xyz, w.xyz_gt = w.xyz[good].copy(), w.xyz_gt[good].copy()
            w.lm, w.frame, w.pts = remap[w.lm[keep]].astype(np.int32), w.frame[keep].copy(), w.pts[keep].copy()
            w.n_landmarks, w.n_observations = int(good.sum()), int(keep.sum())
    if rng.rand() < 0.3 and w.n_observations > 8:      # some landmarks lose observations, never below three
        keep = rng.rand(w.n_observations) > 0.2
        cnt = np.bincount(w.lm, minlength=w.n_landmarks)
        first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        keep[first] = True
        keep[first + 1] = True
        keep[first + 2] = True
        w.lm, w.frame, w.pts = w.lm[keep].copy(), w.frame[keep].copy(), w.pts[keep].copy()
        w.n_observations = int(keep.sum())
    if rng.rand() < 0.3:
        w.preint = list(w.preint)
        for k in rng.choice(10, size=int(rng.randint(1, 4)), replace=False):
            w.preint[int(k)] = None
    if only >= 0 and case != only:
        continue
    if only >= 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_oracle_golden import solve_trace_stepwise
        kw2 = dict(ext_fixed=ext_fixed, loss_type=loss)
        th, to = solve_trace_stepwise(hip, w, kw2), solve_trace_stepwise(orc, w, kw2)
        for i, (x, y) in enumerate(zip(th, to)):
            print("it %2d  hip chi2 %.12e lambda %.6e trials %d | oracle chi2 %.12e lambda %.6e trials %d | state diff %.2e"
                  % (i, x[1], x[2], x[3], y[1], y[2], y[3], np.abs(x[0] - y[0]).max()))
    ch, co = hip.context(ext_fixed=ext_fixed, loss_type=loss), orc.context(ext_fixed=ext_fixed, loss_type=loss)
    ch.load(w); co.load(w)
    a, b = tu.run_stepwise(ch), tu.run_stepwise(co)
    if not np.isfinite(b["dx_pose"]).all():
        okd = not np.isfinite(a["dx_pose"]).all()
        bad += 0 if okd else 1
        print("%s case %2d: degenerate (non-finite in both)" % ("ok  " if okd else "FAIL", case))
        continue
    dx = max(np.abs(a["dx_pose"] - b["dx_pose"]).max(), np.abs(a["dx_lm"] - b["dx_lm"]).max())
    ok = dx <= 1e-8 and int(a["accepted"]) == int(b["accepted"]) and a["lambda0"] == b["lambda0"]
    ch.load(w); co.load(w)
    ch.linearize(); _, lam = ch.init_lm()
    for _ in range(3):
        ch.gn_iteration(lam); co.gn_iteration(lam)
    gn = np.abs(ch.get_window()[0] - co.get_window()[0]).max()
    ok = ok and gn <= 1e-7
    ch.load(w); co.load(w)
    try:
        ro = co.solve(10)
    except vio.VioError as exc:                       # the oracle itself ends non-finite: ill-posed, nothing to compare with
        print("ok   case %2d: ill-posed (oracle: %s) | dx %.1e gn %.1e" % (case, str(exc)[:40], dx, gn))
        bad += 0 if ok else 1
        continue
    if np.abs(co.get_landmarks_xyz()).max() > 1e6:    # a point ran off (rank-deficient 3x3 block): LM amplifies the last bit
        print("ok   case %2d: ill-posed (a landmark ran off to %.1e in the oracle) | dx %.1e gn %.1e" % (case, np.abs(co.get_landmarks_xyz()).max(), dx, gn))
        bad += 0 if ok else 1
        continue
    try:
        rh = ch.solve(10)
        st = np.abs(ch.get_window()[0] - co.get_window()[0]).max()
        ok = ok and rh.iterations == ro.iterations and st <= 1e-5 and abs(rh.final_chi2 - ro.final_chi2) <= 1e-6 * max(ro.final_chi2, 1.0)
        its = "%d/%d" % (rh.iterations, ro.iterations)
    except vio.VioError as exc:
        st, ok, its = float("nan"), False, str(exc)[:60]
    worst["dx"], worst["state"], worst["gn"] = max(worst["dx"], dx), max(worst["state"], st), max(worst["gn"], gn)
    bad += 0 if ok else 1
    print("%s case %2d: n=%5d ragged=%d K=%2d ext_fixed=%d loss=%d prior=%d | dx %.1e gn %.1e solve %.1e iters %s"
          % ("ok  " if ok else "FAIL", case, n, ragged, k_obs + 1, ext_fixed, loss, with_prior, dx, gn, st, its))
print("worst:", worst, "failures:", bad)
