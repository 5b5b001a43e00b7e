#!/usr/bin/env python3
"""Problem::Solve on a few windows through the HIP library, the trace printed as JSON (tests/test_gpu_lm_loop.py runs it twice: the
four-launch loop of vio_solve and, with VIO_LM_CLASSIC=1, the trial / re-linearisation slots it replaced, and compares)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
wp = vio.synth.make_window(200, seed=71, t0=0.9)
cp = orc.context()
cp.load(wp)
cp.solve(10)
prior = cp.marginalize(vio.MARG_OLD)
out = []
cases = [("invdepth", 300, 43, False, 10), ("invdepth", 300, 45, True, 10), ("invdepth", 2000, 42, True, 25), ("invdepth", 60, 7, True, 40),
         ("xyz", 300, 52, False, 10), ("xyz", 40, 9, True, 40), ("invdepth", 20000, 100, True, 10)]
for kind, n, seed, with_prior, its in cases:
    w = (vio.synth.make_window_xyz(n, seed=seed, obs_per_landmark=4) if kind == "xyz" else vio.synth.make_window(n, seed=seed, ragged=bool(seed & 1)))
    w.prior = prior if with_prior else None
    c = hip.context(ext_fixed=seed & 1)
    c.load(w)
    r = c.solve(its)
    p, s, e = c.get_window()
    lm = c.get_landmarks_xyz() if kind == "xyz" else c.get_landmarks()
    pr = c.get_prior() if with_prior else (np.zeros(1), np.zeros(1))
    out.append(dict(case=[kind, n, seed, with_prior, its], iterations=r.iterations, trials=r.trials, accepted=r.accepted, stop_reason=r.stop_reason,
                    final_chi2=r.final_chi2, final_lambda=r.final_lambda, initial_chi2=r.initial_chi2,
                    chi2_trace=[float(x) for x in r.chi2_trace[:r.iterations + 1]], lambda_trace=[float(x) for x in r.lambda_trace[:r.iterations + 1]],
                    poses=p.ravel().tolist(), sb=s.ravel().tolist(), lm_head=np.asarray(lm).ravel()[:200].tolist(), lm_norm=float(np.abs(lm).sum()),
                    bprior=np.asarray(pr[0]).ravel().tolist(), errprior=np.asarray(pr[1]).ravel().tolist()))
print("LMTRACE " + json.dumps(out))
