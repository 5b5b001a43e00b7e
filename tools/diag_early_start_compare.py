#!/usr/bin/env python3
"""k_pose_solve_c's two prologues — the chain's first level started without a barrier (default) and the copy-in with its barriers
(VIO_NO_EARLY_START=1) — must give the same bits: the GN loop, the stepwise path and Solve on windows with and without a prior, ragged tracks,
a free extrinsic.  Prints a digest of everything the solves leave; tests/test_gpu_early_start.py runs it once in each mode (the switch is read
once per process) and compares."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
h = hashlib.sha256()


def feed(*arrays):
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())


c0 = hip.context()
c0.load(vio.synth.make_window(300, seed=41, t0=0.9))
c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
for n, seed, with_prior, ragged, ext_fixed in ((500, 17, False, False, 1), (700, 42, True, True, 1), (2000, 5, True, False, 0), (20000, 42, True, False, 1)):
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    if with_prior:
        w.prior = prior
    a = hip.context(ext_fixed=ext_fixed)
    a.load(w)
    for it in range(5):
        a.gn_iteration(5e5 / (1 + it))
    p, s, e = a.get_window()
    feed(p, s, e, a.get_landmarks(), [a.chi2()], *a.get_prior())
    b = hip.context(ext_fixed=ext_fixed)
    b.load(w)
    b.linearize()
    for lam in (5e5, 1e3):
        b.solve_linear(lam)
        feed(b.get_delta()[0])
    b.update_states()
    p, s, e = b.get_window()
    feed(p, s, e, b.get_landmarks())
    c = hip.context(ext_fixed=ext_fixed)
    c.load(w)
    rep = c.solve(6)
    p, s, e = c.get_window()
    feed(p, s, e, c.get_landmarks(), [rep.final_chi2, rep.final_lambda, rep.iterations, rep.trials])
# vio_solve's loop in the reference's real regime: ragged tracks, chained priors — windows on which trials are rejected (a rejected step makes
# k_pose_solve_c solve the other set again: the second copy of either prologue)
cs = hip.context()
pr, trials = None, []
for r in range(6):
    w = vio.synth.make_window(150, seed=300 + r, t0=1.0 + 0.1 * r, ragged=True)
    w.prior = pr
    cs.load(w)
    rep = cs.solve(10)
    trials.append(rep.trials)
    p, s, e = cs.get_window()
    feed(p, s, e, cs.get_landmarks(), [rep.final_chi2, rep.final_lambda, rep.iterations, rep.trials], rep.chi2_trace[:rep.iterations + 1], rep.lambda_trace[:rep.iterations + 1])
    pr = cs.marginalize(vio.MARG_OLD)
    feed(pr["H"], pr["b"], pr["err"], pr["jt_inv"])
assert max(trials) > 10, trials        # (some step was rejected: the case is exercised)
print("digest", h.hexdigest())
