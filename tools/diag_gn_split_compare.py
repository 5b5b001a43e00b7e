#!/usr/bin/env python3
"""The GN loop's split solve (VIO_GN_SPLIT=1: the speed-bias chain eliminated by a workgroup of k_linearize's grid, k_pose_solve_cs behind it;
VIO_GN_SPLIT=2: the same split with the chain in a launch of its own) against the stepwise path — linearize, solve_linear, update — on the
same window: bit for bit, with and without a marginalisation prior, over several iterations (the step owed by one iteration is settled by
the next).  Run by tests/test_gpu_gn_split.py in a process of its own (the mode is read once per process).  Prints 'OK' or the differences."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
bad = 0
for n, seed, with_prior, ragged in ((500, 17, False, False), (700, 42, True, True), (64, 3, True, False)):
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    if with_prior:
        c0 = hip.context()
        c0.load(vio.synth.make_window(300, seed=41, t0=0.9))
        c0.solve(10)
        w.prior = c0.marginalize(vio.MARG_OLD)
    a, b = hip.context(), hip.context()
    a.load(w)
    b.load(w)
    lam = 5e5
    for it in range(4):
        a.gn_iteration(lam)
        b.linearize()
        if it == 0:
            Ha, ba = a.get_schur_system()
            Hb, bb = b.get_schur_system()
            # (with a prior the right-hand side a getter assembles after the step carries b_prior of the NEW state: only H compares)
            if not (np.array_equal(Ha, Hb) and (with_prior or np.array_equal(ba, bb))):
                print("window %d: reduced system differs: H %.3e b %.3e" % (n, np.abs(Ha - Hb).max(), np.abs(ba - bb).max()))
                bad += 1
        b.solve_linear(lam)
        b.update_states()
    pa, sa, _ = a.get_window()
    pb, sb, _ = b.get_window()
    la, lb = a.get_landmarks(), b.get_landmarks()
    pra, prb = a.get_prior(), b.get_prior()
    same = np.array_equal(pa, pb) and np.array_equal(sa, sb) and np.array_equal(la, lb) and np.array_equal(pra[0], prb[0]) and np.array_equal(pra[1], prb[1])
    if not same or a.chi2() != b.chi2():
        print("window %d (prior %s): poses %.3e speed-bias %.3e landmarks %.3e b_prior %.3e err_prior %.3e" % (
            n, with_prior, np.abs(pa - pb).max(), np.abs(sa - sb).max(), np.abs(la - lb).max(), np.abs(pra[0] - prb[0]).max(), np.abs(pra[1] - prb[1]).max()))
        bad += 1
print("OK" if bad == 0 else "FAILED %d" % bad)
sys.exit(1 if bad else 0)
