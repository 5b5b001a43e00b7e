#!/usr/bin/env python3
"""tests/golden/ldlt_exact.npz: the solutions of (H + lambda I) x = b of tests/golden/ldlt.npz — the reference's own reduced
system, at lambda = lambda_0, 1e3, 1 — computed with 50 significant digits (mpmath LU) and rounded to double.

Why: the vectors ldlt.npz holds are Eigen::LDLT's (problem.cc:439), i.e. ONE double-precision solver's rounding of an ill-scaled
system (entries from 1e16 down to 1e3; cond 3e10 / 1.6e13 / 2.7e16).  A second backward-stable solver — k_pose_solve's blocked
LDL^T on the matrix cores — lands elsewhere inside the same rounding ball.  With the exact solution beside them the tests can ask
the question that has an answer: is the HIP solve as close to the truth as Eigen's is?  (tools/diag_parity_exact.py, profiles/
parity_exact.json: it is, on every golden window and lambda, with the smaller backward error.)

Needs only ldlt.npz and mpmath; no reference code involved.   python tests/golden/make_golden_ldlt_exact.py
"""
import os

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 50
z = np.load(os.path.join(HERE, "ldlt.npz"))
out = {}
for i in range(3):
    lam = float(z["lambda_%d" % i])
    A = mp.matrix(z["Hs"].tolist())
    for k in range(171):
        A[k, k] += mp.mpf(lam)
    x = mp.lu_solve(A, mp.matrix(z["bs"].tolist()))
    out["lambda_%d" % i] = np.float64(lam)
    out["x_exact_%d" % i] = np.array([float(v) for v in x])
    print("lambda %.3g: |x_eigen - x_exact|_inf = %.3e" % (lam, np.abs(z["x_%d" % i] - out["x_exact_%d" % i]).max()))
np.savez_compressed(os.path.join(HERE, "ldlt_exact.npz"), **out)
