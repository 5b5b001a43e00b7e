#!/usr/bin/env python3
"""How far are the HIP solve and Eigen's LDLT from the EXACT solution of the same damped system?

Input: the arrays tools/diag_parity_split.py dumps on the GPU box (VIO_PARITY_SPLIT_DUMP=...npz): the HIP path's reduced system
(H, b), and per lambda the solution of the HIP kernel (x_hh) and of Eigen's LDLT arithmetic on that same system (x_he, through
the oracle's operation-for-operation restatement of Cholesky/LDLT.h).  Here (anywhere, no GPU): (H + lambda I) x = b solved with
50 significant digits (mpmath LU), and the distance of both double-precision solutions from it.

If the two distances are of one size, the difference between the HIP path and the reference is two double-precision
solvers disagreeing inside the rounding ball of an ill-scaled system — not an inaccuracy of one of them.

  python tools/diag_parity_exact.py gpurun_out/r03b/parity_split_arrays.npz   -> profiles/parity_exact.json
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mp.mp.dps = 50
z = np.load(sys.argv[1])
names = sorted({k[:-3] for k in z.files if k.endswith("_Hh")})
report = {}
for name in names:
    H, b = z[name + "_Hh"], z[name + "_bh"]
    rows = []
    lams = sorted({float(k.split("_lam")[1].split("_x_")[0]) for k in z.files if k.startswith(name + "_lam")}, reverse=True)
    for lam in lams:
        A = mp.matrix(H.tolist())
        for i in range(171):
            A[i, i] += mp.mpf(lam)
        x = mp.lu_solve(A, mp.matrix(b.tolist()))
        xe = np.array([float(v) for v in x])
        key = "%s_lam%g_x_" % (name, lam)
        x_hh, x_he = z[key + "hh"], z[key + "he"]
        # the residual b - (H + lambda I) x of each, in extended precision too: backward error relative to |H| |x| + |b|
        def backward(xv):
            r = mp.matrix(b.tolist()) - A * mp.matrix(xv.tolist())
            den = np.abs(H + lam * np.eye(171)) @ np.abs(xv) + np.abs(b)
            return float(max(abs(float(r[i])) / den[i] for i in range(171) if den[i] > 0))
        row = {"lambda": lam, "dx_inf": float(np.abs(xe).max()), "hip_minus_exact": float(np.abs(x_hh - xe).max()),
               "eigen_minus_exact": float(np.abs(x_he - xe).max()), "hip_minus_eigen": float(np.abs(x_hh - x_he).max()),
               "hip_backward_error": backward(x_hh), "eigen_backward_error": backward(x_he)}
        if key + "hp" in z.files:       # round 4: the HIP kernel in Eigen's pivot order beside the chain order that ships (x_hh)
            row["hip_pivoted_minus_exact"] = float(np.abs(z[key + "hp"] - xe).max())
            row["hip_pivoted_backward_error"] = backward(z[key + "hp"])
        rows.append(row)
        print("%-32s lambda %9.3g  hip-exact %.2e  eigen-exact %.2e  hip-eigen %.2e   backward: hip %.1e eigen %.1e" % (
            name, lam, row["hip_minus_exact"], row["eigen_minus_exact"], row["hip_minus_eigen"], row["hip_backward_error"], row["eigen_backward_error"]))
    report[name] = rows
json.dump(report, open(os.path.join(ROOT, "profiles", "parity_exact.json"), "w"), indent=1)
