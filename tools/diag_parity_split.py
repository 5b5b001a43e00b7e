#!/usr/bin/env python3
"""Where does the HIP path's per-step difference to the reference's delta_x come from: the system (k_linearize / k_reduce /
k_assemble: another summation order than Problem::MakeHessian's) or its solution (k_pose_solve: a blocked LDL^T on the matrix
cores instead of Eigen's column-by-column one)?

For a golden window and a list of lambdas:
    x_hh   HIP system, HIP solve                       (what ships: the chain order since round 4)
    x_hp   HIP system, HIP solve in Eigen's pivot order (vio_set_solve_order(VIO_ORDER_EIGEN): round 3's kernel)
    x_he   HIP system, Eigen's LDLT arithmetic         (oracle/vio_oracle.c: vioo_ldlt_solve restates Cholesky/LDLT.h operation for operation)
    x_oe   oracle system, Eigen's LDLT arithmetic      (the oracle's own delta_x: 4e-15 from the reference's)
    |x_hh - x_he|  = the solver's share,   |x_he - x_oe| = the system's share
and, where the fixture holds them (tests/golden/ldlt.npz: the reference's own matrix through Eigen::LDLT), the reference's vectors.

  python tools/diag_parity_split.py            (GPU box) -> profiles/parity_split.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN_DIR, load_package  # noqa: E402
import vio_testutil as tu  # noqa: E402

vio = load_package()
orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")
hip = vio.load_hip()          # (VIO_HIP_LIB selects another build of the library: __init__.py)
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))      # noqa: E731


def eigen_ldlt(A, b):
    x = np.zeros(A.shape[0])
    tr = np.zeros(A.shape[0], dtype=np.int32)
    f = orc.dll.vioo_ldlt_solve
    f.restype = None
    f(C.c_int(A.shape[0]), dp(np.ascontiguousarray(A)), dp(np.ascontiguousarray(b)), dp(x), tr.ctypes.data_as(C.POINTER(C.c_int32)))
    return x


report, dump = {}, {}
zl = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt.npz")))
for name in ("window_n50_s42", "window_n300_s45_prior", "window_n300_s43", "window_n120_s44_ragged_extfree"):
    z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    w = tu.arrays_to_window(vio, z)
    kw = {"ext_fixed": int(z["cfg_ext_fixed"])} if "cfg_ext_fixed" in z else {}
    ch, co = hip.context(**kw), orc.context(**kw)
    ch.load(w)
    co.load(w)
    ch.linearize()
    co.linearize()
    Hh, bh = ch.get_schur_system()
    Ho, bo = co.get_schur_system()
    _, lam0 = co.init_lm()
    ch.init_lm()
    rows = []
    # round 4: what ships is the chain order (static structured elimination, csrc/vio_pose_solve_chain.h); the kernel in Eigen's pivot
    # order (vio_set_solve_order) is solved beside it: x_hp
    x_pivoted = {}
    ch.set_solve_order(vio.capi.ORDER_EIGEN)
    ch.linearize()
    ch.init_lm()
    for lam in (lam0, 1e3, 240.0, 15.0, 1.0):
        ch.solve_linear(lam)
        x_pivoted[lam] = ch.get_delta()[0]
    ch.set_solve_order(vio.capi.ORDER_CHAIN)
    ch.linearize()
    ch.init_lm()
    for lam in (lam0, 1e3, 240.0, 15.0, 1.0):
        ch.solve_linear(lam)
        co.solve_linear(lam)
        x_hh, x_oe = ch.get_delta()[0], co.get_delta()[0]
        dump["%s_lam%g_x_hp" % (name, lam)] = x_pivoted[lam]
        x_he = eigen_ldlt(Hh + lam * np.eye(171), bh)
        row = {"lambda": lam, "dx_inf": float(np.abs(x_oe).max()), "solver_share": float(np.abs(x_hh - x_he).max()),
               "system_share": float(np.abs(x_he - x_oe).max()), "hip_vs_oracle": float(np.abs(x_hh - x_oe).max()),
               "solver_share_pivoted_kernel": float(np.abs(x_pivoted[lam] - x_he).max())}
        if name == "window_n50_s42":
            for i in range(3):
                if abs(float(zl["lambda_%d" % i]) - lam) <= 1e-9 * lam:
                    row["hip_vs_reference_eigen"] = float(np.abs(x_hh - zl["x_%d" % i]).max())
                    row["oracle_vs_reference_eigen"] = float(np.abs(x_oe - zl["x_%d" % i]).max())
        rows.append(row)
        dump["%s_lam%g_x_hh" % (name, lam)], dump["%s_lam%g_x_he" % (name, lam)], dump["%s_lam%g_x_oe" % (name, lam)] = x_hh, x_he, x_oe
        print("%-32s lambda %9.3g  |dx| %.3g  solver %.2e  system %.2e  hip-oracle %.2e  %s" % (
            name, lam, row["dx_inf"], row["solver_share"], row["system_share"], row["hip_vs_oracle"],
            ("hip-ref %.2e oracle-ref %.2e" % (row["hip_vs_reference_eigen"], row["oracle_vs_reference_eigen"])) if "hip_vs_reference_eigen" in row else ""))
    dump[name + "_Hh"], dump[name + "_bh"], dump[name + "_Ho"], dump[name + "_bo"] = Hh, bh, Ho, bo
    report[name] = {"scaled_system_diff": tu.scaled_sym_err(Hh, Ho), "rows": rows}
json.dump(report, open(os.path.join(ROOT, "profiles", "parity_split.json"), "w"), indent=1)
if os.environ.get("VIO_PARITY_SPLIT_DUMP"):       # the systems and solutions themselves, for tools/diag_parity_exact.py (extended precision, anywhere)
    np.savez_compressed(os.environ["VIO_PARITY_SPLIT_DUMP"], **dump)
