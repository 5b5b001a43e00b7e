#!/usr/bin/env python3
"""tests/golden/window_n200_s46_huber_d50.npz from the COMPILED REFERENCE: a Huber window whose step is comparable.

For an outlier edge (e2 > delta^2) rho' + 2 rho'' e2 = delta/sqrt(e2) - delta/sqrt(e2) is zero in exact arithmetic for EVERY
outlier, however far from the inlier boundary (loss_function.cc:16-20), so the `> 0` test of Edge::RobustInfo (edge.cc:62) is
decided by the last bit of e2 in the reference itself and the edge's weight along its residual is rho' or 0 at random: no two
evaluation orders agree, and window_n200_s46_huber.npz (delta = 1) is compared on chi2 and lambda_0 only.  With delta = 50
every edge of the same window is an inlier (largest whitened residual: see the print below), the loss object is on the path
of every edge (Compute, RobustInfo) and everything downstream is well defined.
Run where /root/reference exists:   python tests/golden/make_golden_huber.py"""
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
z = dict(np.load(os.path.join(HERE, "window_n200_s46_huber.npz")))
w = tu.arrays_to_window(vio, z)
d = tu.window_to_arrays(w)
kw = dict(loss_type=vio.LOSS_HUBER, loss_delta=50.0)
ctx = ref.context(**kw)
ctx.load(w)
step = tu.run_stepwise(ctx)
step.pop("Hs")
d.update({"step_" + k: v for k, v in step.items()})
ctx2 = ref.context(**kw)
ctx2.load(w)
sol, _ = tu.run_solve(ctx2, 10)
d.update({"solve_" + k: v for k, v in sol.items()})
d["cfg_loss_type"], d["cfg_loss_delta"] = np.int32(vio.LOSS_HUBER), np.float64(50.0)
# largest whitened residual of the window at the linearisation point, from chi2 of the trivial loss: not needed; report e2 max
orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")
import ctypes as C
f = orc.dll.vioo_reproj_edge
f.restype = None
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
e2max = 0.0
for e in range(w.lm.size):
    r = np.zeros(2)
    f(dp(np.ascontiguousarray(w.poses[w.host[e]])), dp(np.ascontiguousarray(w.poses[w.target[e]])), dp(np.ascontiguousarray(w.ext)),
      C.c_double(w.inv_depth[w.lm[e]]), dp(np.ascontiguousarray(w.pts_i[e])), dp(np.ascontiguousarray(w.pts_j[e])), dp(r), None, None, None, None)
    e2max = max(e2max, float(r @ r) * (460 / 1.5) ** 2)
print("largest e2 at the linearisation point: %.1f (delta^2 = 2500)" % e2max)
assert e2max < 2500
np.savez_compressed(os.path.join(HERE, "window_n200_s46_huber_d50.npz"), **d)
print("window_n200_s46_huber_d50.npz written,", int(sol["iterations"]), "iterations")
