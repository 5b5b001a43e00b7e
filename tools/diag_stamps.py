#!/usr/bin/env python3
"""Diagnostic only: phase shares of k_linearize from the -DVIO_STAMPS build (csrc/diag/libvio_hip_stamps.so).
Never used for timing claims: stamps change the schedule; read the SHARES."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
lib = vio.VioLib(os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc", "diag", os.environ.get("VIO_DIAG_LIB", "libvio_hip_stamps.so")), "vio_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
xyz = len(sys.argv) > 2 and sys.argv[2] == "xyz"          # k_linearize_xyz's phases (the finer slots below are the inverse-depth kernel's)
w = (vio.synth.make_window_xyz if xyz else vio.synth.make_window)(n, seed=42)
ctx = lib.context()
ctx.load(w)
for _ in range(3):
    ctx.linearize()
if os.environ.get("VIO_DIAG_GN") == "1":        # the GN loop's k_linearize (carries the previous step's landmark update)
    _, lam_gn = ctx.init_lm()
    for _ in range(4):
        ctx.gn_iteration(lam_gn)
ctx.synchronize()
gmax = int(os.environ.get("VIO_G_MAX", "82"))
nb = (n + gmax - 1) // gmax + 10
buf = np.zeros((nb, 16), dtype=np.uint64)
f = lib.dll.vio_debug_stamps
f.restype = C.c_int
assert f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb)) == 0
st = buf.astype(np.int64)
valid = st[:, 5] > 0
vis = valid & (st[:, 1] > 0) & (np.arange(nb) < (n + gmax - 1) // gmax) & (st[:, 5] - st[:, 0] < 10**7)
imu = valid & (st[:, 1] == 0)
names = ["load pair table", "phase 1 (per observation)", "phase 1.5 (per landmark)", "phase 2 (strips)", "combine + store"]
d = np.diff(st[vis][:, :6], axis=1)
print("visual workgroups: %d   (s_memtime ticks = 100 MHz?  constant clock; shares are what matters)" % vis.sum())
for k, nm in enumerate(names):
    print("  %-28s mean %8.1f  max %8.1f" % (nm, d[:, k].mean(), d[:, k].max()))
if xyz:
    print("  total                        mean %8.1f  max %8.1f" % ((st[vis][:, 5] - st[vis][:, 0]).mean(), (st[vis][:, 5] - st[vis][:, 0]).max()))
    sys.exit(0)
print("  inside phase 2, wave 0: product done after %.0f, vector sums done after %.0f (of the phase)" % ((st[vis][:, 6] - st[vis][:, 3]).mean(), (st[vis][:, 7] - st[vis][:, 3]).mean()))
print("  wave 0 (direct product): chunk loop done after %.0f, remainder %.0f, stored %.0f | wave 4 (Schur tile): starts %.0f, chunk loop done %.0f (of the phase)" % tuple(
    (st[vis][:, b_] - st[vis][:, 3]).mean() for b_ in (14, 15, 6, 12, 13)))
print("  combine: block sums done after %.0f, slab elements after %.0f, rows stored after %.0f (of the phase)" % tuple((st[vis][:, b_] - st[vis][:, 4]).mean() for b_ in (12, 13, 5)))
print("  total                        mean %8.1f  max %8.1f" % ((st[vis][:, 5] - st[vis][:, 0]).mean(), (st[vis][:, 5] - st[vis][:, 0]).max()))
if imu.any():
    t = st[imu][:, 5] - st[imu][:, 0]
    print("IMU workgroups: %d  total mean %.1f max %.1f" % (imu.sum(), t.mean(), t.max()))
t0, t1 = st[valid][:, 0].min(), st[valid][:, 5].max()
print("kernel span (first start -> last end): %d ticks; start spread %d" % (t1 - t0, st[valid][:, 0].max() - t0))

# ---- k_pose_solve (stamps are cumulative ticks since kernel start; slots 8/9 are sums over the 22 panels)
ctx.linearize()
_, lam = ctx.init_lm()
for _ in range(3):
    ctx.solve_linear(lam)
ctx.synchronize()
assert f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(4)) == 0
st2 = buf.astype(np.int64)[1]
st3 = buf.astype(np.int64)[2]; st4 = buf.astype(np.int64)[3]
print("per step K: wave-1 update loop      ", list(st2[:10]))
print("per step K: wave-0 update tile (K+1) ", list(st3[:10]))
print("per step K: wave-0 F(K+1)            ", list(st4[:10]))
st = buf.astype(np.int64)[0]
print("k_pose_solve prologue: diag+rhs loaded %d | pivot order known %d" % (st[4], st[5]))
print("k_pose_solve: load+permute %d | factorisation done %d (F+U phases %d, S phases %d) | back-substitution done %d | end %d"
      % (st[0], st[1], st[8], st[9], st[2], st[3]))
