#!/usr/bin/env python3
"""Diagnostic (stamps build, tools/build_diag.sh stamps): the workgroups of the LAST k_linearize launch of a Problem::Solve(10) on a window of the
reference's real regime (N = 150 / 300, ragged tracks, chained prior): per workgroup the cycles between its first and last stamp, the phases of the
item workgroups, and when it started / ended on the device-wide clock.  Usage: python tools/diag_small_wg.py [n] [ragged=1]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
vio = load_package()
lib = vio.VioLib(os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc", "diag", "libvio_hip_stamps.so"), "vio_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ragged = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
ctx = lib.context()
prior = None
for r in range(3):
    w = vio.synth.make_window(n, seed=300 + r, t0=1.0 + 0.1 * r, ragged=ragged)
    w.prior = prior
    ctx.load(w)
    rep = ctx.solve(10)
    if r < 2:
        prior = ctx.marginalize(vio.MARG_OLD)
ctx.synchronize()
f = lib.dll.vio_debug_stamps; f.restype = C.c_int
for nb in range(400, 10, -2):          # (the buffer holds items + IMU workgroups + 48 blocks: the largest read that fits)
    buf = np.zeros((nb, 16), dtype=np.uint64)
    if f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb)) == 0:
        break
nb -= 46
buf = buf[:nb]
st = buf.astype(np.int64)
v = (st[:, 9] > st[:, 8]) & (st[:, 8] > 0) & (st[:, 9] - st[:, 8] < 10_000_000)
idx = np.nonzero(v)[0]
base = st[v][:, 8].min()
print("n=%d ragged=%s: %d workgroups stamped (blocks %d..%d), solve: %d iterations %d trials" % (n, ragged, v.sum(), idx.min(), idx.max(), rep.iterations, rep.trials))
last = idx.max()
n_items = last + 1 - 10
cyc = st[:, 5] - st[:, 0]
it = idx[idx < n_items]
im = idx[idx >= n_items]
print("item workgroups (%d): cycles first->last stamp min %d median %d max %d; realtime us median %.2f max %.2f; start offset max %.2f us, end offset max %.2f us"
      % (len(it), cyc[it].min(), np.median(cyc[it]), cyc[it].max(), np.median(st[it, 9] - st[it, 8]) / 100.0, (st[it, 9] - st[it, 8]).max() / 100.0,
         (st[it, 8] - base).max() / 100.0, (st[it, 9] - base).max() / 100.0))
ph = np.diff(st[it][:, 0:6], axis=1)
print("  phases (median cycles): head %d | phase 1 %d | phase 1.5 %d | phase 2 %d | combine %d" % tuple(np.median(ph, axis=0)))
w_ = it[np.argmax(cyc[it])]
print("  slowest item workgroup b%d: phases %s" % (w_, np.diff(st[w_, 0:6]).tolist()))
print("IMU workgroups (%d): cycles min %d median %d max %d; realtime us median %.2f max %.2f; start offset max %.2f us, end offset max %.2f us"
      % (len(im), cyc[im].min(), np.median(cyc[im]), cyc[im].max(), np.median(st[im, 9] - st[im, 8]) / 100.0, (st[im, 9] - st[im, 8]).max() / 100.0,
         (st[im, 8] - base).max() / 100.0, (st[im, 9] - base).max() / 100.0))
print("kernel-internal span (first start -> last end): %.2f us" % ((st[v][:, 9].max() - base) / 100.0))
