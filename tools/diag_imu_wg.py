#!/usr/bin/env python3
"""Diagnostic (stamps build): how long the ten IMU workgroups of one k_linearize launch run beside the visual ones (round 4: 8.1-8.6 us against 10.2)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
vio = load_package()
lib = vio.VioLib(os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc", "diag", "libvio_hip_stamps.so"), "vio_")
w = vio.synth.make_window(20000, seed=42)
ctx = lib.context(); ctx.load(w)
for _ in range(3): ctx.linearize()
_, lam = ctx.init_lm()
for _ in range(4): ctx.gn_iteration(lam)
ctx.synchronize()
nb = 270
buf = np.zeros((nb, 16), dtype=np.uint64)
f = lib.dll.vio_debug_stamps; f.restype = C.c_int
assert f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb)) == 0
st = buf.astype(np.int64)
for b in range(240, 256):
    print(b, "cycles first->last stamp", st[b,5]-st[b,0], "realtime us", (st[b,9]-st[b,8])/100.0, "start rel", (st[b,8]-st[240:255,8].min())/100.0)
