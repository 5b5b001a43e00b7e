#!/usr/bin/env python3
"""Diagnostic: one frame of the reference's real regime through the HIP library (ctypes), phase by phase with the library's own host split —
bench.py's per_frame_small protocol for one backend.  Usage: python tools/diag_small_frame.py [n=150] [frames=20] [pipelined=0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
vio = load_package()
hip = vio.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pipelined = len(sys.argv) > 3 and sys.argv[3] == "1"
wins = [vio.synth.make_window(n, seed=300 + r, t0=1.0 + 0.1 * r, ragged=True) for r in range(22)]
for w in wins:
    w.preint = [p if (p is None or isinstance(p, vio.VioPreint)) else vio.VioPreint.from_dict(p) for p in w.preint]
c = hip.context()
acc = {k: [] for k in ("set", "plan_upload_linearize", "solve10", "marginalize", "frame")}
split = {k: [] for k in ("marg_device_us", "marg_tail_us", "marg_prepare_us", "activate_plan_us", "activate_push_us")}
prior = None
for r in range(reps + 3):
    w = wins[r % len(wins)]
    t0 = time.perf_counter()
    if pipelined:
        c.set_window(w.poses, w.speed_bias, w.ext); c.set_landmarks(w.inv_depth)
        c.set_observations(w.lm, w.host, w.target, w.pts_i, w.pts_j); c.set_imu_all(w.preint)
        if os.environ.get("VIO_DIAG_NO_PREPARE") is None:
            c.prepare()
        if r > 0:
            prior = c.marginalize_end()
        c.set_prior(prior)
    else:
        w.prior = prior
        c.load(w)
    t1 = time.perf_counter()
    c.linearize(); c.synchronize()
    hta = c.host_timing()
    t2 = time.perf_counter()
    rep = c.solve(10)
    t3 = time.perf_counter()
    if pipelined:
        c.marginalize_begin(vio.MARG_OLD)
    else:
        prior = c.marginalize(vio.MARG_OLD)
    t4 = time.perf_counter()
    if r < 3:
        continue
    for k, v in zip(("set", "plan_upload_linearize", "solve10", "marginalize", "frame"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
        acc[k].append(v * 1e3)
    ht = c.host_timing()
    for k in ("marg_device_us", "marg_tail_us", "marg_prepare_us"):
        split[k].append(ht[k])
    for k in ("activate_plan_us", "activate_push_us"):
        split[k].append(hta[k])
    live = int(ht["marg_live_rows"])
if pipelined:
    c.marginalize_end()
med = lambda v: sorted(v)[len(v) // 2]
print("small_frame n=%d pipelined=%d: " % (n, pipelined) + "  ".join("%s %.4f" % (k, med(v)) for k, v in acc.items()) + " ms | " +
      "  ".join("%s %.1f" % (k, med(v)) for k, v in split.items()) + " | live rows %d, iterations %d" % (live, rep.iterations))
