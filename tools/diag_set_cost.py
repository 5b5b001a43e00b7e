#!/usr/bin/env python3
"""Host cost of each vio_set_* call of a frame through the ctypes binding (fresh window every frame), microseconds, medians.
  python tools/diag_set_cost.py [landmarks] [frames]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ws = [vio.synth.make_window(n, seed=42 + r, t0=1.0 + 0.1 * r) for r in range(frames)]
wp = vio.synth.make_window(300, seed=41, t0=0.9)
cp = hip.context(); cp.load(wp); cp.solve(10); prior = cp.marginalize(vio.MARG_OLD); del cp
c = hip.context()
acc = {k: [] for k in ("window", "landmarks", "observations", "imu x10", "prior", "map+fill+commit")}
for r, w in enumerate(ws):
    pr = {k: (v + 1e-9 * r) for k, v in prior.items()}       # a new prior every frame
    t = [time.perf_counter()]
    c.set_window(w.poses, w.speed_bias, w.ext); t.append(time.perf_counter())
    c.set_landmarks(w.inv_depth); t.append(time.perf_counter())
    c.set_observations(w.lm, w.host, w.target, w.pts_i, w.pts_j); t.append(time.perf_counter())
    for k, p in enumerate(w.preint):
        c.set_imu(k, p)
    t.append(time.perf_counter())
    c.set_prior(pr); t.append(time.perf_counter())
    lm, host, target, pi, pj = c.map_observations(w.n_observations)
    lm[:], host[:], target[:], pi[:], pj[:] = w.lm, w.host, w.target, w.pts_i, w.pts_j
    c.commit_observations(); t.append(time.perf_counter())
    c.linearize(); c.synchronize()
    if r:
        for k, a, b in zip(acc, t[:-1], t[1:]):
            acc[k].append((b - a) * 1e6)
print("  ".join("%s %.0f us" % (k, statistics.median(v)) for k, v in acc.items()))
