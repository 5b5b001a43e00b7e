#!/usr/bin/env python3
"""Host cost of a frame's set + plan build, phase by phase (VIO_HOST_TIMING=1 prints from the library), on fresh windows of a stream.
  VIO_HOST_TIMING=1 python tools/diag_host_plan.py [landmarks] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ws = [vio.synth.make_window(n, seed=42 + r, t0=1.0 + 0.1 * r) for r in range(frames)]
c = hip.context()
acc = []
for w in ws:
    t0 = time.perf_counter()
    c.load(w)
    t1 = time.perf_counter()
    c.linearize()
    c.synchronize()
    t2 = time.perf_counter()
    acc.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, c.host_timing()["activate_plan_us"]))
    print("set %.0f us, plan + upload + linearize %.0f us" % acc[-1][:2], flush=True)
import statistics
print("medians over %d frames: set %.0f us, plan + upload + linearize %.0f us, of which build_plan + upload %.0f us" % (
    len(acc) - 1, *(statistics.median(v[i] for v in acc[1:]) for i in range(3))))
