#!/usr/bin/env python3
"""Diagnostic: 20 x Problem::Solve(10) on the bench window and nothing else, for `rocprofv3 --kernel-trace` (the kernels of vio_solve's loop alone:
round 4: k_pose_solve_c 32.1 us where it solves (29.8 with the loop's last, empty, launch), k_linearize 14.0, k_reduce_c 6.1 = 54 us per LM iteration against
49.5 in the fixed-lambda GN loop)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package
vio = load_package()
hip = vio.load_hip()
w0 = vio.synth.make_window(300, seed=41, t0=0.9)
c0 = hip.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
w = vio.synth.make_window(20000, seed=42); w.prior = prior
ctx = hip.context()
for r in range(20):
    ctx.load(w); ctx.linearize(); rep = ctx.solve(10)
print(rep.iterations, rep.trials)
