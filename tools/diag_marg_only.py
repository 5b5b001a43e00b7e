#!/usr/bin/env python3
"""Diagnostic: the frame's tail — Solve(10) then MargOldFrame on the bench window, 20 times — for `rocprofv3 --kernel-trace`: which kernels the
marginalisation's device part (0.09 ms of the frame) is made of."""
import os, sys
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
    ctx.load(w); ctx.solve(2)
    ctx.marginalize(vio.MARG_OLD)
print("done")
