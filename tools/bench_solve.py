#!/usr/bin/env python3
"""Wall time of the LM solve the Estimator actually calls, Problem::Solve(10) == vio_solve, split into the part that
depends on the graph topology (vio_set_* + plan build + upload: once per frame) and the solve itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
vio = load_package()
hip = vio.load_hip()
w0 = vio.synth.make_window(300, seed=41, t0=0.9)
c0 = hip.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
for n, with_prior in ((300, False), (300, True), (2000, True), (20000, False), (20000, True)):
    w = vio.synth.make_window(n, seed=42)
    if with_prior:
        w.prior = prior
    ctx = hip.context()
    tl, tp, ts = [], [], []
    for r in range(5):
        t = time.perf_counter(); ctx.load(w); tl.append(time.perf_counter() - t)
        t = time.perf_counter(); ctx.linearize(); ctx.synchronize(); tp.append(time.perf_counter() - t)   # plan build + upload + 1 linearisation
        t = time.perf_counter(); rep = ctx.solve(10); ts.append(time.perf_counter() - t)
    print("N=%5d%s: set_* %.3f ms | plan + upload + first linearisation %.3f ms | Solve(10) %.3f ms (%d iterations, %d trials, chi2 %.4g -> %.4g)"
          % (n, " + prior" if with_prior else "        ", min(tl) * 1e3, min(tp) * 1e3, min(ts) * 1e3, rep.iterations, rep.trials, rep.initial_chi2, rep.final_chi2))
