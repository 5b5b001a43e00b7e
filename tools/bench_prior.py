#!/usr/bin/env python3
"""GN iteration time with and without a marginalisation prior (the steady-state window carries one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
vio = load_package()
hip = vio.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w0 = vio.synth.make_window(300, seed=41, t0=0.9)
c0 = hip.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
for with_prior in (False, True):
    w = vio.synth.make_window(n, seed=42)
    if with_prior:
        w.prior = prior
    ctx = hip.context(); ctx.load(w)
    ctx.linearize(); _, lam = ctx.init_lm()
    for _ in range(30): ctx.gn_iteration(lam)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(300): ctx.gn_iteration(lam)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 300
    per = {}
    for kid, name in enumerate(hip.KERNELS[:4]):
        ctx.profile_begin(kid)
        for _ in range(20): ctx.gn_iteration(lam)
        ms, cnt = ctx.profile_end()
        per[name] = round(ms / max(cnt, 1) * 1e3, 2)
    print("N=%d prior=%s: %.2f us per GN iteration, chi2 %.6g  events: %s" % (n, with_prior, dt * 1e6, ctx.chi2(), per))
