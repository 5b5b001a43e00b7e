"""Wall-clock time of the GN loop without any profiling events in the stream (python tools/diag_gn_timing.py [landmarks] [iterations] [xyz]);
VIO_GN_GRAPH=1 runs the steady-state iteration as an instantiated hipGraph (experiment)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
its = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
xyz = len(sys.argv) > 3 and sys.argv[3] == "xyz"
w = (vio.synth.make_window_xyz if xyz else vio.synth.make_window)(n, seed=42)
ctx = hip.context()
ctx.load(w)
ctx.linearize()
_, lam = ctx.init_lm()
for _ in range(50):
    ctx.gn_iteration(lam)
ctx.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(its):
        ctx.gn_iteration(lam)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    best = min(best, (t2 - t0) / its)
    print("n=%d: %.3f us per iteration (host enqueue %.3f us per iteration)" % (n, (t2 - t0) / its * 1e6, (t1 - t0) / its * 1e6))
print("chi2", ctx.chi2())
