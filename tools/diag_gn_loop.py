"""A short GN loop on the headline window for counter passes (rocprofv3 --pmc ... -- python3 tools/diag_gn_loop.py [xyz])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
xyz = len(sys.argv) > 1 and sys.argv[1] == "xyz"
w = (vio.synth.make_window_xyz if xyz else vio.synth.make_window)(20000, seed=42)
ctx = hip.context()
ctx.load(w)
ctx.linearize()
_, lam = ctx.init_lm()
for _ in range(30):
    ctx.gn_iteration(lam)
ctx.synchronize()
print("chi2", ctx.chi2())
