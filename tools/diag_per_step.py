#!/usr/bin/env python3
"""The per-step parity figure of SURVEY.md section 7 (one trial from the reference's own state k at the reference's lambda against the
reference's state k + 1; tests/test_gpu_parity.py::per_step_differences) for both elimination orders: worst difference per window.
  python tools/diag_per_step.py      (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

vio = load_package()
hip = vio.load_hip()
for name, order in (("chain", vio.capi.ORDER_CHAIN), ("pivoted (Eigen's order)", vio.capi.ORDER_EIGEN)):
    d = tp.per_step_differences(vio, hip, order)
    print(name, "worst over all windows and iterations: %.2e" % max(max(v) for v in d.values()))
    for k, v in d.items():
        print("   %-40s first %.1e  worst %.1e  (%d steps)" % (k, v[0], max(v), len(v)))
