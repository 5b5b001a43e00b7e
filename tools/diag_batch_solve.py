#!/usr/bin/env python3
"""Diagnostic: vio_batch_solve of B 20k-landmark windows, for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/diag_batch_solve.py).
  python tools/diag_batch_solve.py [B] [landmarks] [prior 0/1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
with_prior = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prior = None
if with_prior:
    orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
    w = vio.synth.make_window(200, seed=71, t0=0.9)
    c = orc.context()
    c.load(w)
    c.solve(10)
    prior = c.marginalize(vio.MARG_OLD)
lead = hip.context()
members = [lead] + [hip.context(stream=lead.get_stream()) for _ in range(B - 1)]
wbs = []
for i, cb in enumerate(members):
    wb = vio.synth.make_window(n, seed=100 + i)
    wb.prior = prior
    wbs.append(wb)
for rep in range(2):
    for cb, wb in zip(members, wbs):
        cb.load(wb)
        cb.linearize()
    lead.synchronize()
    t0 = time.perf_counter()
    reps = hip.batch_solve(members, 10)
    dt = time.perf_counter() - t0
    print("batch solve %.3f ms; iterations %s trials %s accepted %s" % (dt * 1e3, [r.iterations for r in reps][:16], [r.trials for r in reps][:16], [r.accepted for r in reps][:16]))
    print("  max trials %d, sum trials %d, stop reasons %s" % (max(r.trials for r in reps), sum(r.trials for r in reps), sorted(set(r.stop_reason for r in reps))))
del cb, members, lead
