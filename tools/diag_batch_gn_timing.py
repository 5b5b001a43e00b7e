#!/usr/bin/env python3
"""Wall-clock time of the batched GN loop (vio_batch_gn_iteration), no profiling events in the stream.
  python tools/diag_batch_gn_timing.py [windows] [landmarks] [iterations] [xyz]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
its = int(sys.argv[3]) if len(sys.argv) > 3 else 50
xyz = len(sys.argv) > 4 and sys.argv[4] == "xyz"
make = vio.synth.make_window_xyz if xyz else vio.synth.make_window
policy = int(os.environ.get("VIO_ITEM_POLICY", "1"))          # 1: VIO_ITEMS_THROUGHPUT (what a batch wants), 0: the single-window default
lead = hip.context(item_policy=policy)
members = [lead] + [hip.context(stream=lead.get_stream(), item_policy=policy) for _ in range(B - 1)]
for i, c in enumerate(members):
    c.load(make(n, seed=100 + i))
lam = 5e5
for _ in range(5):
    hip.batch_gn_iteration(members, lam)
lead.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(its):
        hip.batch_gn_iteration(members, lam)
    lead.synchronize()
    dt = time.perf_counter() - t0
    print("B=%d n=%d%s: %.3f us per window-iteration (%.3f ms per batch iteration)" % (B, n, " xyz" if xyz else "", dt * 1e6 / (its * B), dt * 1e3 / its))
print("chi2 of window 0", lead.chi2())
del c, members, lead
