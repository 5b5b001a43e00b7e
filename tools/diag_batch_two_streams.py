#!/usr/bin/env python3
"""Batched GN loop as ONE batch of B windows on one stream against TWO batches of B/2 on two streams, enqueued alternately from
one thread: the small kernels of one half (k_reduce_b, k_rank_b, k_assemble_b, k_pose_solve_b: 11 % of a batch iteration, most of
the chip idle) can then run beside the other half's k_linearize_b.
  python tools/diag_batch_two_streams.py [windows] [landmarks] [iterations]"""
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
its = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lam = 5e5
POL = int(os.environ.get("VIO_ITEM_POLICY", "1"))          # 1: VIO_ITEMS_THROUGHPUT (what a batch wants)
ws = [vio.synth.make_window(n, seed=100 + i) for i in range(B)]


def group(windows):
    lead = hip.context(item_policy=POL, )
    members = [lead] + [hip.context(item_policy=POL, stream=lead.get_stream()) for _ in windows[1:]]
    for c, w in zip(members, windows):
        c.load(w)
    return members


for parts in (1, 2, 4):
    groups = [group(ws[i::parts]) for i in range(parts)]
    for _ in range(5):
        for g in groups:
            hip.batch_gn_iteration(g, lam)
    for g in groups:
        g[0].synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(its):
            for g in groups:
                hip.batch_gn_iteration(g, lam)
        for g in groups:
            g[0].synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%d stream(s) x %d windows: %.3f us per window-iteration (chi2 of window 0: %.6f)" % (parts, B // parts, best * 1e6 / (its * B), groups[0][0].chi2()))
    for g in groups:
        while len(g) > 1:
            g.pop()
        g.pop()
