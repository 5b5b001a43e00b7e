#!/usr/bin/env python3
"""Problem::Solve(10) wall time in the reference's real regime (N = 300 landmarks): the reference's own problem.cc on this
box's host, and the drop-in problem_hip.cc over libvio_hip.so — both behind the same harness (oracle/ref_harness.cpp:
graph built from the reference's Vertex / Edge objects, then Problem::Solve).  Needs oracle/_ref (built where
/root/reference exists; it travels to the GPU box as built libraries).   python tools/bench_shim.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = vio.synth.make_window(n, seed=43)
ref = vio.VioLib(os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so"), "vior_")
shim = vio.VioLib(os.path.join(ROOT, "oracle", "_ref", "libvio_refshim_hip.so"), "vior_")
hip = vio.load_hip()


def timed(lib, reps):
    c = lib.context()
    out = []
    for r in range(reps + 1):
        c.load(w)
        t = time.perf_counter()
        rep = c.solve(10)
        out.append((time.perf_counter() - t) * 1e3)
    return min(out[1:]), sum(out[1:]) / reps, rep.iterations


print("N = %d landmarks, %d observations; Solve(10), ms (best / mean of the repetitions after the first)" % (w.n_landmarks, w.n_observations))
for name, lib, reps in (("reference problem.cc, 1 host thread", ref, 3), ("problem_hip.cc over libvio_hip.so (graph build + flatten + upload + solve + write-back)", shim, 20),
                        ("C ABI directly (vio_set_* + vio_solve)", hip, 20)):
    if lib is hip:
        c = lib.context()
        ts = []
        for r in range(21):
            t = time.perf_counter()
            c.load(w)
            rep = c.solve(10)
            ts.append((time.perf_counter() - t) * 1e3)
        best, mean, it = min(ts[1:]), sum(ts[1:]) / 20, rep.iterations
    else:
        best, mean, it = timed(lib, reps)
    print("  %-90s %9.3f %9.3f   (%d iterations)" % (name, best, mean, it))
