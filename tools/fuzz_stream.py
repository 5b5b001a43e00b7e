#!/usr/bin/env python3
"""Randomised stream sweep (diagnostic): the stream driver on the HIP backend against the oracle over random stream
shapes (length, landmarks per frame, track length, triangulated depths, non-keyframes).  python tools/fuzz_stream.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for case in range(cases):
    nf = int(rng.randint(13, 30)); per = int(rng.choice([4, 8, 15, 30, 60])); tl = int(rng.randint(2, 9))
    tri = bool(rng.randint(2)); nk = int(rng.choice([0, 0, 3, 4, 5]))
    st = vio.stream.SyntheticStream(n_frames=nf, landmarks_per_frame=per, track_len=tl, seed=100 + case)
    try:
        dh = vio.stream.StreamDriver(hip, st, triangulate=tri, nonkey_every=nk)
        do = vio.stream.StreamDriver(orc, st, triangulate=tri, nonkey_every=nk)
        th, to = dh.run(), do.run()
        d = np.abs(th[:, 1:4] - to[:, 1:4]).max() if len(th) else 0.0
        ok = len(th) == len(to) and dh.flags == do.flags and d < 2e-3
        worst = max(worst, d)
        print("%s case %2d: frames %2d per-frame %2d track %d triangulate %d nonkey %d | windows %d max |dp| %.1e" % ("ok  " if ok else "FAIL", case, nf, per, tl, tri, nk, len(th), d), flush=True)
    except Exception as exc:  # both backends must at least fail alike
        print("EXC  case %2d: frames %2d per-frame %2d track %d triangulate %d nonkey %d | %r" % (case, nf, per, tl, tri, nk, exc), flush=True)
print("worst", worst)
