#!/usr/bin/env python3
"""Measurement of the 8f-2 row: FeatureManager::triangulate for N tracks, HIP (vio_triangulate: upload + kernel +
download, host buffers in and out as the boundary hands them over) against the oracle's C restatement on one core."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
from test_triangulate import make_tracks  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sf, off, pts, poses, ext, d0, _ = make_tracks(vio, n, seed=1, noise=1.0 / 460.0, have_depth_frac=0.0)
ch, co = hip.context(), orc.context()
ch.triangulate(sf, off, pts, poses, ext, d0)
t = time.perf_counter()
for _ in range(20):
    got = ch.triangulate(sf, off, pts, poses, ext, d0)
th = (time.perf_counter() - t) / 20
t = time.perf_counter()
for _ in range(5):
    want = co.triangulate(sf, off, pts, poses, ext, d0)
to = (time.perf_counter() - t) / 5
print("tracks %d, observations %d: HIP %.3f ms per call (PCIe in/out included), oracle 1 core %.3f ms, max rel diff %.2e"
      % (n, off[-1], th * 1e3, to * 1e3, np.max(np.abs(got - want) / np.abs(want))))
