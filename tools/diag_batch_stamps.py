#!/usr/bin/env python3
"""Diagnostic only (-DVIO_STAMPS build, csrc/diag/libvio_hip_stamps.so): what a k_linearize_b workgroup costs when every CU is
busy with other windows' workgroups, and how long a CU sits between two of them.
  python tools/diag_batch_stamps.py [windows] [landmarks]
Phase stamps are shader-clock ticks (s_memtime), the timeline is the 100 MHz device-wide clock (s_memrealtime)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
lib = vio.VioLib(os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc", "diag", os.environ.get("VIO_DIAG_LIB", "libvio_hip_stamps.so")), "vio_")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
policy = int(os.environ.get("VIO_ITEM_POLICY", "1"))
lead = lib.context(item_policy=policy)
members = [lead] + [lib.context(stream=lead.get_stream(), item_policy=policy) for _ in range(B - 1)]
for i, c in enumerate(members):
    c.load(vio.synth.make_window(n, seed=100 + i))
for _ in range(4):
    lib.batch_gn_iteration(members, 5e5)
lead.synchronize()
f = lib.dll.vio_debug_stamps
f.restype = C.c_int
rows = []
for wi, c in enumerate(members):
    g_ = int(os.environ.get("VIO_G_MAX", "110" if policy == 1 else "82"))
    nb = (n + g_ - 1) // g_ + 10
    buf = np.zeros((nb, 16), dtype=np.uint64)
    assert f(c.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb)) == 0
    st = buf.astype(np.int64)
    ok = (st[:, 5] > 0) & (st[:, 1] > 0) & (st[:, 5] - st[:, 0] < 10**7) & (st[:, 5] > st[:, 0])
    rows.append(st[ok])
st = np.concatenate(rows)
names = ["head (landmark update + loads)", "phase 1 (per observation)", "phase 1.5 (per landmark)", "phase 2 (matrix cores)", "combine + store"]
d = np.diff(st[:, :6], axis=1)
print("visual workgroups of the last batch iteration: %d (%d windows)" % (len(st), B))
for k, nm in enumerate(names):
    print("  %-32s mean %8.1f  p10 %8.1f  p90 %8.1f" % (nm, d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
tot = st[:, 5] - st[:, 0]
print("  total (shader ticks)             mean %8.1f  p10 %8.1f  p90 %8.1f" % (tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
rt = (st[:, 9] - st[:, 8]) * 10.0       # ns
print("  total (100 MHz clock)            mean %8.0f ns  -> shader clock %.2f GHz while the device is full" % (rt.mean(), tot.mean() / rt.mean()))
key = st[:, 10] * 65536 + ((st[:, 11] >> 8) & 0xff)
gaps, per_cu = [], []
for k in np.unique(key):
    s = st[key == k]
    s = s[np.argsort(s[:, 8])]
    per_cu.append(len(s))
    g = (s[1:, 8] - s[:-1, 9]) * 10.0
    gaps.append(g)
gaps = np.concatenate(gaps)
print("CU slots seen: %d, workgroups per slot %d..%d" % (len(per_cu), min(per_cu), max(per_cu)))
print("gap between the end of a workgroup and the start of the next on the same CU: median %.0f ns, mean %.0f, p90 %.0f (negative = co-resident)" % (
    np.median(gaps), gaps.mean(), np.percentile(gaps, 90)))
span = (st[:, 9].max() - st[:, 8].min()) * 10.0
print("span of the launch: %.1f us = %.2f us per window" % (span / 1e3, span / 1e3 / B))
del c, members, lead
