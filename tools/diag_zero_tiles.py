#!/usr/bin/env python3
"""Could k_pose_solve skip tiles that are EXACTLY zero?  The reduced pose system of a window in Eigen's
pivot order (rank of |diag(H) + lambda|, what k_assemble writes), factored right-looking in 16x16 tiles as k_pose_solve does:
how many S tiles (A_IK M_K) and update tiles (A_IJ -= U_IK D^-1 U_JK^T) are identically zero, fill-in included?
The system comes from the CPU checker (structure only: which entries are exactly zero does not depend on who summed them).
  python tools/diag_zero_tiles.py  > profiles/r03j_pose_solve_zero_tiles.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

vio = load_package()
orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")


def name(i):
    if i < 6:
        return "ext"
    f, r = divmod(i - 6, 15)
    return ("p%d" if r < 6 else "v%d" if r < 9 else "ba%d" if r < 12 else "bg%d") % f


for n, seed in ((2000, 3), (20000, 42)):
    w = vio.synth.make_window(n, seed=seed)
    c = orc.context()
    c.load(w)
    c.linearize()
    H, _ = c.get_schur_system()
    for lam in (1e5, 1e2):
        A = H + lam * np.eye(171)
        p = np.argsort(-np.abs(np.diag(A)), kind="stable")
        A = np.pad(A[np.ix_(p, p)], ((0, 5), (0, 5)))
        A[np.arange(171, 176), np.arange(171, 176)] = 1.0
        nz_in = int((np.abs(A[:171, :171]) > 0).sum())
        s_nz = s_tot = u_nz = u_tot = 0
        for K in range(11):
            k0 = 16 * K
            for j in range(k0, k0 + 16):           # the tile's 16 pivots, trailing matrix updated entry by entry
                d = A[j, j]
                if d != 0:
                    l = A[j + 1:, j] / d
                    A[j + 1:, j + 1:] -= np.outer(l, A[j + 1:, j])
                    A[j + 1:, j] = l
            live = [I for I in range(K + 1, 11) if np.any(A[16 * I:16 * I + 16, k0:k0 + 16] != 0)]
            s_tot += 10 - K
            s_nz += len(live)
            for I in range(K + 1, 11):
                for J in range(K + 1, I + 1):
                    u_tot += 1
                    u_nz += (I in live) and (J in live)
        print("window of %d landmarks, lambda %g: %d of 171^2 entries non-zero before the factorisation; S tiles non-zero %d of %d, update tiles non-zero %d of %d"
              % (n, lam, nz_in, s_nz, s_tot, u_nz, u_tot))
    print("  pivot order (first 48): " + " ".join(name(i) for i in p[:48]))
print("""
=> no tile is ever exactly zero.  The order is a sort of the diagonal: the 33 gyro-bias entries of ALL frames come first, then the 33
accelerometer-bias entries, then poses and velocities interleaved over the frames; every 16-pivot tile therefore holds variables of
at least five frames, each coupled (IMU factor) to both neighbours' poses, velocities and biases, and the landmark Schur complement
couples every pose with every other.  Exact-zero skipping would need an order that keeps a frame's variables together — Eigen's does not,
and its order is what the parity tests pin (ldlt.npz transpositions).""")
