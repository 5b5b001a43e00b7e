#!/usr/bin/env python3
"""vio_batch_solve / vio_batch_gn_iteration against the single-window calls on random batches (diagnostic): windows of different
sizes and shapes, contexts with different losses / extrinsic flags, priors on some, IMU factors missing on some, batches of 1..9;
results must be bit-identical.      python tools/fuzz_batch_solve.py [batches] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
w0 = vio.synth.make_window(200, seed=41, t0=0.9)
c0 = orc.context(); c0.load(w0); c0.solve(10)
prior = c0.marginalize(vio.MARG_OLD)
bad = 0
only = int(os.environ.get("VIO_FUZZ_ONLY", "-1"))          # that batch alone (the sequence's random draws are kept), with the reason
for bi in range(n_batches):
    B = int(rng.randint(1, 10))
    xyz = rng.rand() < 0.3
    ws, kws = [], []
    for q in range(B):
        n = int(rng.choice([1, 9, 70, 300, 1100, 4000]))
        mk = vio.synth.make_window_xyz if xyz else vio.synth.make_window
        kw = dict(ragged=bool(rng.randint(2)))
        if xyz:
            kw["obs_per_landmark"] = int(rng.randint(3, 8))
        w = mk(n, seed=1000 * bi + q, **kw)
        if rng.rand() < 0.4:
            w.prior = prior
        if rng.rand() < 0.3:
            w.preint = list(w.preint)
            w.preint[int(rng.randint(10))] = None
        ws.append(w)
        kws.append(dict(ext_fixed=int(rng.randint(2)), loss_type=int(rng.choice([0, 1, 2]))))
    if only >= 0 and bi != only:
        rng.randint(1, 14)
        continue
    lead = hip.context(**kws[0])
    batch = [lead] + [hip.context(stream=lead.get_stream(), **kw) for kw in kws[1:]]
    solo = [hip.context(**kw) for kw in kws]
    for c, r, w in zip(batch, solo, ws):
        c.load(w); r.load(w)
    its = int(rng.randint(1, 14))
    ok = True
    try:
        reps = hip.batch_solve(batch, its)
        for qi, (c, r, rb) in enumerate(zip(batch, solo, reps)):
            rs = r.solve(its)
            if (rb.iterations, rb.trials, rb.final_chi2) != (rs.iterations, rs.trials, rs.final_chi2):
                print("  window %d (%d landmarks, %s): batch %d its / %d trials chi2 %.15e | solo %d / %d chi2 %.15e | initial %.15e %.15e"
                      % (qi, ws[qi].n_landmarks, kws[qi], rb.iterations, rb.trials, rb.final_chi2, rs.iterations, rs.trials, rs.final_chi2, rb.initial_chi2, rs.initial_chi2))
            o1 = (rb.iterations, rb.trials, rb.accepted, rb.stop_reason, rb.final_chi2, rb.final_lambda) == (rs.iterations, rs.trials, rs.accepted, rs.stop_reason, rs.final_chi2, rs.final_lambda)
            o2 = all(np.array_equal(x, y) for x, y in zip(c.get_window(), r.get_window()))
            o3 = np.array_equal(c.get_landmarks_xyz() if xyz else c.get_landmarks(), r.get_landmarks_xyz() if xyz else r.get_landmarks())
            if only >= 0 and not (o1 and o2 and o3):
                print("  window %d %s: report %s states %s landmarks %s (max |d pose| %.3e)" % (qi, kws[qi], o1, o2, o3, np.abs(c.get_window()[0] - r.get_window()[0]).max()))
            ok = ok and o1 and o2 and o3
        for _ in range(3):
            hip.batch_gn_iteration(batch, 3e5)
            for r in solo:
                r.gn_iteration(3e5)
        for qi, (c, r) in enumerate(zip(batch, solo)):
            # (a window without information goes non-finite in both — the reference's outcome too —: NaN equals NaN here)
            o4 = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(c.get_window(), r.get_window())) and np.array_equal(c.chi2(), r.chi2(), equal_nan=True)
            if only >= 0 and not o4:
                print("  window %d %s: after the batched GN iterations: max |d pose| %.3e chi2 %.17g / %.17g" % (qi, kws[qi], np.abs(c.get_window()[0] - r.get_window()[0]).max(), c.chi2(), r.chi2()))
            ok = ok and o4
    except vio.VioError as exc:
        print("  error:", exc)
        ok = False
    bad += 0 if ok else 1
    print("%s batch %2d: B=%d %s its=%d sizes=%s" % ("ok  " if ok else "FAIL", bi, B, "xyz" if xyz else "invdepth", its, [w.n_landmarks for w in ws]))
    del batch, solo, lead
print("failures:", bad)
