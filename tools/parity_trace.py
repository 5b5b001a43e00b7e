#!/usr/bin/env python3
"""Where do the end-state differences of Solve(10) come from?  Per LM iteration: ||state - state_reference||_inf for the
oracle and for the HIP library on the golden windows, with lambda and chi2 beside it.

  python tools/parity_trace.py --make-reference    (where /root/reference exists: the compiled reference's per-iteration
                                                    states -> tests/golden/solve_trace.npz)
  python tools/parity_trace.py                     (GPU box: oracle + HIP against that file -> profiles/parity_trace.json)

The loop is Problem::Solve's (problem.cc:188-245) spelled out with the single-step entry points of the ABI, which every
library exports: linearize, init_lm, then per outer iteration up to 10 trials of solve_linear / update_states / eval_step
(+ rollback_states on a rejected step)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN_DIR, load_package  # noqa: E402
import vio_testutil as tu  # noqa: E402

vio = load_package()
FILES = [f for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "window_*.npz"))) if "solve_posesF" in np.load(f)]


def solve_trace(lib, w, kw, iterations=10):
    c = lib.context(**kw)
    c.load(w)
    c.linearize()
    chi, lam = c.init_lm()
    out, last = [], 1e20
    solve_trace.lam0 = lam
    for it in range(iterations):
        ok, false_cnt, trials = False, 0, 0
        while not ok and false_cnt < 10:
            c.solve_linear(lam)
            c.update_states()
            ok, chi, lam = c.eval_step()
            trials += 1
            if ok:
                c.linearize()
            else:
                false_cnt += 1
                c.rollback_states()
        p, s, e = c.get_window()
        lm = c.get_landmarks() if c.lm_dim == 1 else c.get_landmarks_xyz().ravel()
        bp, ep = c.get_prior()
        out.append(dict(state=np.concatenate([p.ravel(), s.ravel(), e.ravel(), lm]), chi=chi, lam=lam, trials=trials, bprior=bp, errprior=ep))
        if last - chi < 1e-5:
            break
        last = chi
    return out


def cfg_of(z):
    kw = {}
    if "cfg_ext_fixed" in z:
        kw["ext_fixed"] = int(z["cfg_ext_fixed"])
    if "cfg_loss_type" in z:
        kw["loss_type"] = int(z["cfg_loss_type"])
    if "cfg_loss_delta" in z:
        kw["loss_delta"] = float(z["cfg_loss_delta"])
    return kw


if "--make-reference" in sys.argv:
    ref = vio.VioLib(os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so"), "vior_")
    d = {}
    for f in FILES:
        z = dict(np.load(f))
        tr = solve_trace(ref, tu.arrays_to_window(vio, z), cfg_of(z))
        name = os.path.basename(f)[:-4]
        d[name + "_state"] = np.stack([t["state"] for t in tr])
        d[name + "_chi"] = np.array([t["chi"] for t in tr])
        d[name + "_lam"] = np.array([t["lam"] for t in tr])
        d[name + "_trials"] = np.array([t["trials"] for t in tr], dtype=np.int32)
        # what a restart from iteration k needs beside the states (tests/test_gpu_parity.py: every step from the reference's own state)
        d[name + "_lam0"] = np.float64(solve_trace.lam0)
        if "in_prior_H" in z:
            d[name + "_bprior"] = np.stack([t["bprior"] for t in tr])
            d[name + "_errprior"] = np.stack([t["errprior"] for t in tr])
        # the stepwise loop must be Problem::Solve: same end state as the golden file's Solve(10)
        endz = np.concatenate([z["solve_posesF"].ravel(), z["solve_sbF"].ravel(), z["solve_extF"].ravel(), z["solve_invdF"].ravel()])
        assert np.abs(tr[-1]["state"] - endz).max() == 0.0, (name, np.abs(tr[-1]["state"] - endz).max())
        print(name, len(tr), "iterations; end state identical to the golden Solve(10)")
    np.savez_compressed(os.path.join(GOLDEN_DIR, "solve_trace.npz"), **d)
    sys.exit(0)

zr = np.load(os.path.join(GOLDEN_DIR, "solve_trace.npz"))
orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")
libs = {"oracle": orc}
try:
    libs["hip"] = vio.load_hip()
    libs["hip"].context()
except Exception as exc:      # no GPU here: the oracle's columns only
    libs.pop("hip", None)
    print("HIP library not usable here (%s): oracle only" % exc)
report = {}
for f in FILES:
    z = dict(np.load(f))
    name = os.path.basename(f)[:-4]
    w, kw = tu.arrays_to_window(vio, z), cfg_of(z)
    rs, rl, rc = zr[name + "_state"], zr[name + "_lam"], zr[name + "_chi"]
    entry = {"lambda_reference": rl.tolist(), "chi2_reference": rc.tolist(), "trials_reference": zr[name + "_trials"].tolist()}
    for ln, lib in libs.items():
        tr = solve_trace(lib, w, kw)
        n = min(len(tr), len(rs))
        entry[ln] = {"iterations": len(tr),
                     "state_inf_diff": [float(np.abs(tr[i]["state"] - rs[i]).max()) for i in range(n)],
                     "lambda_rel_diff": [float(abs(tr[i]["lam"] - rl[i]) / rl[i]) for i in range(n)],
                     "chi2_rel_diff": [float(abs(tr[i]["chi"] - rc[i]) / rc[i]) for i in range(n)],
                     "trials": [int(t["trials"]) for t in tr]}
        print("%-38s %-6s" % (name, ln), " ".join("%.1e" % v for v in entry[ln]["state_inf_diff"]))
    print("%-38s lambda" % name, " ".join("%.1e" % v for v in rl))
    report[name] = entry
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(report, open(os.path.join(ROOT, "profiles", "parity_trace.json"), "w"), indent=1)
