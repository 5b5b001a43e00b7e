#!/usr/bin/env python3
"""Generates tests/golden/window_noimu_*.npz: whole windows WITHOUT IMU edges from the COMPILED REFERENCE
(oracle/_ref/libvio_ref.so) — fixtures that are reference code end to end.

Every other window fixture carries ten pre-integrations, and the IMU edge of the harness is `EdgeImuPort` (the oracle's
residual/Jacobians inside the reference's Problem: the reference's own edge_imu.cc needs a Ceres header the image lacks).
The reference itself produces graphs without IMU edges — Estimator::problemSolve skips the edge of an interval with
sum_dt > 10 (VM/src/estimator.cpp:956-970) and MargOldFrame does the same (:737) — and on such a graph no line of the
oracle runs inside the harness: vertices, EdgeReprojection, CauchyLoss, MakeHessian, Schur, LDLT, LM loop, Marginalize
are all the reference's.  lambda_0 = 1e-5 max diag is O(10) here instead of the 5e5 the IMU information pins, and Solve(10)
walks it further down: the regime where the unpivoted chain-order pose solve has no IMU information to lean on.

Run only in the container that mounts /root/reference:   python tests/golden/make_golden_noimu.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
import vio_testutil as tu  # noqa: E402

vio = load_package()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
ref = vio.VioLib(os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so"), "vior_")


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-36s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def solve_trace(w, kw, iterations=10):
    """Problem::Solve's loop (problem.cc:188-245) through the harness' single-step entry points: state, chi2, lambda, trials,
    prior vectors after every outer iteration (as tools/parity_trace.py --make-reference records them for solve_trace.npz)."""
    c = ref.context(**kw)
    c.load(w)
    c.linearize()
    chi, lam = c.init_lm()
    lam0 = lam
    st, chis, lams, trials, bps, eps = [], [], [], [], [], []
    last = 1e20
    for it in range(iterations):
        ok, false_cnt, t = False, 0, 0
        while not ok and false_cnt < 10:
            c.solve_linear(lam)
            c.update_states()
            ok, chi, lam = c.eval_step()
            t += 1
            if ok:
                c.linearize()
            else:
                false_cnt += 1
                c.rollback_states()
        p, s, e = c.get_window()
        st.append(np.concatenate([p.ravel(), s.ravel(), e.ravel(), c.get_landmarks()]))
        bp, ep = c.get_prior()
        chis.append(chi), lams.append(lam), trials.append(t), bps.append(bp), eps.append(ep)
        if last - chi < 1e-5:
            break
        last = chi
    return dict(trace_state=np.array(st), trace_chi=np.array(chis), trace_lam=np.array(lams), trace_lam0=np.float64(lam0),
                trace_trials=np.array(trials, dtype=np.int32), trace_bprior=np.array(bps), trace_errprior=np.array(eps))


def window_case(name, w, ext_fixed=1, marg=True):
    kw = dict(ext_fixed=ext_fixed)
    d = tu.window_to_arrays(w)
    d["cfg_ext_fixed"] = np.int32(ext_fixed)
    ctx = ref.context(**kw)
    ctx.load(w)
    step = tu.run_stepwise(ctx)
    d.update({"step_" + k: v for k, v in step.items()})
    ctx2 = ref.context(**kw)
    ctx2.load(w)
    sol, _ = tu.run_solve(ctx2, 10)
    d.update({"solve_" + k: v for k, v in sol.items()})
    d.update(solve_trace(w, kw))
    ws = w.copy()
    ws.poses, ws.speed_bias, ws.ext, ws.inv_depth = sol["posesF"], sol["sbF"], sol["extF"], sol["invdF"]
    if w.prior is not None:     # estimator.cpp:1040-1049: b/err prior come back updated, H/Jt stay
        ws.prior = dict(w.prior)
        ws.prior["b"] = sol["bpriorF"][:156].copy()
        ws.prior["err"] = sol["errpriorF"].copy()
    if marg:
        ctx3 = ref.context(**kw)
        ctx3.load(ws)
        m = ctx3.marginalize(vio.MARG_OLD)
        d.update({"marg0_" + k: v for k, v in m.items()})
        d.update(tu.window_to_arrays(ws, prefix="marg0_in_"))
    save(name, **d)
    return d


def no_imu(w):
    w.preint = [None] * 10
    return w


# N = 300 (the reference's real regime), every IMU edge skipped, no prior
dA = window_case("window_noimu_n300_s47", no_imu(vio.synth.make_window(300, seed=47)))
# the window one frame later with the prior the reference's MargOldFrame made of the first (IMU-less too)
wB = no_imu(vio.synth.make_window(300, seed=48, t0=1.1))
wB.prior = {k: dA["marg0_" + k] for k in tu.PRIOR_FIELDS}
window_case("window_noimu_n300_s48_prior", wB)
# ragged tracks (every (host, track length) pattern), extrinsic free
window_case("window_noimu_n150_s49_ragged_extfree", no_imu(vio.synth.make_window(150, seed=49, ragged=True)), ext_fixed=0)
print("done")
