"""The chain-order pose solve (csrc/vio_pose_solve_chain.h) on its own: the kernel on caller-supplied systems against the numpy
model of the same algorithm (tools/chain_solve_model.py), against the 50-digit solutions of the reference's own system, and the
order switch of the ABI (vio_set_solve_order)."""
import os
import sys

import numpy as np
import pytest

import vio_testutil as tu
from conftest import GOLDEN_DIR, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import chain_solve_model as cm  # noqa: E402

pytestmark = pytest.mark.gpu


def backward_error(H, b, lam, x):
    A = (H + lam * np.eye(171)).astype(np.longdouble)
    r = b.astype(np.longdouble) - A @ x.astype(np.longdouble)
    den = np.abs(A) @ np.abs(x).astype(np.longdouble) + np.abs(b)
    ok = den > 0
    return float((np.abs(r[ok]) / den[ok]).max())


def test_kernel_against_the_model_block_by_block(hip_debug_lib):
    """every intermediate the kernel leaves in LDS (L tiles, M, pivots, forward-substituted right-hand side, solution) against
    the numpy model; random systems with the sparsity and the 1e16 scaling of a window's reduced system"""
    ctx = hip_debug_lib.context()
    rng = np.random.default_rng(7)
    for case in range(8):
        # cases 0..5: bias random walk of weight 1e16 beside entries of 1e4 (a window's scaling: the pivots of the chain are what
        # 1e16 - 1e16 leaves, so intermediates agree to the digits the cancellation keeps); 6, 7: a benign scaling, tight bounds
        benign = case >= 6
        H, b = cm.chain_pattern_system(rng, scale_bias=1e3 if benign else 1e16, with_prior=case % 2 == 0)
        if case == 4:                       # a fixed extrinsic: six zero rows that only lambda resolves
            H[:6, :] = 0.0; H[:, :6] = 0.0; b[:6] = 0.0
        if case == 5:                       # a skipped IMU edge (estimator.cpp:959-960): a speed-bias block without information
            idx = [12 + 15 * 10 + k for k in range(9)]
            H[idx, :] = 0.0; H[:, idx] = 0.0
        for lam in (5e5, 1e3, 1.0):
            x, dump = ctx.debug_chain_solve(H, b, lam, dump=True)
            m = cm.ChainModel(H, b, lam)
            diffs = m.compare_dump(dump)
            assert max(diffs.values()) <= (1e-11 if benign else 1e-5), (case, lam, diffs)
            assert np.abs(x - m.x).max() <= (1e-11 if benign else 1e-6) * max(np.abs(m.x).max(), 1e-300), (case, lam)
            assert backward_error(H, b, lam, x) <= 5e-15, (case, lam, backward_error(H, b, lam, x))
            if case == 4:
                assert np.all(x[:6] == 0.0)


def test_kernel_on_the_reference_system_against_exact_arithmetic(hip_debug_lib):
    """tests/golden/ldlt.npz (the reference's H_pp_schur, its Eigen LDLT vectors) and ldlt_exact.npz (50-digit solutions): the chain
    order is closer to the exact solution than Eigen's own vectors are, at every lambda"""
    z = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt.npz")))
    ze = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt_exact.npz")))
    ctx = hip_debug_lib.context()
    for i in range(3):
        lam = float(z["lambda_%d" % i])
        x = ctx.debug_chain_solve(z["Hs"], z["bs"], lam)
        m = cm.ChainModel(z["Hs"], z["bs"], lam)
        e_hip = np.abs(x - ze["x_exact_%d" % i]).max()
        e_model = np.abs(m.x - ze["x_exact_%d" % i]).max()
        e_eigen = np.abs(z["x_%d" % i] - ze["x_exact_%d" % i]).max()
        print("lambda %.3g: kernel %.2e, numpy model %.2e, Eigen %.2e from the exact solution" % (lam, e_hip, e_model, e_eigen))
        assert e_hip <= 0.5 * e_eigen and e_hip <= 20.0 * e_model, (i, e_hip, e_model, e_eigen)
        assert backward_error(z["Hs"], z["bs"], lam, x) <= 5e-15


def test_kernel_below_lambda_1_on_the_reference_system(hip_debug_lib):
    """The unpivoted chain order has no pivot check where Eigen's LDLT pivots.  tests/golden/ldlt.npz's H_pp_schur at lambda = 1e-3
    and 1e-6 — far below anything Solve reaches on a window with IMU factors (lambda >= 8 on the goldens) — is INDEFINITE: the
    reduced system's smallest eigenvalues are -0.57 .. -0.02, rounding of the 1e16-sized terms the Schur complement cancels
    (SURVEY.md section 7).  No solver has 'the' answer there (numpy's LU and the chain model differ by 2.0 in a solution of size 2..4);
    what can be asked: the kernel's result is finite, of the size of the other solvers' answers, and solves the system it was given:
    measured backward error 3.3e-16 at both lambdas (profiles/r05b_noimu_and_low_lambda.txt) — the numpy model of the same order,
    with numpy's quotients instead of the kernel's correctly rounded ones and no FMA, reaches 4e-13 there, LU with pivoting 1e-15."""
    z = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt.npz")))
    ctx = hip_debug_lib.context()
    for lam in (1e-3, 1e-6):
        x = ctx.debug_chain_solve(z["Hs"], z["bs"], lam)
        m = cm.ChainModel(z["Hs"], z["bs"], lam)
        be, bm = backward_error(z["Hs"], z["bs"], lam, x), backward_error(z["Hs"], z["bs"], lam, m.x)
        print("lambda %.0e: max|x| %.3f (model %.3f), backward error %.2e (model %.2e), |x - model| %.2e"
              % (lam, np.abs(x).max(), np.abs(m.x).max(), be, bm, np.abs(x - m.x).max()))
        assert np.isfinite(x).all() and np.abs(x).max() <= 10.0
        assert be <= 5e-15, (lam, be, bm)
        assert np.all(x[:6] == 0.0) or np.abs(z["bs"][:6]).max() > 0      # the fixed extrinsic's rows: 0 / lambda


def test_order_switch_and_fallback(vio, hip_lib):
    """vio_set_solve_order: the two orders on one window give the same step to the solvers' rounding; a prior that couples
    speed-bias blocks which are not neighbours falls back to Eigen's order by itself"""
    w = vio.synth.make_window(200, seed=21)
    ctx = hip_lib.context()
    assert ctx.get_solve_order() == (vio.capi.ORDER_CHAIN, vio.capi.ORDER_CHAIN) or os.environ.get("VIO_SOLVE_ORDER")
    out = {}
    for order in (vio.capi.ORDER_CHAIN, vio.capi.ORDER_EIGEN, vio.capi.ORDER_CHAIN):
        ctx.set_solve_order(order)
        ctx.load(w)
        out[order] = tu.run_stepwise(ctx)
        assert ctx.get_solve_order() == (order, order)
    a, b = out[vio.capi.ORDER_CHAIN], out[vio.capi.ORDER_EIGEN]
    assert np.array_equal(a["Hs"], b["Hs"]) and np.array_equal(a["bs"], b["bs"])
    assert np.abs(a["dx_pose"] - b["dx_pose"]).max() <= 1e-10 and np.abs(a["dx_lm"] - b["dx_lm"]).max() <= 1e-10
    # an artificial prior with a (sb_2, sb_7) entry
    H = np.zeros((156, 156))
    i, j = 12 + 15 * 2 + 1, 12 + 15 * 7 + 1
    H[i, i] = H[j, j] = 10.0
    H[i, j] = H[j, i] = 1.0
    w2 = w.copy()
    w2.prior = {"H": H, "b": np.zeros(156), "err": np.zeros(156), "jt_inv": np.zeros((156, 156))}
    ctx.set_solve_order(vio.capi.ORDER_CHAIN)
    ctx.load(w2)
    assert ctx.get_solve_order() == (vio.capi.ORDER_CHAIN, vio.capi.ORDER_EIGEN)
    c2 = hip_lib.context()
    c2.set_solve_order(vio.capi.ORDER_EIGEN)
    c2.load(w2)
    s1, s2 = tu.run_stepwise(ctx), tu.run_stepwise(c2)
    assert np.array_equal(s1["dx_pose"], s2["dx_pose"])
