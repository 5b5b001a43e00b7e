"""GPU parity tests: the HIP path (csrc/libvio_hip.so, called through the C ABI) against the CPU oracle on the
same seeded inputs, against the golden vectors of the compiled reference, and — at BASELINE.json's full sizes —
through size-independent properties.

Tolerance (north_star: "state delta within 1e-6 of the reference solve"): ||dx_hip - dx_oracle||_inf <= 1e-8 at
the reference's own lambda for every compared step (measured: ~2e-12), 1e-6 on the end state of Solve(10).
"""
import glob
import os

import numpy as np
import pytest

import vio_testutil as tu
from conftest import GOLDEN_DIR
from test_oracle_golden import check_prior, check_window_against_golden

pytestmark = pytest.mark.gpu
WINDOW_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "window_*.npz")))


def compare_stepwise(a, b, dx_tol=1e-8):
    """a: HIP, b: oracle."""
    assert tu.scaled_sym_err(a["Hs"], b["Hs"]) <= 1e-9
    assert np.abs(a["bs"] - b["bs"]).max() <= 1e-10 * max(np.abs(b["bs"]).max(), 1e-300)
    assert np.abs(a["bpp"] - b["bpp"]).max() <= 1e-10 * max(np.abs(b["bpp"]).max(), 1e-300)
    assert tu.rel_max(a["diag"], b["diag"]) <= 1e-11
    assert tu.rel_max(a["hll"], b["hll"]) <= 1e-10 and tu.rel_max(a["bl"], b["bl"]) <= 1e-9
    assert abs(a["chi0"] - b["chi0"]) <= 1e-10 * abs(b["chi0"])
    assert a["lambda0"] == b["lambda0"]
    assert np.abs(a["dx_pose"] - b["dx_pose"]).max() <= dx_tol
    assert np.abs(a["dx_lm"] - b["dx_lm"]).max() <= dx_tol
    for k in ("poses1", "sb1", "ext1", "invd1"):
        assert np.abs(a[k] - b[k]).max() <= dx_tol, k
    assert abs(a["chi1"] - b["chi1"]) <= 1e-8 * abs(b["chi1"])
    assert int(a["accepted"]) == int(b["accepted"])
    assert abs(a["lambda1"] - b["lambda1"]) <= 1e-9 * abs(b["lambda1"])


@pytest.mark.parametrize("n,seed,ragged,ext_fixed,loss", [
    (1, 1, False, 1, 2),          # one landmark, four edges
    (9, 2, True, 1, 2),           # ragged tracks of length 1..10
    (64, 3, False, 1, 2),         # exactly one full item per pattern boundary
    (65, 4, False, 0, 2),         # extrinsic free, item tail of one landmark
    (300, 5, True, 0, 2),         # every (host, track-length) pattern, extrinsic free
    (1000, 6, True, 1, 2),
    (300, 7, False, 1, 0),        # no loss object
    (300, 8, False, 1, 3),        # Tukey
])
def test_stepwise_against_oracle(vio, oracle_lib, hip_lib, n, seed, ragged, ext_fixed, loss):
    kw = {}
    if loss == 3:
        kw = dict(pos_noise=0.001, rot_noise=0.0002, depth_noise=0.003, pixel_noise=0.25 / 460, outlier_fraction=0.05)
    w = vio.synth.make_window(n, seed=seed, ragged=ragged, **kw)
    ch, co = hip_lib.context(ext_fixed=ext_fixed, loss_type=loss), oracle_lib.context(ext_fixed=ext_fixed, loss_type=loss)
    ch.load(w)
    co.load(w)
    compare_stepwise(tu.run_stepwise(ch), tu.run_stepwise(co))


@pytest.mark.parametrize("path", WINDOW_FILES, ids=[os.path.basename(p)[:-4] for p in WINDOW_FILES])
def test_against_reference_golden_vectors(vio, hip_lib, path):
    """The same check the oracle passes on CPU, with the HIP library in its place."""
    # end state of Solve(10): the difference to the reference starts at 1e-12...4e-12 (first step, lambda = 1.7e5) and grows smoothly as lambda
    # walks down to O(10..100) and cond(H + lambda I) up to 1e14..1e15 — no step jumps (profiles/parity_trace.json, tools/parity_trace.py).
    # Round 4: with the chain-order solve (3-25 x closer to the exact solution than Eigen's LDLT at these lambdas, profiles/parity_exact.json)
    # the largest end value is 4.9e-7 (window_n300_s43; window_n300_s45_prior, 1.3e-6 under the pivoted kernel, ends at 2.8e-7): what is
    # left is the reference's own rounding, and the north star's 1e-6 holds on every golden window.
    check_window_against_golden(vio, hip_lib, path, dx_tol=1e-8, state_tol=1e-6, lambda_rtol=3e-4)


@pytest.mark.parametrize("path", WINDOW_FILES, ids=[os.path.basename(p)[:-4] for p in WINDOW_FILES])
def test_golden_vectors_with_half_width_workgroups(vio, hip_lib, path):
    """The same fixtures through the plans of the throughput policy (vio_config.item_policy = VIO_ITEMS_THROUGHPUT): k_linearize_h,
    512 threads, two workgroups to a CU, the largest items half the LDS holds — another grouping of the same sums."""
    check_window_against_golden(vio, hip_lib, path, dx_tol=1e-8, state_tol=1e-6, lambda_rtol=3e-4, extra_cfg={"item_policy": vio.capi.ITEMS_THROUGHPUT})


def test_solve_trace_against_the_reference_iteration_by_iteration(vio, hip_lib):
    """tests/golden/solve_trace.npz: the compiled reference's state, lambda and chi2 after every outer iteration of
    Solve(10) at full precision (its printout has 6 digits).  HIP, through the single-step entry points."""
    zr = np.load(os.path.join(GOLDEN_DIR, "solve_trace.npz"))
    from test_oracle_golden import cfg_of, solve_trace_stepwise
    for path in WINDOW_FILES:
        z = dict(np.load(path))
        name = os.path.basename(path)[:-4]
        if name + "_state" not in zr:
            continue
        tr = solve_trace_stepwise(hip_lib, tu.arrays_to_window(vio, z), cfg_of(z))
        rs, rl = zr[name + "_state"], zr[name + "_lam"]
        assert len(tr) == len(rs) and [t[3] for t in tr] == list(zr[name + "_trials"]), name
        d = [float(np.abs(t[0] - r).max()) for t, r in zip(tr, rs)]
        assert d[0] <= 1e-10 and max(d) <= 1e-6, (name, d)
        assert max(abs(t[2] - l) / l for t, l in zip(tr, rl)) <= 3e-4, name


@pytest.mark.parametrize("path", [p for p in WINDOW_FILES if "window_noimu_" in p], ids=lambda p: os.path.basename(p)[:-4])
def test_imu_less_windows_iteration_by_iteration(vio, hip_lib, path):
    """Windows without IMU edges, generated by the compiled reference with nothing of the oracle in the graph (make_golden_noimu.py):
    Solve(10) iteration by iteration — lambda_0 = O(10..100) walking down to O(1), the speed-bias blocks carrying lambda (and the
    prior) alone — for both elimination orders of the pose solve and both item policies.  Same trial counts as the reference; state
    within 1e-8 of the reference's after every outer iteration (measured: see profiles/r05_noimu_parity.json)."""
    from test_oracle_golden import check_noimu_trace
    z = dict(np.load(path))
    for order in (vio.capi.ORDER_CHAIN, vio.capi.ORDER_EIGEN):
        d = check_noimu_trace(vio, hip_lib, path, 1e-8, order=order)
        print(os.path.basename(path), "order", order, "max |state - reference| per outer iteration:", ["%.1e" % v for v in d])
    for policy in (vio.capi.ITEMS_THROUGHPUT,):
        w, kw = tu.arrays_to_window(vio, z), {"ext_fixed": int(z["cfg_ext_fixed"]), "item_policy": policy}
        c = hip_lib.context(**kw)
        c.load(w)
        sol, rep = tu.run_solve(c, 10)
        assert int(sol["iterations"]) == int(z["solve_iterations"])
        for k in ("posesF", "sbF", "extF", "invdF"):
            assert np.abs(sol[k] - z["solve_" + k]).max() <= 1e-8, k


def restart_states(vio, z, zr, name, k):
    """the window of fixture z with the reference's states (and prior vectors) after outer iteration k of its Solve(10); k = 0: the input"""
    w = tu.arrays_to_window(vio, z)
    if k == 0:
        return w
    st = zr[name + "_state"][k - 1]
    w.poses, w.speed_bias, w.ext = st[:77].reshape(11, 7).copy(), st[77:176].reshape(11, 9).copy(), st[176:183].copy()
    if getattr(w, "xyz", None) is not None:
        w.xyz = st[183:].reshape(-1, 3).copy()
    else:
        w.inv_depth = st[183:].copy()
    if w.prior is not None:
        w.prior = dict(w.prior)
        w.prior["b"], w.prior["err"] = zr[name + "_bprior"][k - 1][:156].copy(), zr[name + "_errprior"][k - 1].copy()
    return w


def per_step_differences(vio, lib, order=None):
    """SURVEY.md section 7 states the tolerance per trial step at the reference's own lambda sequence: for every golden window and
    every outer iteration k of the reference's Solve(10) (tests/golden/solve_trace.npz holds its state, prior vectors and lambda
    after each), restart from the REFERENCE's state k, take the one accepted trial of iteration k + 1 at the reference's lambda
    (after t - 1 rejections lambda has been multiplied by 2, 4, 8, ...: problem.cc:569-570) and compare the state it leads to with
    the reference's own state k + 1."""
    from test_oracle_golden import cfg_of
    zr = np.load(os.path.join(GOLDEN_DIR, "solve_trace.npz"))
    out = {}
    for path in WINDOW_FILES:
        z = dict(np.load(path))
        name = os.path.basename(path)[:-4]
        if name + "_state" not in zr:
            continue
        c = lib.context(**cfg_of(z))
        if order is not None:
            c.set_solve_order(order)
        rs, rl, rt = zr[name + "_state"], zr[name + "_lam"], zr[name + "_trials"]
        d = []
        for k in range(len(rs)):
            lam = float(zr[name + "_lam0"]) if k == 0 else float(rl[k - 1])
            t = int(rt[k])
            if t > 10 or (t == 10 and k + 1 == len(rs) and np.array_equal(rs[k], rs[k - 1] if k else rs[k])):
                continue                                        # (ten rejections: the state did not move)
            lam *= 2.0 ** ((t - 1) * t // 2)
            c.load(restart_states(vio, z, zr, name, k))
            c.linearize()
            c.solve_linear(lam)
            c.update_states()
            p, s, e = c.get_window()
            lm = c.get_landmarks() if c.lm_dim == 1 else c.get_landmarks_xyz().ravel()
            d.append(float(np.abs(np.concatenate([p.ravel(), s.ravel(), e.ravel(), lm]) - rs[k]).max()))
        out[name] = d
    return out


def test_every_step_from_the_reference_state_within_1e_6(vio, hip_lib):
    """the per-step figure of SURVEY.md section 7 (1e-6 per trial step at the reference's lambda sequence), for both elimination
    orders of the pose solve; the accumulated end state of Solve(10) is bounded by the tests above"""
    for order in (vio.capi.ORDER_CHAIN, vio.capi.ORDER_EIGEN):
        d = per_step_differences(vio, hip_lib, order)
        assert len(d) >= 8
        worst = {k: max(v) for k, v in d.items()}
        assert max(worst.values()) <= 1e-6, (order, worst)
        assert all(v[0] <= 1e-9 for v in d.values()), (order, {k: v[0] for k, v in d.items()})


@pytest.mark.parametrize("n,seed,ragged,ext_fixed", [(50, 11, False, 1), (300, 12, True, 1), (400, 13, False, 0), (2000, 14, False, 1)])
def test_full_solve_against_oracle(vio, oracle_lib, hip_lib, n, seed, ragged, ext_fixed):
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    ch, co = hip_lib.context(ext_fixed=ext_fixed), oracle_lib.context(ext_fixed=ext_fixed)
    ch.load(w)
    co.load(w)
    sh, rh = tu.run_solve(ch)
    so, ro = tu.run_solve(co)
    assert rh.iterations == ro.iterations and rh.trials == ro.trials and rh.accepted == ro.accepted
    np.testing.assert_allclose(sh["chi2_trace"], so["chi2_trace"], rtol=1e-6)
    # the Nielsen factor 1-(2 rho-1)^3 amplifies rounding once chi2 stops changing (see test_oracle_golden.py)
    np.testing.assert_allclose(sh["lambda_trace"], so["lambda_trace"], rtol=2e-3)
    assert abs(rh.final_chi2 - ro.final_chi2) <= 1e-6 * ro.final_chi2
    for k in ("posesF", "sbF", "extF", "invdF"):
        assert np.abs(sh[k] - so[k]).max() <= 1e-6, k
    assert rh.solve_ms > 0 and rh.hessian_ms > 0


@pytest.mark.parametrize("n,seed,ragged,ext_fixed", [(300, 12, True, 1), (400, 13, False, 0), (150, 300, True, 1)])
def test_lambda_sequence_against_the_oracle(vio, oracle_lib, hip_lib, n, seed, ragged, ext_fixed):
    """IsGoodStepInLM's factor 1 - (2 rho - 1)^3 (problem.cc:561) is formed on the device as 1 - (t t) t, two products, where the oracle (and
    the reference) call pow: at most an ulp apart per step, and the step feeds lambda.  Over the ten outer iterations of
    Solve(10) the lambda sequence must stay within 1e-6 of the oracle's (measured: 0 on most windows, 3e-8 at worst,
    tools/diag_lambda_deviation.py) — the looser 2e-3 of the full-solve tests covers windows whose chi2 has stopped changing."""
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    ch, co = hip_lib.context(ext_fixed=ext_fixed), oracle_lib.context(ext_fixed=ext_fixed)
    ch.load(w)
    co.load(w)
    sh, rh = tu.run_solve(ch)
    so, ro = tu.run_solve(co)
    assert rh.iterations == ro.iterations == 10 and rh.trials == ro.trials
    np.testing.assert_allclose(sh["lambda_trace"], so["lambda_trace"], rtol=1e-6)


def test_solve_with_prior_and_marginalisation_chain(vio, oracle_lib, hip_lib):
    """Three consecutive windows: solve, MargOldFrame, shift, solve with the prior, MargNewFrame — the sequence of
    Estimator::backendOptimization (estimator.cpp:1075-1141), HIP and oracle side by side."""
    w = vio.synth.make_window(250, seed=31)
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(w)
    co.load(w)
    sh, _ = tu.run_solve(ch)
    so, _ = tu.run_solve(co)
    ws = w.copy()
    ws.poses, ws.speed_bias, ws.ext, ws.inv_depth = so["posesF"], so["sbF"], so["extF"], so["invdF"]
    ch.load(ws)
    co.load(ws)
    mh, mo = ch.marginalize(vio.MARG_OLD), co.marginalize(vio.MARG_OLD)
    check_prior(mh, mo)
    w2 = vio.synth.make_window(250, seed=32, t0=1.1)
    w2.prior = mo
    ch.load(w2)
    co.load(w2)
    compare_stepwise(tu.run_stepwise(ch), tu.run_stepwise(co))
    ch.load(w2)
    co.load(w2)
    sh, rh = tu.run_solve(ch)
    so, ro = tu.run_solve(co)
    assert rh.iterations == ro.iterations
    assert np.abs(sh["posesF"] - so["posesF"]).max() <= 1e-6
    assert tu.rel_max(sh["bpriorF"], so["bpriorF"]) <= 1e-7 and tu.rel_max(sh["errpriorF"], so["errpriorF"]) <= 1e-6
    w3 = w2.copy()
    w3.poses, w3.speed_bias, w3.ext, w3.inv_depth = so["posesF"], so["sbF"], so["extF"], so["invdF"]
    w3.prior = dict(mo)
    w3.prior["b"], w3.prior["err"] = so["bpriorF"][:156].copy(), so["errpriorF"].copy()
    ch.load(w3)
    co.load(w3)
    check_prior(ch.marginalize(vio.MARG_SECOND_NEW), co.marginalize(vio.MARG_SECOND_NEW))


def test_damped_solve_against_eigen_ldlt_vectors(vio, hip_lib):
    """(H_pp_schur + lambda I)^-1 b at lambda = lambda_0, 1e3, 1 (cond 3e10 / 1.6e13 / 2.7e16) against
    (i)  Eigen::LDLT run on the reference's own matrix (tests/golden/ldlt.npz): bounds = 10 x the measured differences
         2.95e-12 / 4.12e-8 / 5.07e-5 (profiles/parity_split.json; all of it the solver's share: the HIP system itself is within
         6e-11 of the oracle's in delta_x terms).  These are two double-precision solvers disagreeing inside the rounding ball of
         an ill-scaled system:
    (ii) the EXACT solution of that system (tests/golden/ldlt_exact.npz: 50 digits): Eigen's own vectors are 4.1e-12 / 1.2e-7 /
         1.4e-4 away from it.  The criterion with an answer: the HIP solve is as close to the truth as Eigen's is, and
    (iii) its component-wise backward error |b - (H + lambda I) x| / (|H + lambda I| |x| + |b|), residual in extended
         precision, is at rounding level (measured 3e-16 .. 1.4e-15 on the golden windows; Eigen's: 6e-16 .. 1.9e-15)."""
    z = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt.npz")))
    ze = dict(np.load(os.path.join(GOLDEN_DIR, "ldlt_exact.npz")))
    zw = dict(np.load(os.path.join(GOLDEN_DIR, "window_n50_s42.npz")))
    w = tu.arrays_to_window(vio, zw)
    ctx = hip_lib.context()
    ctx.load(w)
    ctx.linearize()
    H, b = ctx.get_schur_system()
    for i, tol in enumerate((3e-11, 4.2e-7, 5.1e-4)):
        lam = float(z["lambda_%d" % i])
        ctx.solve_linear(lam)
        dx, _ = ctx.get_delta()
        err = np.abs(dx - z["x_%d" % i]).max()
        assert err <= tol, (i, err)
        e_hip, e_eigen = np.abs(dx - ze["x_exact_%d" % i]).max(), np.abs(z["x_%d" % i] - ze["x_exact_%d" % i]).max()
        assert e_hip <= 3.0 * e_eigen, (i, e_hip, e_eigen)
        A = (H + lam * np.eye(171)).astype(np.longdouble)
        r = b.astype(np.longdouble) - A @ dx.astype(np.longdouble)
        den = np.abs(A) @ np.abs(dx).astype(np.longdouble) + np.abs(b)
        ok = den > 0                                   # (the rows of the fixed extrinsic: 0 = 0)
        assert float((np.abs(r[ok]) / den[ok]).max()) <= 5e-15 and float(np.abs(r[~ok]).max() if (~ok).any() else 0.0) == 0.0, (i, float((np.abs(r[ok]) / den[ok]).max()))


def test_initial_lambda_sees_the_landmark_diagonal_under_both_item_policies(vio, oracle_lib, hip_lib):
    """ComputeLambdaInitLM takes the maximum over the WHOLE diagonal, landmarks included (problem.cc:511-516).  With IMU factors their
    information (capped at 5e10) decides; without them the largest h_ll does — which the half-width kernels of the throughput policy
    dropped until round 4 (the wave that held it never ran the reduction: ADVICE r03)."""
    for n, seed in ((6, 17), (8, 3), (10, 5)):
        # few, far landmarks, no IMU factors, no loss: the largest diagonal entry of the Hessian is a landmark's
        w = vio.synth.make_window(n, seed=seed)
        w.preint = [None] * 10
        w.inv_depth = w.inv_depth * (0.2 if n == 10 else 0.1)
        co = oracle_lib.context(loss_type=0)
        co.load(w)
        co.linearize()
        chi_o, lam_o = co.init_lm()
        hll, _ = co.get_landmark_system()
        _, diag = co.get_pose_gradient()
        assert np.abs(hll).max() > np.abs(diag).max() and abs(lam_o - 1e-5 * np.abs(hll).max()) <= 1e-12 * lam_o
        for policy in (vio.capi.ITEMS_LATENCY, vio.capi.ITEMS_THROUGHPUT):
            ch = hip_lib.context(item_policy=policy, loss_type=0)
            ch.load(w)
            ch.linearize()
            chi_h, lam_h = ch.init_lm()
            assert abs(lam_h - lam_o) <= 1e-12 * lam_o, (n, policy, lam_h, lam_o)
            assert abs(chi_h - chi_o) <= 1e-9 * chi_o


def test_imu_only_and_missing_edges(vio, oracle_lib, hip_lib):
    w = vio.synth.make_window(0, seed=5)
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(w)
    co.load(w)
    ch.linearize()
    co.linearize()
    (ca, la), (cb, lb) = ch.init_lm(), co.init_lm()
    assert abs(ca - cb) <= 1e-9 * max(abs(cb), 1e-12) and la == lb
    ch.solve_linear(la)
    co.solve_linear(lb)
    assert np.abs(ch.get_delta()[0] - co.get_delta()[0]).max() <= 1e-8
    w = vio.synth.make_window(80, seed=6)
    w.preint[4] = None          # estimator.cpp:959-960 skips an IMU edge with sum_dt > 10
    ch.load(w)
    co.load(w)
    compare_stepwise(tu.run_stepwise(ch), tu.run_stepwise(co))


def test_error_paths(vio, hip_lib):
    ctx = hip_lib.context()
    w = vio.synth.make_window(0, seed=5)
    w.preint = [None] * 10
    ctx.load(w)
    with pytest.raises(vio.VioError) as e:
        ctx.solve(10)           # Problem::Solve returns false on an empty graph (problem.cc:172-175)
    assert e.value.status == -4
    w = vio.synth.make_window(10, seed=1)
    ctx.set_landmarks(w.inv_depth)
    with pytest.raises(vio.VioError):
        ctx.set_observations(w.lm + 100, w.host, w.target, w.pts_i, w.pts_j)      # landmark index out of range
    bad = w.copy()
    bad.host = bad.host.copy()
    bad.host[1] = (bad.host[1] + 5) % 11       # two edges of one landmark with different hosts
    bad.target = bad.target.copy()
    bad.target[1] = (bad.host[1] + 1) % 11
    ctx2 = hip_lib.context()
    ctx2.load(bad)
    with pytest.raises(vio.VioError) as e:
        ctx2.linearize()
    assert e.value.status == -5


def test_observation_list_in_any_order(vio, oracle_lib, hip_lib):
    """The reference's loop emits the edges landmark by landmark (estimator.cpp:975-1016) and the library has a fast path for such
    lists (the list is its own CSR, the consistency of a landmark's edges is checked while they are copied, the observations go to the
    device as listed and a kernel puts them into item order); the ABI takes any order.  The same window listed observation-index-major
    (every landmark's edges far apart, their order inside a landmark kept) gives the same bits; a fully shuffled list (another order
    of a landmark's edges = another pattern, another summation order) the same step to rounding; both agree with the oracle."""
    w = vio.synth.make_window(700, seed=31, ragged=True)
    first = np.concatenate([[0], np.cumsum(np.bincount(w.lm, minlength=w.n_landmarks))[:-1]])
    kidx = np.arange(w.n_observations) - first[w.lm]
    rng = np.random.RandomState(3)

    def relisted(order):
        v = w.copy()
        for f in ("lm", "host", "target", "pts_i", "pts_j"):
            setattr(v, f, np.ascontiguousarray(getattr(w, f)[order]))
        return v

    def step(lib, win):
        c = lib.context()
        c.load(win)
        c.linearize()
        _, lam = c.init_lm()
        c.solve_linear(lam)
        dx, dl = c.get_delta()
        return c.chi2(), lam, dx, dl

    base = step(hip_lib, w)
    kmajor = step(hip_lib, relisted(np.lexsort((w.lm, kidx))))
    assert kmajor[0] == base[0] and kmajor[1] == base[1] and np.array_equal(kmajor[2], base[2]) and np.array_equal(kmajor[3], base[3])
    shuffled = relisted(rng.permutation(w.n_observations))
    sh, so = step(hip_lib, shuffled), step(oracle_lib, shuffled)
    assert abs(sh[0] - base[0]) <= 1e-9 * base[0]
    np.testing.assert_allclose(sh[2], base[2], rtol=0, atol=1e-9)
    np.testing.assert_allclose(sh[2], so[2], rtol=0, atol=1e-8)
    np.testing.assert_allclose(sh[3], so[3], rtol=0, atol=1e-8 * max(1.0, np.abs(so[3]).max()))
    # a landmark whose edges disagree about its host observation: refused whatever the order of the list
    for order in (np.arange(w.n_observations), np.lexsort((w.lm, kidx))):
        bad = relisted(order)
        two = int(np.nonzero(np.bincount(w.lm) >= 2)[0][3])          # a landmark with at least two edges
        e = int(np.nonzero(bad.lm == two)[0][-1])
        bad.pts_i = bad.pts_i.copy()
        bad.pts_i[e, 0] += 1e-3
        c = hip_lib.context()
        c.load(bad)
        with pytest.raises(vio.VioError) as err:
            c.linearize()
        assert err.value.status == -5


def test_observation_list_written_in_place_on_the_gpu(vio, hip_lib):
    """vio_map_observations hands out the HIP library's own arrays (pts_j: the pinned buffer its upload leaves from); filled in place and
    committed they give the bits vio_set_observations gives, frame after frame on one context (the arrays are re-used), also when a
    frame has more or fewer edges than the one before; between map and commit the context holds no list."""
    ws = [vio.synth.make_window(n, seed=60 + i, ragged=(i % 2 == 1)) for i, n in enumerate((400, 900, 250))]
    a, b = hip_lib.context(), hip_lib.context()
    for w in ws:
        a.load(w)
        b.set_window(w.poses, w.speed_bias, w.ext)
        b.set_landmarks(w.inv_depth)
        lm, host, target, pi, pj = b.map_observations(w.n_observations)
        lm[:], host[:], target[:], pi[:], pj[:] = w.lm, w.host, w.target, w.pts_i, w.pts_j
        with pytest.raises(vio.VioError):
            b.linearize()
        b.commit_observations()
        for k, p in enumerate(w.preint):
            b.set_imu(k, p)
        b.set_prior(None)
        ra, rb = a.solve(10), b.solve(10)
        assert ra.final_chi2 == rb.final_chi2 and ra.iterations == rb.iterations
        assert np.array_equal(a.get_landmarks(), b.get_landmarks())
        assert np.array_equal(a.get_window()[0], b.get_window()[0])
    with pytest.raises(vio.VioError):
        b.commit_observations()
    lm, host, target, pi, pj = b.map_observations(2)
    lm[:], host[:], target[:] = (0, 1), (0, 3), (1, 3)          # host == target
    with pytest.raises(vio.VioError):
        b.commit_observations()


def test_mapping_protocol_on_the_gpu(vio, hip_lib):
    tu.check_mapping_protocol(vio, hip_lib, pytest)


def test_rollback_restores_the_states(vio, hip_lib):
    w = vio.synth.make_window(120, seed=9)
    ctx = hip_lib.context()
    ctx.load(w)
    ctx.linearize()
    _, lam = ctx.init_lm()
    ctx.solve_linear(lam)
    ctx.update_states()
    p1, _, _ = ctx.get_window()
    assert np.abs(p1 - w.poses).max() > 1e-6
    ctx.rollback_states()
    p0, s0, e0 = ctx.get_window()
    np.testing.assert_array_equal(p0, w.poses)
    np.testing.assert_array_equal(s0, w.speed_bias)
    np.testing.assert_array_equal(ctx.get_landmarks(), w.inv_depth)


# ---- BASELINE.json's full sizes: properties that do not need the oracle --------------------------------------
@pytest.fixture(scope="module")
def big_window(vio):
    return vio.synth.make_window(20000, seed=42)


def test_headline_window_is_bitwise_reproducible(vio, hip_lib, big_window):
    """Fixed reduction order everywhere (no float atomics): two independent contexts give identical bits."""
    outs = []
    for _ in range(2):
        ctx = hip_lib.context()
        ctx.load(big_window)
        ctx.linearize()
        chi0, lam = ctx.init_lm()
        ctx.solve_linear(lam)
        outs.append((ctx.get_schur_system(), ctx.get_delta(), chi0))
    np.testing.assert_array_equal(outs[0][0][0], outs[1][0][0])
    np.testing.assert_array_equal(outs[0][1][0], outs[1][1][0])
    np.testing.assert_array_equal(outs[0][1][1], outs[1][1][1])
    assert outs[0][2] == outs[1][2]


def test_headline_window_under_both_item_policies(vio, hip_lib, big_window):
    """The same 20 000-landmark window cut into items the latency way (82 landmarks, 1024-thread workgroups) and the throughput way
    (56, half-width workgroups, two to a CU): another grouping of the same sums — the reduced systems agree to rounding, Solve(10)
    takes the same decisions, and each is bitwise reproducible."""
    res = []
    for pol in (vio.capi.ITEMS_LATENCY, vio.capi.ITEMS_THROUGHPUT, vio.capi.ITEMS_THROUGHPUT):
        ctx = hip_lib.context(item_policy=pol)
        ctx.load(big_window)
        ctx.linearize()
        H, b = ctx.get_schur_system()
        rep = ctx.solve(10)
        res.append((H, b, rep.iterations, rep.trials, rep.final_chi2, ctx.get_window()[0], ctx.get_landmarks()))
    (H0, b0, it0, tr0, c0, p0, l0), (H1, b1, it1, tr1, c1, p1, l1), (H2, b2, it2, tr2, c2, p2, l2) = res
    assert tu.scaled_sym_err(H1, H0) <= 1e-11 and np.abs(b1 - b0).max() <= 1e-11 * np.abs(b0).max()
    assert (it1, tr1) == (it0, tr0) and abs(c1 - c0) <= 1e-9 * c0
    assert np.abs(p1 - p0).max() <= 1e-8 and np.abs(l1 - l0).max() <= 1e-7
    np.testing.assert_array_equal(H1, H2)
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(l1, l2)
    assert (it1, tr1, c1) == (it2, tr2, c2)


def test_headline_window_linearity_over_landmark_shards(vio, hip_lib, big_window):
    """The reduced visual system is a sum over landmarks: the two half-window systems add up to the full one.
    (This is the property the multi-GPU all-reduce relies on.)"""
    cam = tu.CAM_IDX
    full = hip_lib.context()
    full.load(big_window)
    full.linearize()
    Hf, bf = full.get_schur_system()
    nov = big_window.copy()
    nov_parts = []
    for r in range(2):
        s = vio.synth.shard_window(big_window, r, 2)
        c = hip_lib.context()
        c.load(s)
        c.linearize()
        nov_parts.append(c.get_schur_system())
    # IMU + prior are replicated on every shard: subtract one copy of the non-visual part
    empty = big_window.copy()
    empty.inv_depth, empty.lm = empty.inv_depth[:0], empty.lm[:0]
    empty.host, empty.target, empty.pts_i, empty.pts_j = empty.host[:0], empty.target[:0], empty.pts_i[:0], empty.pts_j[:0]
    c0 = hip_lib.context()
    c0.load(empty)
    c0.linearize()
    H0, b0 = c0.get_schur_system()
    Hsum = nov_parts[0][0] + nov_parts[1][0] - H0
    bsum = nov_parts[0][1] + nov_parts[1][1] - b0
    assert tu.scaled_sym_err(Hsum, Hf) <= 1e-10
    assert np.abs(bsum - bf).max() <= 1e-10 * np.abs(bf).max()
    assert np.abs(Hf - Hf.T)[np.ix_(cam, cam)].max() <= 1e-9 * np.abs(Hf[np.ix_(cam, cam)]).max()


def test_headline_window_solve_converges_and_matches_oracle_step(vio, oracle_lib, hip_lib, big_window):
    """N = 20 000 / M = 80 000 (BASELINE.json configs[2]).  The sparse oracle finishes one step in well under a
    second; the full LM run must decrease chi2 monotonically over its accepted steps."""
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(big_window)
    co.load(big_window)
    a, b = tu.run_stepwise(ch), tu.run_stepwise(co)
    compare_stepwise(a, b)
    ch.load(big_window)
    sol, rep = tu.run_solve(ch)
    tr = sol["chi2_trace"]
    assert rep.accepted >= 3 and np.all(np.diff(tr) <= 0) and rep.final_chi2 < 0.05 * rep.initial_chi2
    # ground truth is known for synthetic windows: the solve must move the poses towards it
    err0 = np.abs(big_window.poses[:, :3] - big_window.poses_gt[:, :3]).max()
    errF = np.abs((sol["posesF"][:, :3] - sol["posesF"][0, :3]) - (big_window.poses_gt[:, :3] - big_window.poses_gt[0, :3])).max()
    assert errF < err0


def test_200k_landmarks_in_eight_shards(vio, oracle_lib, hip_lib):
    """BASELINE.json configs[3]: 200 000 landmarks / 800 000 observations, 25 000 per GPU on eight GPUs.  One GPU is what a
    test box has, so: (a) the whole window on it, one LM step against the CPU restatement (the oracle's sparse Schur
    finishes a step of this size in a few seconds; the reference's dense solver would need 320 GB); (b) the eight 25 000-landmark
    shards of the exchange protocol one after the other on the same device — their reduced systems, the payload of the
    all-reduce, add up to the whole window's (minus the seven extra copies of the replicated IMU + prior part), and so do their
    chi2 sums."""
    w = vio.synth.make_window(200000, seed=44)
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(w)
    co.load(w)
    a, b = tu.run_stepwise(ch), tu.run_stepwise(co)
    compare_stepwise(a, b)
    Hf, bf, chif = a["Hs"], a["bs"], float(a["chi0"])
    empty = w.copy()
    empty.inv_depth, empty.lm = empty.inv_depth[:0], empty.lm[:0]
    empty.host, empty.target, empty.pts_i, empty.pts_j = empty.host[:0], empty.target[:0], empty.pts_i[:0], empty.pts_j[:0]
    c0 = hip_lib.context()
    c0.load(empty)
    c0.linearize()
    H0, b0 = c0.get_schur_system()
    chi00, _ = c0.init_lm()
    Hsum, bsum, chisum, n_seen = -7.0 * H0, -7.0 * b0, -7.0 * chi00, 0
    for r in range(8):
        s = vio.synth.shard_window(w, r, 8)
        assert s.n_landmarks == 25000
        n_seen += s.n_landmarks
        c = hip_lib.context()
        c.load(s)
        c.linearize()
        Hs, bs = c.get_schur_system()
        chi, _ = c.init_lm()
        Hsum += Hs
        bsum += bs
        chisum += chi
        del c
    assert n_seen == 200000
    assert tu.scaled_sym_err(Hsum, Hf) <= 1e-10
    assert np.abs(bsum - bf).max() <= 1e-10 * np.abs(bf).max()
    assert abs(chisum - chif) <= 1e-10 * chif


def test_gn_iteration_is_the_same_arithmetic_as_the_steps(vio, hip_lib):
    """vio_gn_iteration (what bench.py times) = linearize + solve_linear + update at a fixed lambda."""
    w = vio.synth.make_window(500, seed=17)
    a, b = hip_lib.context(), hip_lib.context()
    a.load(w)
    b.load(w)
    lam = 5e5
    for _ in range(3):
        a.gn_iteration(lam)
        b.linearize()
        b.solve_linear(lam)
        b.update_states()
    a.synchronize()
    pa, sa, _ = a.get_window()
    pb, sb, _ = b.get_window()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(sa, sb)
    np.testing.assert_array_equal(a.get_landmarks(), b.get_landmarks())
    assert a.chi2() == b.chi2()


def test_exchange_hook_path_with_rccl_on_one_rank(vio, hip_lib):
    """The multi-GPU path end to end on the one GPU available here: a world_size-1 RCCL group, caller-owned torch
    exchange buffers bound into the library, the all-reduce issued from the library's hook on torch's current
    stream.  With one rank the sums are identities, so the result must equal the unsharded solve bit for bit."""
    import socket
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w = vio.synth.make_window(700, seed=23, ragged=True)
        sb = vio.sharded.ShardedBackend(hip_lib, w, 0, 1, dist=dist, torch_device="cuda", force_hook=True, exchange="hook",
                                        ctx_kwargs=dict(stream=torch.cuda.current_stream().cuda_stream))
        assert sb.exchange == "hook"
        rep = sb.solve(10)
        ps, ss, _ = sb.ctx.get_window()
        ls = sb.gather_landmarks()
        ref = hip_lib.context()
        ref.load(w)
        rr = ref.solve(10)
        pr, sr, _ = ref.get_window()
        assert rep.iterations == rr.iterations and rep.trials == rr.trials and rep.final_chi2 == rr.final_chi2
        np.testing.assert_array_equal(ps, pr)
        np.testing.assert_array_equal(ss, sr)
        np.testing.assert_array_equal(ls, ref.get_landmarks())
        for _ in range(5):
            sb.gn_iteration(5e5)
        sb.ctx.synchronize()
        assert np.isfinite(sb.ctx.chi2())
    finally:
        dist.destroy_process_group()


def test_native_rccl_exchange_on_one_rank(vio, hip_lib):
    """The default multi-GPU exchange: the library dlopens RCCL, builds its own communicator from a 128-byte id
    (vio_comm_unique_id / vio_comm_init) and all-reduces the exchange buffers on its own stream.  One rank here, so
    every sum is an identity and the LM solve must equal the unsharded one bit for bit; the sharded kernel sequence
    (k_step_sum + exchange + k_lm_decide) is the one that runs."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    w = vio.synth.make_window(700, seed=29, ragged=True)
    sb = vio.sharded.ShardedBackend(hip_lib, w, 0, 1, dist=None, torch_device="cuda", force_hook=True, exchange="native")
    assert sb.exchange == "native"
    assert sb.ctx.comm_info() == (1, 0)                 # ncclCommCount / ncclCommUserRank of the library's communicator
    assert hip_lib.context().comm_info() == (0, -1)     # no communicator: nothing to report
    rep = sb.solve(10)
    ps, ss, _ = sb.ctx.get_window()
    ref = hip_lib.context()
    ref.load(w)
    rr = ref.solve(10)
    pr, sr, _ = ref.get_window()
    assert rep.iterations == rr.iterations and rep.trials == rr.trials and rep.final_chi2 == rr.final_chi2
    np.testing.assert_array_equal(ps, pr)
    np.testing.assert_array_equal(ss, sr)
    np.testing.assert_array_equal(sb.ctx.get_landmarks(), ref.get_landmarks())
    # GN mode: the step scalars are not all-reduced per step but ride along with the next linearisation's all-reduce
    # (and are flushed when somebody looks): states and chi2 must still equal the unsharded run bit for bit
    sb.ctx.load(w)
    ref.load(w)
    for _ in range(5):
        sb.gn_iteration(5e5)
        ref.gn_iteration(5e5)
    ps, ss, _ = sb.ctx.get_window()
    pr, sr, _ = ref.get_window()
    np.testing.assert_array_equal(ps, pr)
    np.testing.assert_array_equal(ss, sr)
    np.testing.assert_array_equal(sb.ctx.get_landmarks(), ref.get_landmarks())
    assert sb.ctx.chi2() == ref.chi2()
    # MargOldFrame over the shards: same all-reduce, identical tail on every rank
    ms, mr = sb.marginalize(vio.MARG_OLD), ref.marginalize(vio.MARG_OLD)
    for k in ("H", "b", "err", "jt_inv"):
        np.testing.assert_array_equal(ms[k], mr[k])
    sb.ctx.comm_destroy()


@pytest.mark.parametrize("k_obs,ragged,policy", [(1, False, 1), (2, True, 1), (2, False, 0), (1, False, 0)])
def test_short_tracks_in_large_items(vio, oracle_lib, hip_lib, k_obs, ragged, policy):
    """Tracks of one or two observations in items of 100+ landmarks (the throughput policy; or a window large enough for the latency
    policy to choose them): the head of k_linearize stages (6 nb + 2) G doubles of Schur rows where the per-observation records of
    K G x 9 doubles will live — more than those hold for K <= 2.  Found by tools/fuzz_parity.py under VIO_DEFAULT_ITEM_POLICY=1:
    the LM loop rejected every first trial (Solve(10): 20 trials, chi2 7e5 instead of 565)."""
    n = 2500 if policy == 1 else 40000
    w = vio.synth.make_window(n, seed=1058 + k_obs, ragged=ragged, obs_per_landmark=k_obs)
    ch, co = hip_lib.context(loss_type=vio.LOSS_TRIVIAL, item_policy=policy), oracle_lib.context(loss_type=vio.LOSS_TRIVIAL)
    ch.load(w)
    co.load(w)
    rh, ro = ch.solve(6), co.solve(6)
    assert (rh.iterations, rh.trials) == (ro.iterations, ro.trials)
    assert abs(rh.final_chi2 - ro.final_chi2) <= 1e-7 * ro.final_chi2
    assert np.abs(ch.get_window()[0] - co.get_window()[0]).max() <= 1e-6
    ch.load(w)
    co.load(w)
    ch.linearize()
    _, lam = ch.init_lm()
    for _ in range(3):
        ch.gn_iteration(lam)
        co.gn_iteration(lam)
    assert np.abs(ch.get_window()[0] - co.get_window()[0]).max() <= 1e-9
    assert np.abs(ch.get_landmarks() - co.get_landmarks()).max() <= 1e-8


def test_native_exchange_falls_back_to_the_hook(vio, hip_lib, monkeypatch):
    """If the library cannot build its own RCCL communicator (librccl.so not resolvable, ncclCommInitRank refused), every rank
    takes the portable path instead — the same all-gather from the library's hook — and the results are the same bits."""
    import torch
    def refuse(self, *a, **k):
        raise vio.VioError(-2, "injected: no communicator")
    monkeypatch.setattr(type(hip_lib.context()), "comm_init", refuse)
    torch.cuda.set_device(0)
    w = vio.synth.make_window(500, seed=31, ragged=True)
    sb = vio.sharded.ShardedBackend(hip_lib, w, 0, 1, dist=None, torch_device="cuda", force_hook=True, exchange="native")
    assert sb.exchange == "hook"
    rep = sb.solve(10)
    ref = hip_lib.context()
    ref.load(w)
    rr = ref.solve(10)
    assert rep.iterations == rr.iterations and rep.trials == rr.trials and rep.final_chi2 == rr.final_chi2
    np.testing.assert_array_equal(sb.ctx.get_window()[0], ref.get_window()[0])
    np.testing.assert_array_equal(sb.ctx.get_landmarks(), ref.get_landmarks())


@pytest.mark.gpu
@pytest.mark.parametrize("k_obs,ext_fixed,n", [(10, 0, 600), (10, 1, 3000), (7, 0, 1500), (1, 1, 200)])
def test_long_tracks_and_free_extrinsic(vio, oracle_lib, hip_lib, k_obs, ext_fixed, n):
    """The widest patterns a window can hold: up to 10 observations per landmark (12 pattern blocks with a free
    extrinsic: 45 MFMA products per item in k_linearize, 15 Schur tiles), and the other end, one observation."""
    w = vio.synth.make_window(n, seed=100 + k_obs, obs_per_landmark=k_obs)
    ch, co = hip_lib.context(ext_fixed=ext_fixed), oracle_lib.context(ext_fixed=ext_fixed)
    ch.load(w); co.load(w)
    ch.linearize(); co.linearize()
    Hh, bh = ch.get_schur_system()
    Ho, bo = co.get_schur_system()
    scale = np.abs(Ho).max()
    assert np.abs(Hh - Ho).max() <= 1e-11 * scale
    assert np.abs(bh - bo).max() <= 1e-11 * max(np.abs(bo).max(), 1.0)
    chi_h, lam_h = ch.init_lm()
    chi_o, lam_o = co.init_lm()
    assert abs(chi_h - chi_o) <= 1e-11 * chi_o and abs(lam_h - lam_o) <= 1e-11 * lam_o
    ch.solve_linear(lam_o); co.solve_linear(lam_o)
    dh, _ = ch.get_delta()
    do, _ = co.get_delta()
    assert np.abs(dh - do).max() <= 1e-8 * max(np.abs(do).max(), 1e-12)
    rh, ro = ch.solve(10), co.solve(10)
    assert rh.final_chi2 <= 1.0001 * ro.final_chi2 + 1e-9


@pytest.mark.gpu
def test_gn_iterations_interleaved_with_everything_else(vio, oracle_lib, hip_lib):
    """In GN mode a step leaves its landmark update, its chi2 and its test to the next iteration; any other call in
    between must find them done (flush) and must not disturb the iterations that follow."""
    w = vio.synth.make_window(700, seed=31, ragged=True)
    lam = 5e5
    plain, mixed = hip_lib.context(), hip_lib.context()
    plain.load(w)
    mixed.load(w)
    for _ in range(6):
        plain.gn_iteration(lam)
    pp, sp, _ = plain.get_window()
    lp = plain.get_landmarks()
    chi_p = plain.chi2()
    # the same six iterations with reads, a chi2 evaluation and a stepwise linearisation thrown in
    mixed.gn_iteration(lam)
    p1, _, _ = mixed.get_window()                        # read-back flushes the owed work
    mixed.gn_iteration(lam)
    mixed.gn_iteration(lam)
    c3 = mixed.chi2()
    l3 = mixed.get_landmarks()
    mixed.gn_iteration(lam)
    mixed.linearize()                                    # a stepwise call: same state, nothing moves
    H, b = mixed.get_schur_system()
    assert np.isfinite(H).all() and np.isfinite(b).all()
    mixed.gn_iteration(lam)
    mixed.gn_iteration(lam)
    pm, sm, _ = mixed.get_window()
    np.testing.assert_array_equal(pm, pp)
    np.testing.assert_array_equal(sm, sp)
    np.testing.assert_array_equal(mixed.get_landmarks(), lp)
    assert mixed.chi2() == chi_p
    # against the oracle's plain loop
    o = oracle_lib.context()
    o.load(w)
    for _ in range(3):
        o.gn_iteration(lam)
    po, _, _ = o.get_window()
    assert abs(c3 - o.chi2()) <= 1e-9 * c3
    np.testing.assert_allclose(l3, o.get_landmarks(), rtol=1e-8, atol=1e-10)
    assert p1.shape == po.shape
    # an LM solve and a marginalisation straight after GN iterations start from the flushed state
    rep = mixed.solve(5)
    assert np.isfinite(rep.final_chi2) and rep.final_chi2 <= chi_p * (1 + 1e-9)
    prior = mixed.marginalize(vio.MARG_OLD)
    assert np.isfinite(prior["H"]).all()


def test_gn_iterations_with_a_prior(vio, oracle_lib, hip_lib):
    """With a marginalisation prior a GN step leaves err_prior to the next k_reduce (the pose solve writes b_prior' only)
    or to the flush: b_prior, err_prior and chi2 must equal the classic path bit for bit whenever anybody looks, and
    the oracle's within rounding."""
    w0 = vio.synth.make_window(300, seed=41, t0=0.9)
    c0 = oracle_lib.context()
    c0.load(w0)
    c0.solve(10)
    w = vio.synth.make_window(700, seed=42, ragged=True)
    w.prior = c0.marginalize(vio.MARG_OLD)
    lam = 5e5
    plain, mixed, o = hip_lib.context(), hip_lib.context(), oracle_lib.context()
    for c in (plain, mixed, o):
        c.load(w)
    for _ in range(4):
        plain.gn_iteration(lam)
        o.gn_iteration(lam)
    bp, ep = plain.get_prior()
    chi_p = plain.chi2()
    mixed.gn_iteration(lam)
    b1, e1 = mixed.get_prior()                  # flush: err_prior of the pending step comes from k_errprior
    mixed.gn_iteration(lam)
    mixed.gn_iteration(lam)
    c3 = mixed.chi2()
    mixed.gn_iteration(lam)
    bm, em = mixed.get_prior()
    np.testing.assert_array_equal(bm, bp)
    np.testing.assert_array_equal(em, ep)
    assert mixed.chi2() == chi_p and np.isfinite(c3)
    assert np.abs(e1).max() > 0 and not np.array_equal(e1, em)
    bo, eo = o.get_prior()
    assert tu.rel_max(bp, bo) <= 1e-9 and tu.rel_max(ep, eo) <= 1e-7
    assert abs(chi_p - o.chi2()) <= 1e-9 * chi_p
    # the LM path forms err_prior inside the pose solve: same numbers as the deferred path on the same step
    a, b = hip_lib.context(), hip_lib.context()
    a.load(w)
    b.load(w)
    a.gn_iteration(lam)
    b.linearize()
    b.solve_linear(lam)
    b.update_states()
    np.testing.assert_array_equal(a.get_prior()[1], b.get_prior()[1])


@pytest.mark.parametrize("n,ragged", [(25000, False), (60000, True), (3000, True)])
def test_item_sizing_regimes(vio, oracle_lib, hip_lib, n, ragged):
    """The landmarks-per-item choice follows the number of k_linearize rounds the window needs on the device's CUs: one
    round with items as small as 8 landmarks (3 000), exactly one (the headline), two and more (25 000 = the per-GPU share
    of BASELINE.json's 200k configuration; 60 000 ragged = every pattern, several rounds).  Same arithmetic everywhere."""
    w = vio.synth.make_window(n, seed=77, ragged=ragged)
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(w)
    co.load(w)
    compare_stepwise(tu.run_stepwise(ch), tu.run_stepwise(co))


def test_bench_workload_with_prior_against_oracle(vio, oracle_lib, hip_lib, big_window):
    """The workload bench.py times: the 20 000-landmark window with the marginalisation prior of a preceding window.
    One stepwise LM step and five GN iterations against the oracle; Solve(10) against the oracle's end state."""
    w0 = vio.synth.make_window(300, seed=41, t0=0.9)
    c0 = oracle_lib.context()
    c0.load(w0)
    c0.solve(10)
    w = big_window.copy()
    w.prior = c0.marginalize(vio.MARG_OLD)
    ch, co = hip_lib.context(), oracle_lib.context()
    ch.load(w)
    co.load(w)
    compare_stepwise(tu.run_stepwise(ch), tu.run_stepwise(co))
    ch.load(w)
    co.load(w)
    ch.linearize()
    _, lam = ch.init_lm()
    for _ in range(5):
        ch.gn_iteration(lam)
        co.gn_iteration(lam)
    ph, sh, _ = ch.get_window()
    po, so_, _ = co.get_window()
    assert np.abs(ph - po).max() <= 1e-8 and np.abs(sh - so_).max() <= 1e-7
    assert np.abs(ch.get_landmarks() - co.get_landmarks()).max() <= 1e-8
    assert abs(ch.chi2() - co.chi2()) <= 1e-9 * co.chi2()
    bh, eh = ch.get_prior()
    bo, eo = co.get_prior()
    assert tu.rel_max(bh, bo) <= 1e-9 and tu.rel_max(eh, eo) <= 1e-7
    ch.load(w)
    co.load(w)
    rh, ro = ch.solve(10), co.solve(10)
    assert rh.iterations == ro.iterations and abs(rh.final_chi2 - ro.final_chi2) <= 1e-6 * ro.final_chi2


def test_landmark_without_information_gives_an_error_not_a_crash(vio, oracle_lib, hip_lib):
    """A landmark whose every edge the loss weights to zero (single-observation tracks, Tukey, outliers) has h_ll = 0: the
    reference's Hmm_inv is infinite and its dense products fill H_pp_schur with NaN (problem.cc:419-429).  Same here and
    in the oracle: non-finite step, VIO_ERR_NOT_FINITE from vio_solve — and the NaN diagonal must still rank to a valid
    pivot permutation (it used to index out of bounds)."""
    kw = dict(pos_noise=0.001, rot_noise=0.0002, depth_noise=0.003, pixel_noise=0.25 / 460, outlier_fraction=0.05)
    w = vio.synth.make_window(200, seed=1019, ragged=False, obs_per_landmark=1, **kw)
    ch, co = hip_lib.context(ext_fixed=0, loss_type=3), oracle_lib.context(ext_fixed=0, loss_type=3)
    ch.load(w)
    co.load(w)
    a, b = tu.run_stepwise(ch), tu.run_stepwise(co)
    assert (b["hll"] == 0).any() and np.array_equal(a["hll"] == 0, b["hll"] == 0)
    assert not np.isfinite(a["dx_pose"]).all() and not np.isfinite(b["dx_pose"]).all()
    ch.load(w)
    ch.linearize()
    for _ in range(3):
        ch.gn_iteration(5e5)
    assert not np.isfinite(ch.chi2())
    ch.load(w)
    with pytest.raises(vio.VioError) as e:
        ch.solve(10)
    assert "NOT_FINITE" in str(e.value)
    ch.load(vio.synth.make_window(300, seed=8, **kw))       # the context is still usable (tracks of 4: every landmark has information)
    assert np.isfinite(ch.solve(10).final_chi2)


def test_marginalisation_in_two_halves(vio, oracle_lib, hip_lib):
    """vio_marginalize_begin / vio_marginalize_end: the dense tail on the library's helper thread while the caller uploads the next
    window; the prior is the one vio_marginalize returns, bit for bit; a second begin or a destroy waits for a tail nobody collected."""
    w = vio.synth.make_window(600, seed=61, ragged=True)
    w2 = vio.synth.make_window(500, seed=62, t0=1.1)
    a, b = hip_lib.context(), hip_lib.context()
    for c in (a, b):
        c.load(w)
        c.solve(10)
    m_sync = a.marginalize(vio.MARG_OLD)
    b.marginalize_begin(vio.MARG_OLD)
    b.set_window(w2.poses, w2.speed_bias, w2.ext)                  # the next frame's uploads, under the tail
    b.set_landmarks(w2.inv_depth)
    b.set_observations(w2.lm, w2.host, w2.target, w2.pts_i, w2.pts_j)
    for k, pre in enumerate(w2.preint):
        b.set_imu(k, pre)
    m_bg = b.marginalize_end()
    for k in tu.PRIOR_FIELDS:
        np.testing.assert_array_equal(m_bg[k], m_sync[k])
    w2.prior = m_bg
    b.set_prior(m_bg)
    a.load(w2)
    ra, rb = a.solve(10), b.solve(10)
    assert (ra.iterations, ra.final_chi2) == (rb.iterations, rb.final_chi2)
    with pytest.raises(vio.VioError):
        a.marginalize_end()                                        # nothing begun
    # MargNewFrame the same way; a begin that nobody collects, then another one, then the context goes
    a.marginalize_begin(vio.MARG_SECOND_NEW)
    a.marginalize_begin(vio.MARG_SECOND_NEW)
    n1 = a.marginalize_end()
    n2 = b.marginalize(vio.MARG_SECOND_NEW)
    for k in tu.PRIOR_FIELDS:
        np.testing.assert_array_equal(n1[k], n2[k])
    a.marginalize_begin(vio.MARG_OLD)
    a.close()
    # the CPU library exports the same two calls
    c = oracle_lib.context()
    c.load(w)
    c.marginalize_begin(vio.MARG_OLD)
    mo = c.marginalize_end()
    c.load(w)
    for k in tu.PRIOR_FIELDS:
        np.testing.assert_array_equal(mo[k], c.marginalize(vio.MARG_OLD)[k])
