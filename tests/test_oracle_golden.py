"""The CPU oracle against the golden vectors dumped from the COMPILED REFERENCE (tests/golden/make_golden.py).
This is what pins the oracle; the GPU tests then compare the HIP path with the oracle.

Tolerances (all fp64):
  per-function values      relative 1e-12 (same formulas, different evaluation order)
  H_pp_schur_              1e-9 in the metric |dH_ij| / sqrt(H_ii H_jj) (the matrix spans 1e5 .. 1e16)
  delta_x at the reference's own lambda: 1e-9 absolute (expected <= 1e-9, required <= 1e-6)
  Solve(10) end state      1e-6 absolute: ten LM steps walk lambda down to ~10 where cond(H) ~ 1e15
  marginalisation          compared through invariants (the entries are ill-posed)
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import vio_testutil as tu
from conftest import GOLDEN_DIR

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
WINDOW_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "window_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def cfg_of(z):
    kw = {}
    if "cfg_ext_fixed" in z:
        kw["ext_fixed"] = int(z["cfg_ext_fixed"])
    if "cfg_loss_type" in z:
        kw["loss_type"] = int(z["cfg_loss_type"])
    if "cfg_loss_delta" in z:
        kw["loss_delta"] = float(z["cfg_loss_delta"])
    return kw


def test_golden_files_present():
    assert len(WINDOW_FILES) >= 17 and sum("window_xyz_" in f for f in WINDOW_FILES) >= 6 and sum("window_noimu_" in f for f in WINDOW_FILES) >= 3
    for f in ("reproj_edges", "loss_and_robust", "pose_plus", "ldlt", "symmetric_eigen", "inverse15", "reproj_xyz_edges", "inverse3"):
        assert os.path.exists(os.path.join(GOLDEN_DIR, f + ".npz"))


def test_reprojection_edge(oracle_lib):
    z = load("reproj_edges")
    f = oracle_lib.dll.vioo_reproj_edge
    f.restype = None
    for e in range(z["residual"].shape[0]):
        r, Jl, Ji, Jj, Je = np.zeros(2), np.zeros(2), np.zeros(12), np.zeros(12), np.zeros(12)
        a = [np.ascontiguousarray(z[k][e]) for k in ("pose_i", "pose_j", "ext", "pts_i", "pts_j")]
        f(dp(a[0]), dp(a[1]), dp(a[2]), C.c_double(z["inv_depth"][e]), dp(a[3]), dp(a[4]), dp(r), dp(Jl), dp(Ji), dp(Jj), dp(Je))
        for got, key in ((r, "residual"), (Jl, "J_lambda"), (Ji, "J_pose_i"), (Jj, "J_pose_j"), (Je, "J_ext")):
            want = z[key][e]
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (e, key)


def test_reprojection_xyz_edge(oracle_lib):
    """EdgeReprojectionXYZ::ComputeResidual / ComputeJacobians (edge_reprojection.cc:130-180)."""
    z = load("reproj_xyz_edges")
    f = oracle_lib.dll.vioo_reproj_xyz_edge
    f.restype = None
    ext = np.ascontiguousarray(z["ext"])
    for e in range(z["residual"].shape[0]):
        r, Jf, Jp = np.zeros(2), np.zeros(6), np.zeros(12)
        f(dp(np.ascontiguousarray(z["pose"][e])), dp(ext), dp(np.ascontiguousarray(z["pw"][e])), dp(np.ascontiguousarray(z["obs"][e])),
          dp(r), dp(Jf), dp(Jp))
        for got, key in ((r, "residual"), (Jf, "J_feature"), (Jp, "J_pose")):
            want = z[key][e]
            assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), (e, key)


def test_inverse3_matches_eigens_dynamic_inverse(oracle_lib):
    """Hmm.block(idx, idx, 3, 3).inverse() (problem.cc:424): PartialPivLU, every pivot choice in the file."""
    z = load("inverse3")
    f = oracle_lib.dll.vioo_inverse3
    f.restype = None
    for k in range(z["A"].shape[0]):
        out = np.zeros((3, 3))
        f(dp(np.ascontiguousarray(z["A"][k])), dp(out))
        np.testing.assert_allclose(out, z["Ainv"][k], rtol=1e-13, atol=1e-13 * np.abs(z["Ainv"][k]).max())


def test_loss_functions_and_robust_info(oracle_lib):
    z = load("loss_and_robust")
    fl = oracle_lib.dll.vioo_loss
    fl.restype = None
    for t, nm in ((1, "huber"), (2, "cauchy"), (3, "tukey")):
        for delta in (1.0, 2.5):
            want = z["%s_%g" % (nm, delta)]
            for i, e2 in enumerate(z["e2"]):
                out = np.zeros(3)
                fl(C.c_int(t), C.c_double(delta), C.c_double(e2), dp(out))
                np.testing.assert_allclose(out, want[i], rtol=1e-14, atol=1e-300)
    fr = oracle_lib.dll.vioo_robust_info2
    fr.restype = None
    for t, nm in ((0, "trivial"), (1, "huber"), (2, "cauchy"), (3, "tukey")):
        for i, r in enumerate(z["residuals"]):
            W, d = np.zeros(4), C.c_double()
            fr(C.c_int(t), C.c_double(1.0), C.c_double(460 / 1.5), dp(np.ascontiguousarray(r)), C.byref(d), dp(W))
            np.testing.assert_allclose(W, z["W_" + nm][i], rtol=1e-13, atol=1e-9)
            assert abs(d.value - z["drho_" + nm][i]) <= 1e-14


def test_pose_plus(oracle_lib):
    z = load("pose_plus")
    f = oracle_lib.dll.vioo_pose_plus
    f.restype = None
    for i in range(z["poses"].shape[0]):
        p = z["poses"][i].copy()
        f(dp(p), dp(np.ascontiguousarray(z["deltas"][i])))
        np.testing.assert_allclose(p, z["result"][i], rtol=0, atol=1e-15)


def test_ldlt_solve_matches_eigen(oracle_lib):
    z = load("ldlt")
    f = oracle_lib.dll.vioo_ldlt_solve
    f.restype = None
    x, tr = np.zeros(24), np.zeros(24, dtype=np.int32)
    f(C.c_int(24), dp(np.ascontiguousarray(z["A_small"])), dp(np.ascontiguousarray(z["b_small"])), dp(x),
      tr.ctypes.data_as(C.POINTER(C.c_int)))
    np.testing.assert_array_equal(tr, z["tr_small"])       # same transpositions as Eigen, zero pivot included
    np.testing.assert_allclose(x, z["x_small"], rtol=1e-11, atol=1e-13)
    for i in range(3):
        lam = float(z["lambda_%d" % i])
        A = np.ascontiguousarray(z["Hs"] + lam * np.eye(171))
        x, tr = np.zeros(171), np.zeros(171, dtype=np.int32)
        f(C.c_int(171), dp(A), dp(np.ascontiguousarray(z["bs"])), dp(x), tr.ctypes.data_as(C.POINTER(C.c_int)))
        np.testing.assert_array_equal(tr, z["tr_%d" % i])
        # the allowed error grows with cond(H + lambda I): 3e10 / 1.6e13 / 2.7e16
        tol = (1e-11, 1e-8, 1e-4)[i]
        assert np.abs(x - z["x_%d" % i]).max() <= tol, (i, np.abs(x - z["x_%d" % i]).max())
        # ... and Eigen's vectors themselves are only that close to the EXACT solution of the system (ldlt_exact.npz: 50 digits):
        # 4.1e-12 / 1.2e-7 / 1.4e-4.  Any backward-stable solver lands in that ball; which point of it is the solver's rounding.
        xe = load("ldlt_exact")["x_exact_%d" % i]
        e_eigen = np.abs(z["x_%d" % i] - xe).max()
        assert (3e-12, 8e-8, 8e-5)[i] <= e_eigen <= (6e-12, 2e-7, 2e-4)[i], (i, e_eigen)
        assert np.abs(x - xe).max() <= 1.5 * e_eigen


def test_symmetric_eigen(oracle_lib):
    z = load("symmetric_eigen")
    f = oracle_lib.dll.vioo_symmetric_eigen
    for key, n in (("small", 24), ("prior", 156)):
        A = np.ascontiguousarray(z["A_" + key])
        ev, V = np.zeros(n), np.zeros((n, n))
        assert f(C.c_int(n), dp(A), dp(ev), dp(V)) == 0
        scale = np.abs(z["evals_" + key]).max()
        assert np.abs(ev - z["evals_" + key]).max() <= 1e-12 * scale
        As = np.tril(A) + np.tril(A, -1).T
        assert np.abs(V @ np.diag(ev) @ V.T - As).max() <= 1e-11 * scale
        assert np.abs(V.T @ V - np.eye(n)).max() <= 1e-12


def test_inverse15(oracle_lib):
    z = load("inverse15")
    f = oracle_lib.dll.vioo_inverse15
    f.restype = None
    for k in range(z["cov"].shape[0]):
        out = np.zeros((15, 15))
        f(dp(np.ascontiguousarray(z["cov"][k])), dp(out))
        # element-wise, scaled by the diagonal: information spans 1e4 .. 4e15
        assert tu.scaled_sym_err(out, z["info"][k]) <= 1e-9


@pytest.mark.parametrize("path", WINDOW_FILES, ids=[os.path.basename(p)[:-4] for p in WINDOW_FILES])
def test_window_against_reference(vio, oracle_lib, path):
    check_window_against_golden(vio, oracle_lib, path)


def check_window_against_golden(vio, lib, path, dx_tol=1e-9, state_tol=1e-6, lambda_rtol=2e-5, extra_cfg=None):
    """Shared with the GPU tests: `lib` is the oracle here and the HIP library there."""
    z = dict(np.load(path))
    w = tu.arrays_to_window(vio, z)
    kw = cfg_of(z)
    kw.update(extra_cfg or {})
    ctx = lib.context(**kw)
    ctx.load(w)
    if "step_dx_pose" in z and "step_bs" not in z:          # the N = 2000 file: delta_x only
        ctx.linearize()
        chi0, lam0 = ctx.init_lm()
        assert abs(chi0 - float(z["step_chi0"])) <= 1e-10 * abs(float(z["step_chi0"]))
        assert lam0 == float(z["step_lambda0"])
        ctx.solve_linear(lam0)
        dxp, dxl = ctx.get_delta()
        assert np.abs(dxp - z["step_dx_pose"]).max() <= dx_tol
        assert np.abs(dxl - z["step_dx_lm"]).max() <= dx_tol
        return
    got = tu.run_stepwise(ctx)
    g = lambda k: z["step_" + k]
    if kw.get("loss_type") == vio.LOSS_HUBER and kw.get("loss_delta", 1.0) == 1.0:
        # Knife edge in the reference itself: for EVERY Huber outlier, however far beyond delta, rho' + 2 rho'' e2 is 0 in
        # exact arithmetic (loss_function.cc:16-20), so the `> 0` test of Edge::RobustInfo (edge.cc:62) is decided by the
        # last bit of e2 and the edge's weight along the residual is either rho' or 0.  Any two evaluation orders of the
        # residual disagree on some edges, so only W-independent quantities are comparable at window level; the
        # branch itself is pinned with identical inputs in test_loss_functions_and_robust_info, and the whole Huber path
        # on the same window with every edge an inlier in window_n200_s46_huber_d50.npz (make_golden_huber.py).
        assert abs(got["chi0"] - g("chi0")) <= 1e-11 * abs(g("chi0"))
        assert got["lambda0"] == g("lambda0")
        return
    if "step_Hs" in z:
        assert tu.scaled_sym_err(got["Hs"], g("Hs")) <= 1e-9
    scale_b = np.abs(g("bs")).max()
    assert np.abs(got["bs"] - g("bs")).max() <= 1e-11 * scale_b
    assert np.abs(got["bpp"] - g("bpp")).max() <= 1e-11 * np.abs(g("bpp")).max()
    assert tu.rel_max(got["diag"], g("diag")) <= 1e-12
    assert tu.rel_max(got["hll"], g("hll")) <= 1e-11 and tu.rel_max(got["bl"], g("bl")) <= 1e-10
    assert abs(got["chi0"] - g("chi0")) <= 1e-11 * abs(g("chi0"))
    # with IMU factors lambda_0 is the cap 1e-5 * 5e10 (problem.cc:511-520): exact; without them it is 1e-5 * the largest diagonal
    # entry, a sum whose last bits depend on the order of the terms
    assert got["lambda0"] == g("lambda0") or ("in_pre_valid" in z and abs(got["lambda0"] - g("lambda0")) <= 1e-12 * g("lambda0"))
    assert np.abs(got["dx_pose"] - g("dx_pose")).max() <= dx_tol
    assert np.abs(got["dx_lm"] - g("dx_lm")).max() <= dx_tol
    for k in ("poses1", "sb1", "ext1", "invd1"):
        assert np.abs(got[k] - g(k)).max() <= dx_tol, k
    if "in_prior_H" in z:
        assert tu.rel_max(got["bprior1"], g("bprior1")) <= 1e-9
        assert tu.rel_max(got["errprior1"], g("errprior1")) <= 1e-8
    assert abs(got["chi1"] - g("chi1")) <= 1e-8 * abs(g("chi1"))
    assert int(got["accepted"]) == int(g("accepted"))
    assert abs(got["lambda1"] - g("lambda1")) <= 1e-9 * abs(g("lambda1"))
    if "solve_posesF" in z:
        ctx2 = lib.context(**kw)
        ctx2.load(w)
        sol, rep = tu.run_solve(ctx2, 10)
        assert int(sol["iterations"]) == int(z["solve_iterations"])
        assert abs(sol["final_chi2"] - z["solve_final_chi2"]) <= 1e-6 * abs(z["solve_final_chi2"])
        # the reference prints chi/lambda with 6 significant digits: that is what its trace pins
        n = len(z["solve_chi2_trace"])
        np.testing.assert_allclose(sol["chi2_trace"][:n], z["solve_chi2_trace"], rtol=2e-5)
        # lambda_{k+1}/lambda_k = 1-(2 rho-1)^3 with rho = (chi-chi_new)/scale: close to convergence chi-chi_new is a
        # difference of nearly equal numbers, so the schedule amplifies rounding (a 1e-9 state difference shows up
        # in the 4th digit of lambda after six steps)
        np.testing.assert_allclose(sol["lambda_trace"][:n], z["solve_lambda_trace"], rtol=lambda_rtol)
        for k in ("posesF", "sbF", "extF", "invdF"):
            assert np.abs(sol[k] - z["solve_" + k]).max() <= state_tol, k
    for kind in (0, 1):
        if "marg%d_H" % kind not in z:
            continue
        wm = tu.arrays_to_window(vio, z, prefix="marg%d_in_" % kind)
        ctx3 = lib.context(**kw)
        ctx3.load(wm)
        m = ctx3.marginalize(kind)
        check_prior(m, {k: z["marg%d_%s" % (kind, k)] for k in tu.PRIOR_FIELDS})


def solve_trace_stepwise(lib, w, kw, iterations=10, order=None):
    """Problem::Solve's loop (problem.cc:188-245) through the single-step entry points; per outer iteration
    (state vector, chi2, lambda, trials).  Same loop as tools/parity_trace.py, which made tests/golden/solve_trace.npz."""
    c = lib.context(**kw)
    if order is not None:
        c.set_solve_order(order)        # (HIP library only: the elimination order of the pose solve)
    c.load(w)
    c.linearize()
    chi, lam = c.init_lm()
    out, last = [], 1e20
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
        out.append((np.concatenate([p.ravel(), s.ravel(), e.ravel(), lm]), chi, lam, trials))
        if last - chi < 1e-5:
            break
        last = chi
    return out


def test_every_step_from_the_reference_state(vio, oracle_lib):
    """the per-step figure of SURVEY.md section 7 for the oracle: one trial from the reference's own state k at the reference's
    lambda against the reference's state k + 1 (harness: tests/test_gpu_parity.py, which runs it on the HIP library)"""
    from test_gpu_parity import per_step_differences
    d = per_step_differences(vio, oracle_lib)
    assert len(d) >= 8 and max(max(v) for v in d.values()) <= 1e-6, {k: max(v) for k, v in d.items()}


def test_solve_trace_against_the_reference_iteration_by_iteration(vio, oracle_lib):
    """The oracle stays within 1e-13 of the reference while lambda is large; the difference appears where lambda has walked
    down to O(10..100) (cond(H + lambda I) ~ 1e15) and stays below 1e-6 (measured 1.9e-7, profiles/parity_trace.json)."""
    zr = np.load(os.path.join(GOLDEN_DIR, "solve_trace.npz"))
    n_checked = 0
    for path in WINDOW_FILES:
        z = dict(np.load(path))
        name = os.path.basename(path)[:-4]
        if name + "_state" not in zr:
            continue
        tr = solve_trace_stepwise(oracle_lib, tu.arrays_to_window(vio, z), cfg_of(z))
        rs, rl = zr[name + "_state"], zr[name + "_lam"]
        assert len(tr) == len(rs) and [t[3] for t in tr] == list(zr[name + "_trials"]), name
        d = [float(np.abs(t[0] - r).max()) for t, r in zip(tr, rs)]
        assert max(d[:4]) <= 1e-11 and max(d) <= 1e-6, (name, d)
        assert max(abs(t[2] - l) / l for t, l in zip(tr, rl)) <= 1e-6, name
        n_checked += 1
    assert n_checked >= 7


NOIMU_FILES = [p for p in WINDOW_FILES if "window_noimu_" in p]


def check_noimu_trace(vio, lib, path, tol, order=None):
    """tests/golden/window_noimu_*.npz (make_golden_noimu.py): windows without a single IMU edge (estimator.cpp:956-970 skips the
    edge of an interval with sum_dt > 10) — fixtures in which every line that ran was the reference's (no EdgeImuPort in the
    graph).  The reference's state, lambda and trial count after every outer iteration of Solve(10), lambda_0 = O(10..100) walking
    down to O(1): the low-lambda regime no IMU information damps.  Returns the per-iteration differences."""
    z = dict(np.load(path))
    w, kw = tu.arrays_to_window(vio, z), cfg_of(z)
    assert all(p is None for p in w.preint)
    tr = solve_trace_stepwise(lib, w, kw, order=order)
    rs, rl = z["trace_state"], z["trace_lam"]
    assert len(tr) == len(rs) and [t[3] for t in tr] == list(z["trace_trials"]), path
    d = [float(np.abs(t[0] - r).max()) for t, r in zip(tr, rs)]
    assert max(d) <= tol, (path, d)
    ok = [i for i, t in enumerate(z["trace_trials"]) if t <= 10 and rl[i] < 1e10]       # (ten rejections leave lambda at 2^55 lambda)
    assert max(abs(tr[i][2] - rl[i]) / rl[i] for i in ok) <= 1e-5, path
    return d


@pytest.mark.parametrize("path", NOIMU_FILES, ids=[os.path.basename(p)[:-4] for p in NOIMU_FILES])
def test_imu_less_windows_iteration_by_iteration(vio, oracle_lib, path):
    """measured: 1e-14 .. 1.4e-10 over the three windows (the one with a prior the largest)"""
    check_noimu_trace(vio, oracle_lib, path, 1e-9)


def check_prior(m, ref):
    """Marginalisation outputs are compared through invariants: the Schur complement subtracts O(1e16) terms
    whose difference is O(1e5), the eps = 1e-8 eigenvalue cut and the 1e-9 zeroing are discontinuous, and
    eigenvector signs are arbitrary ('Marginalisation parity is ill-posed entry-wise')."""
    Hs = np.abs(ref["H"]).max()
    assert np.abs(m["H"] - ref["H"]).max() <= 2e-5 * Hs
    assert np.abs(m["b"] - ref["b"]).max() <= 1e-6 * max(np.abs(ref["b"]).max(), 1.0)
    ev, evr = np.linalg.eigvalsh(m["H"]), np.linalg.eigvalsh(ref["H"])
    # spectrum: absolute agreement relative to the largest eigenvalue.  (The count of eigenvalues above the 1e-8
    # cut is NOT an invariant: with the extrinsic free the smallest ones are O(1e-3..1) = 1e-7 of the largest,
    # i.e. inside the rounding of the 1e16-sized terms the Schur complement cancels.)
    assert np.abs(ev - evr).max() <= 2e-5 * evr.max()
    # err_prior = -Jt_inv b (problem.cc:774).  Its components along eigenvectors whose eigenvalue lies between the
    # 1e-8 cut and the rounding noise of H (~1e-6 of the largest) are amplified noise in the reference itself, so
    # across implementations only the well-conditioned part is comparable; the full vector is checked for
    # consistency with the implementation's own Jt_inv and b.
    assert np.abs(m["err"] + m["jt_inv"] @ m["b"]).max() <= 1e-9 * max(np.abs(m["err"]).max(), 1e-12)

    def damped_energy(H, b):
        # b^T (H + mu I)^-1 b, mu = 1e-4 of the largest eigenvalue: what ||err_prior||^2 would be with a smooth cut
        Hs_ = 0.5 * (H + H.T)
        mu = 1e-4 * np.linalg.eigvalsh(Hs_).max()
        return float(b @ np.linalg.solve(Hs_ + mu * np.eye(H.shape[0]), b))
    qa, qb = damped_energy(m["H"], m["b"]), damped_energy(ref["H"], ref["b"])
    # H itself moves by 1e-5 lambda_max = 0.1 mu between any two evaluation orders of the same arithmetic (two builds of
    # the HIP library: 5e-5 .. 1.6e-3 on the golden windows, profiles/r02d_marg_invariant_scatter.txt; the oracle: 1.6e-4)
    assert abs(qa - qb) <= 5e-3 * max(qb, 1e-9)
    # Jt_inv^T Jt_inv is the pseudo-inverse of H_prior restricted to the kept eigenspace: H P H == H
    # (evaluated in double the product H P H carries eps |H|^2 |P|; |P| = 1 / (smallest kept eigenvalue), and the cut keeps anything
    # above 1e-8: the reference's own prior of window_noimu_n300_s47 keeps a gauge direction's noise eigenvalue of 1.17e-8 — |P| = 2e7 —
    # and misses H P H = H by 4.5e4 in this arithmetic, the oracle's cuts it and passes at 1e-5 Hs)
    P = m["jt_inv"].T @ m["jt_inv"]
    Hm = np.abs(m["H"]).max()
    assert np.abs(m["H"] @ P @ m["H"] - m["H"]).max() <= 1e-5 * Hs + 64 * np.finfo(float).eps * Hm * Hm * np.abs(P).max()


def test_openmp_build_of_the_oracle_agrees_with_the_serial_one(vio, oracle_lib):
    """oracle/liboracle_omp.so is bench.py's all-cores CPU figure (never the parity checker): same terms, summed
    landmark-major per thread; it must agree with the pinned serial build to rounding on every path it touches."""
    import subprocess
    from conftest import ORACLE_DIR
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "omp"])
    omp = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle_omp.so"), "vioo_")
    for n, ragged, ext_fixed in ((300, True, 0), (2000, False, 1)):
        w = vio.synth.make_window(n, seed=12, ragged=ragged)
        a, b = oracle_lib.context(ext_fixed=ext_fixed), omp.context(ext_fixed=ext_fixed)
        a.load(w)
        b.load(w)
        sa, sb_ = tu.run_stepwise(a), tu.run_stepwise(b)
        assert tu.scaled_sym_err(sb_["Hs"], sa["Hs"]) <= 1e-11
        assert np.abs(sa["dx_pose"] - sb_["dx_pose"]).max() <= 1e-10 and np.abs(sa["dx_lm"] - sb_["dx_lm"]).max() <= 1e-10
        a.load(w)
        b.load(w)
        ra, rb = a.solve(10), b.solve(10)
        assert ra.iterations == rb.iterations and abs(ra.final_chi2 - rb.final_chi2) <= 1e-8 * ra.final_chi2
        ma, mb = a.marginalize(vio.MARG_OLD), b.marginalize(vio.MARG_OLD)
        assert np.abs(ma["H"] - mb["H"]).max() <= 2e-5 * np.abs(ma["H"]).max()
