"""Live comparison of the oracle with the compiled reference on fresh seeds and on the edge cases the golden
files do not hold.  Runs only where oracle/_ref can be built (the container that mounts /root/reference)."""
import numpy as np
import pytest

import vio_testutil as tu

pytestmark = pytest.mark.ref


@pytest.mark.parametrize("n,seed,ragged,ext_fixed", [(1, 1, False, 1), (7, 2, True, 1), (40, 3, True, 0), (500, 4, False, 1)])
def test_stepwise_and_solve(vio, oracle_lib, ref_lib, n, seed, ragged, ext_fixed):
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    co, cr = oracle_lib.context(ext_fixed=ext_fixed), ref_lib.context(ext_fixed=ext_fixed)
    co.load(w)
    cr.load(w)
    a, b = tu.run_stepwise(co), tu.run_stepwise(cr)
    assert tu.scaled_sym_err(a["Hs"], b["Hs"]) <= 1e-9
    assert np.abs(a["dx_pose"] - b["dx_pose"]).max() <= 1e-9 and np.abs(a["dx_lm"] - b["dx_lm"]).max() <= 1e-9
    assert abs(a["chi1"] - b["chi1"]) <= 1e-8 * abs(b["chi1"])
    assert int(a["accepted"]) == int(b["accepted"])
    co2, cr2 = oracle_lib.context(ext_fixed=ext_fixed), ref_lib.context(ext_fixed=ext_fixed)
    co2.load(w)
    cr2.load(w)
    sa, _ = tu.run_solve(co2)
    sb, _ = tu.run_solve(cr2)
    assert int(sa["iterations"]) == int(sb["iterations"])
    assert np.abs(sa["posesF"] - sb["posesF"]).max() <= 1e-6
    assert np.abs(sa["invdF"] - sb["invdF"]).max() <= 1e-6


def test_imu_only_window(vio, oracle_lib, ref_lib):
    """No landmarks at all: the IMU chain alone (problem.cc handles ordering_landmarks_ == 0)."""
    w = vio.synth.make_window(0, seed=5)
    co, cr = oracle_lib.context(), ref_lib.context()
    co.load(w)
    cr.load(w)
    co.linearize()
    cr.linearize()
    ca, la = co.init_lm()
    cb, lb = cr.init_lm()
    assert abs(ca - cb) <= 1e-9 * max(abs(cb), 1e-12) and la == lb
    co.solve_linear(la)
    cr.solve_linear(lb)
    assert np.abs(co.get_delta()[0] - cr.get_delta()[0]).max() <= 1e-9


def test_missing_imu_edge(vio, oracle_lib, ref_lib):
    """estimator.cpp:959-960 skips an IMU edge whose sum_dt exceeds 10 s."""
    w = vio.synth.make_window(60, seed=6)
    w.preint[4] = None
    co, cr = oracle_lib.context(), ref_lib.context()
    co.load(w)
    cr.load(w)
    a, b = tu.run_stepwise(co), tu.run_stepwise(cr)
    assert np.abs(a["dx_pose"] - b["dx_pose"]).max() <= 1e-8
    assert abs(a["chi0"] - b["chi0"]) <= 1e-10 * abs(b["chi0"])


def test_empty_graph_is_rejected(vio, oracle_lib):
    """Problem::Solve returns false without edges (problem.cc:172-175)."""
    w = vio.synth.make_window(0, seed=5)
    w.preint = [None] * 10
    ctx = oracle_lib.context()
    ctx.load(w)
    with pytest.raises(vio.VioError) as e:
        ctx.solve(10)
    assert e.value.status == -4


def test_landmark_without_information(vio, oracle_lib, ref_lib):
    """Single-observation tracks under Tukey: some landmarks keep no weighted edge, h_ll = 0, and the reference's dense
    Hpm * Hmm_inv multiplies zeros by infinity (problem.cc:419-429): its step is NaN.  The oracle's sparse loop must say
    the same, and its Solve must end the reference's way — NaN trials rejected (problem.cc:559), last accepted states kept —
    with the ABI's VIO_ERR_NOT_FINITE as status."""
    kw = dict(pos_noise=0.001, rot_noise=0.0002, depth_noise=0.003, pixel_noise=0.25 / 460, outlier_fraction=0.05)
    w = vio.synth.make_window(200, seed=1019, ragged=False, obs_per_landmark=1, **kw)
    co, cr = oracle_lib.context(ext_fixed=0, loss_type=3), ref_lib.context(ext_fixed=0, loss_type=3)
    co.load(w)
    cr.load(w)
    a, b = tu.run_stepwise(co), tu.run_stepwise(cr)
    assert (a["hll"] == 0).any()
    assert not np.isfinite(a["dx_pose"]).all() and not np.isfinite(b["dx_pose"]).all()
    co.load(w)
    p0, _, _ = co.get_window()
    with pytest.raises(vio.VioError) as e:
        co.solve(10)
    assert "NOT_FINITE" in str(e.value)
    p1, _, _ = co.get_window()
    np.testing.assert_array_equal(p0, p1)           # no trial was ever accepted: the states did not move
    cr.load(w)
    try:
        cr.solve(10)
    except vio.VioError:
        pass
    np.testing.assert_array_equal(cr.get_window()[0], p0)
