"""Real inertial data: a stretch of the EuRoC MH_05 IMU file and camera stamps the reference ships (VM/config/MH_05_imu0.txt,
MH_05_cam0.txt; fixture tests/golden/mh05_imu_stretch.npz <- make_golden_mh05.py), read and cut at the frame stamps as
VM/test/run_euroc.cpp:26-76 and System.cpp:363-401 do, pre-integrated with euroc_config.yaml's noise parameters
(integration_base.h:54-158) — the nearest runnable stand-in for BASELINE.json configs[0] / [4]: the images and the OpenCV front-end
are not in the tree, so the vision is synthetic on the trajectory the real IMU defines."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR

FIELDS = ("sum_dt", "delta_p", "delta_q", "delta_v", "jacobian", "covariance")


@pytest.fixture(scope="module")
def mh05():
    return dict(np.load(os.path.join(GOLDEN_DIR, "mh05_imu_stretch.npz")))


def as_dict(p):
    return {k: (p.sum_dt if k == "sum_dt" else np.array(getattr(p, k)[:])) for k in FIELDS}


def check_same(got, ref):
    assert abs(got["sum_dt"] - ref["sum_dt"]) < 1e-15
    np.testing.assert_allclose(np.ravel(got["delta_p"]), np.ravel(ref["delta_p"]), rtol=0, atol=2e-15)
    np.testing.assert_allclose(np.ravel(got["delta_q"]), np.ravel(ref["delta_q"]), rtol=0, atol=2e-15)
    np.testing.assert_allclose(np.ravel(got["delta_v"]), np.ravel(ref["delta_v"]), rtol=0, atol=2e-14)
    np.testing.assert_allclose(np.ravel(got["jacobian"]), np.ravel(ref["jacobian"]), rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(np.ravel(got["covariance"]), np.ravel(ref["covariance"]), rtol=1e-10, atol=1e-30)


def test_fixture_is_the_sensor_as_the_reference_reads_it(mh05):
    t, cam = mh05["imu_t"], mh05["cam_t"]
    assert len(cam) == 36 and len(t) > 20 * 35
    d = np.diff(t)
    assert 0.0049 < d.min() and d.max() < 0.0051                    # 200 Hz; absolute seconds as doubles: 2.4e-7 s of stamp resolution
    assert len(np.unique(np.round(d, 9))) > 1                       # ... so the steps are NOT all equal, unlike the simulator's
    np.testing.assert_allclose(np.diff(cam), 0.1, atol=1e-6)       # freq: 10 Hz of the 20 Hz camera
    assert abs(np.linalg.norm(mh05["imu_acc"], axis=1).mean() - 9.8) < 0.5      # gravity is in the accelerometer, whatever the MAV does
    assert float(mh05["acc_n"]) == 0.08 and float(mh05["gyr_w"]) == 2.0e-6 and float(mh05["g_norm"]) == 9.81007


def test_preintegration_of_real_intervals_three_ways(vio, oracle_lib, mh05):
    """IntegrationBase's mid-point propagation of every frame-to-frame interval of the stretch (ragged steps, the interpolated
    sample at the image stamp): the product's host C++ (vio_preintegrate), the oracle's C and the generator's numpy agree; and
    re-propagation with other biases (repropagate, integration_base.h:41-52) is the same call."""
    st = vio.stream.RealImuStream(mh05, landmarks_per_frame=1)
    lib = vio.load_hip()             # host code of the product library: runs without a GPU
    nz = st.noise
    args = (nz["acc_n"], nz["gyr_n"], nz["acc_w"], nz["gyr_w"])
    assert len(st.imu) == 35
    rng = np.random.RandomState(4)
    for k, iv in enumerate(st.imu):
        assert 20 <= len(iv["dt"]) <= 21 and abs(sum(iv["dt"]) - (st.times[k + 1] - st.times[k])) < 1e-9
        for ba, bg in ((np.zeros(3), np.zeros(3)), (rng.normal(0, 0.05, 3), rng.normal(0, 0.005, 3))):
            py = vio.synth.preintegrate(iv["acc0"], iv["gyr0"], ba, bg, iv["dt"], iv["acc"], iv["gyr"], **nz)
            got = as_dict(lib.preintegrate(iv["acc0"], iv["gyr0"], ba, bg, iv["dt"], iv["acc"], iv["gyr"], *args))
            orc = as_dict(oracle_lib.preintegrate(iv["acc0"], iv["gyr0"], ba, bg, iv["dt"], iv["acc"], iv["gyr"], *args))
            check_same(got, py)
            check_same(got, orc)
    # the real sensor moves: rotations of several degrees and velocity changes of decimetres per second within one interval
    assert max(np.linalg.norm(p["delta_q"][0:3]) for p in st.preint) > 0.01
    assert max(np.linalg.norm(p["delta_v"]) for p in st.preint) > 0.5


def test_trajectory_defined_by_the_real_imu_zeroes_the_inertial_residuals(vio, oracle_lib, mh05):
    """the stream's ground truth is the IMU factors' own model: a window at the ground truth without vision has chi2 = 0"""
    st = vio.stream.RealImuStream(mh05, landmarks_per_frame=1)
    w = vio.synth.make_window(0, seed=1)
    w.poses = np.concatenate([st.P[:11], st.Q[:11]], axis=1)
    w.speed_bias = np.concatenate([st.V[:11], np.zeros((11, 6))], axis=1)
    w.ext = st.ext.copy()
    w.preint = st.preint[:10]
    c = oracle_lib.context(gravity=(0.0, 0.0, st.g_norm))
    c.load(w)
    c.linearize()
    assert c.chi2() < 1e-12


def run(vio, lib, mh05, **kw):
    st = vio.stream.RealImuStream(mh05, landmarks_per_frame=30, seed=7)
    drv = vio.stream.StreamDriver(lib, st, seed=2, **kw)
    traj = drv.run()
    return drv, traj, drv.ground_truth()


def test_oracle_runs_the_real_imu_stream(vio, oracle_lib, mh05):
    drv, traj, gt = run(vio, oracle_lib, mh05)
    assert len(traj) == 36 - 10
    # 2.5 m of flight in 3.5 s; evo_ape's SE(3)-aligned RMSE (the reference's metric) 9 mm, the unaligned one carries the first window's
    # gauge (initial pose noise 2 cm): 7 cm
    assert vio.stream.ape_stats(traj, gt)["rmse"] < 0.02 and vio.stream.ate_rmse(traj, gt) < 0.15
    assert all(r.final_chi2 < r.initial_chi2 for r in drv.reports)


@pytest.mark.ref
def test_oracle_real_imu_stream_tracks_the_reference_backend(vio, oracle_lib, ref_lib, mh05):
    do, to, gt = run(vio, oracle_lib, mh05)
    dr, tr, _ = run(vio, ref_lib, mh05)
    ate_o, ate_r = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(tr, gt)
    assert abs(ate_o - ate_r) <= 0.01 * ate_r
    assert np.abs(to[:, 1:4] - tr[:, 1:4]).max() < 3e-3


@pytest.mark.gpu
def test_hip_real_imu_stream_tracks_the_oracle(vio, oracle_lib, hip_lib, mh05):
    do, to, gt = run(vio, oracle_lib, mh05)
    dh, th, _ = run(vio, hip_lib, mh05)
    ate_o, ate_h = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(th, gt)
    assert vio.stream.ape_stats(th, gt)["rmse"] < 0.02
    assert abs(ate_h - ate_o) <= 0.01 * ate_o
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 1e-3


@pytest.mark.gpu
def test_hip_real_imu_stream_with_non_keyframes_and_triangulation(vio, oracle_lib, hip_lib, mh05):
    do, to, gt = run(vio, oracle_lib, mh05, nonkey_every=3, triangulate=True)
    dh, th, _ = run(vio, hip_lib, mh05, nonkey_every=3, triangulate=True)
    ate_o, ate_h = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(th, gt)
    assert abs(ate_h - ate_o) <= 0.02 * max(ate_o, 1e-3)
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 2e-3
