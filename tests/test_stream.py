"""End-to-end stream: solve -> re-anchor -> MargOldFrame -> slide, over consecutive keyframes (SURVEY.md 8f-3,
BASELINE.json configs[4] stand-in: the reference's own simulator inputs are not shipped, so the stream is synthetic
and the 'reference trajectory' is the one the compiled reference backend produces on the same stream).
Criterion of the north star: ATE within 1 % of the reference trajectory."""
import numpy as np
import pytest


def run(vio, lib, n_frames=24, per_frame=25, seed=3, **kw):
    st = vio.stream.SyntheticStream(n_frames=n_frames, landmarks_per_frame=per_frame, seed=seed)
    drv = vio.stream.StreamDriver(lib, st, **kw)
    traj = drv.run()
    return drv, traj, drv.ground_truth()


@pytest.mark.ref
def test_oracle_stream_tracks_the_reference_backend(vio, oracle_lib, ref_lib):
    do, to, gt = run(vio, oracle_lib)
    dr, tr, _ = run(vio, ref_lib)
    ate_o, ate_r = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(tr, gt)
    assert ate_r < 0.1                                  # the reference publishes 0.04 m on its own simulation
    assert abs(ate_o - ate_r) <= 0.01 * ate_r           # north star: within 1 %
    # frame by frame the two stay within 1e-6 m until a window stops on the other side of the 1e-5 chi2-decrease test
    # (problem.cc:240: 5 iterations here, 10 there) or of a step rejection; from there on they sit at different points
    # of the same flat valley (chi2 equal to 1e-3): millimetres, ATE within 0.3 %
    assert np.abs(to[:, 1:4] - tr[:, 1:4]).max() < 3e-3
    same_stops(do.reports, dr.reports)


def same_stops(ra, rb):
    """The LM loops stop at the same point up to the knife edge of the 1e-5 chi2-decrease test (problem.cc:240): the
    solved chi2 agree, the iteration counts agree on (nearly) every window."""
    np.testing.assert_allclose([r.final_chi2 for r in ra], [r.final_chi2 for r in rb], rtol=2e-2)   # streams compound their stops
    ia, ib = [r.iterations for r in ra], [r.iterations for r in rb]
    assert sum(a != b for a, b in zip(ia, ib)) <= max(2, len(ia) // 6), (ia, ib)


def test_stream_writes_tum_format(vio, oracle_lib, tmp_path):
    _, traj, _ = run(vio, oracle_lib, n_frames=14, per_frame=12)
    p = tmp_path / "pose_output.txt"
    vio.stream.write_tum(str(p), traj)
    rows = [l.split() for l in open(p)]
    assert len(rows) == len(traj) and all(len(r) == 8 for r in rows)      # stamp px py pz qx qy qz qw (System.cpp:438)
    q = np.array([[float(v) for v in r[4:]] for r in rows])
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_hip_stream_tracks_the_oracle(vio, oracle_lib, hip_lib):
    do, to, gt = run(vio, oracle_lib, n_frames=30, per_frame=40, seed=5)
    dh, th, _ = run(vio, hip_lib, n_frames=30, per_frame=40, seed=5)
    ate_o, ate_h = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(th, gt)
    assert ate_h < 0.1
    assert abs(ate_h - ate_o) <= 0.01 * ate_o
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 1e-3
    same_stops(dh.reports, do.reports)


def test_stream_with_triangulated_depths(vio, oracle_lib):
    """8f-2 inside 8f-3: new landmarks get their first depth from FeatureManager::triangulate on the current pose
    estimates (as Estimator::solveOdometry does) instead of from the perturbed ground truth; the trajectory error
    stays at the level of the reference's published ATE (0.04 m on its own 20 s simulation)."""
    drv, traj, gt = run(vio, oracle_lib, n_frames=24, per_frame=25, triangulate=True)
    assert drv.n_triangulated > 300 and drv.have_depth.sum() > 100
    assert vio.stream.ate_rmse(traj, gt) < 0.1
    true_inv = 1.0 / np.array(drv.s.lm_depth)
    used = drv.have_depth
    assert np.median(np.abs(drv.inv_depth[used] - true_inv[used]) / true_inv[used]) < 0.05


@pytest.mark.gpu
def test_hip_stream_with_triangulation_tracks_the_oracle(vio, oracle_lib, hip_lib):
    do, to, gt = run(vio, oracle_lib, n_frames=22, per_frame=30, seed=9, triangulate=True)
    dh, th, _ = run(vio, hip_lib, n_frames=22, per_frame=30, seed=9, triangulate=True)
    assert vio.stream.ate_rmse(th, gt) < 0.1
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 1e-3
    np.testing.assert_allclose(dh.inv_depth, do.inv_depth, rtol=1e-4)


def test_stream_with_non_keyframes(vio, oracle_lib):
    """MARGIN_SECOND_NEW in the loop: every third frame is not a keyframe, so when it is the second-newest one it is
    marginalised instead of the oldest (MargNewFrame, estimator.cpp:830-901), its observations are dropped and its IMU
    samples merged into the interval before it.  The window then spans more than 11 consecutive frames."""
    st = vio.stream.SyntheticStream(n_frames=26, landmarks_per_frame=25, seed=4, track_len=7)
    drv = vio.stream.StreamDriver(oracle_lib, st, nonkey_every=3)
    traj = drv.run()
    gt = drv.ground_truth()
    assert vio.MARG_SECOND_NEW in drv.flags and vio.MARG_OLD in drv.flags
    assert drv.frames[-1] - drv.frames[0] > vio.WINDOW_SIZE                 # frames were skipped inside the window
    assert max(p["sum_dt"] for p in drv.preint) > 1.5 * st.frame_dt           # merged pre-integrations
    assert vio.stream.ate_rmse(traj, gt) < 0.1
    assert all(np.isfinite(r.final_chi2) for r in drv.reports)


@pytest.mark.ref
def test_non_keyframe_stream_tracks_the_reference_backend(vio, oracle_lib, ref_lib):
    def go(lib):
        st = vio.stream.SyntheticStream(n_frames=22, landmarks_per_frame=20, seed=6, track_len=7)
        d = vio.stream.StreamDriver(lib, st, nonkey_every=3)
        return d, d.run(), d.ground_truth()
    do, to, gt = go(oracle_lib)
    dr, tr, _ = go(ref_lib)
    ate_o, ate_r = vio.stream.ate_rmse(to, gt), vio.stream.ate_rmse(tr, gt)
    assert abs(ate_o - ate_r) <= 0.01 * ate_r
    assert np.abs(to[:, 1:4] - tr[:, 1:4]).max() < 1e-3
    assert do.flags == dr.flags


@pytest.mark.gpu
def test_hip_non_keyframe_stream_tracks_the_oracle(vio, oracle_lib, hip_lib):
    def go(lib):
        st = vio.stream.SyntheticStream(n_frames=24, landmarks_per_frame=30, seed=8, track_len=7)
        d = vio.stream.StreamDriver(lib, st, nonkey_every=3, triangulate=True)
        return d, d.run(), d.ground_truth()
    do, to, gt = go(oracle_lib)
    dh, th, _ = go(hip_lib)
    assert vio.stream.ate_rmse(th, gt) < 0.1
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 1e-3
    assert dh.flags == do.flags and vio.MARG_SECOND_NEW in dh.flags


def test_ape_metric_reproduces_the_reference_published_statistics(vio):
    """The metric of the stream configuration is `evo_ape tum ground-truth.txt vins-estimation.txt -va`
    (README.md:169,216 of the reference's assignment 17).  tests/golden/ape_reference.npz holds the three simulation
    trajectories the reference ships together with the statistics it published for them (summary.csv): stamp
    association + Umeyama alignment + translation error must reproduce them."""
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "ape_reference.npz"))
    for name in ("overestimate", "proper_prior", "underestimate"):
        st = vio.stream.ape_stats(z["est_" + name], z["ground_truth"], align=True)
        for k in ("rmse", "mean", "median", "std", "min", "max", "sse"):
            assert abs(st[k] - float(z["stat_%s_%s" % (name, k)])) <= 1e-9 * max(1.0, abs(st[k])), (name, k)
    # the TUM reader/writer round trip keeps the statistics
    assert vio.stream.ape_stats(z["est_proper_prior"], z["ground_truth"], align=False)["rmse"] > 1.0   # unaligned frames differ


def test_simulator_file_formats_round_trip(vio, oracle_lib, tmp_path):
    """A stream written in the reference simulator's file formats (imu_pose.txt + keyframe/all_points_<n>.txt, SURVEY.md
    appendix B) and read back drives the backend to the same trajectory as the in-memory stream."""
    st = vio.stream.SyntheticStream(n_frames=16, landmarks_per_frame=14, seed=5)
    vio.stream.write_simulator_files(st, str(tmp_path))
    head = open(tmp_path / "imu_pose.txt").readline().strip().split(",")
    assert head[:14] == list(vio.stream.IMU_COLUMNS)
    assert open(tmp_path / "keyframe" / "all_points_3.txt").readline().strip().split(",") == list(vio.stream.KEYFRAME_COLUMNS)
    fs = vio.stream.SimulatorFileStream(str(tmp_path))
    assert fs.n_frames == st.n_frames and fs.lm_host == st.lm_host and fs.has_ground_truth
    np.testing.assert_allclose(fs.times, st.times, rtol=0, atol=1e-12)
    np.testing.assert_allclose(fs.P, st.P, atol=1e-12)
    np.testing.assert_allclose(fs.V, st.V, atol=1e-12)
    np.testing.assert_allclose(fs.lm_depth, st.lm_depth, rtol=1e-10)
    for a, b in zip(fs.preint, st.preint):
        np.testing.assert_allclose(a["delta_p"], b["delta_p"], atol=1e-11)
        np.testing.assert_allclose(a["covariance"], b["covariance"], rtol=1e-7, atol=1e-18)
    fs.init_noise = st.init_noise           # same initial depth errors
    ta = vio.stream.StreamDriver(oracle_lib, st).run()
    tb = vio.stream.StreamDriver(oracle_lib, fs).run()
    # the files carry the samples to the last bit but dt = t[j+1] - t[j] and the quaternions differ in the last place;
    # cond(H + lambda I) ~ 1e14 and the compounding windows turn that into ~1e-6 m after 5 windows
    assert np.abs(ta - tb).max() <= 1e-4
    # image stamps that fall between IMU samples (30 Hz camera, 200 Hz IMU): the interval ends on an interpolated sample
    imu = np.loadtxt(tmp_path / "imu_pose.txt", delimiter=",", skiprows=1)
    t_mid = 0.5 * (imu[40, 0] + imu[41, 0])
    lines = open(tmp_path / "keyframe" / "all_points_2.txt").read().splitlines()
    with open(tmp_path / "keyframe" / "all_points_2.txt", "w") as f:
        f.write(lines[0] + "\n")
        for l in lines[1:]:
            f.write(",".join([repr(float(t_mid))] + l.split(",")[1:]) + "\n")
    fs2 = vio.stream.SimulatorFileStream(str(tmp_path))
    assert abs(sum(fs2.imu[1]["dt"]) - (t_mid - fs2.times[1])) <= 1e-12 and len(fs2.imu[1]["dt"]) == 21
    assert abs(fs2.preint[1]["sum_dt"] + fs2.preint[2]["sum_dt"] - 0.2) <= 1e-9


@pytest.mark.gpu
def test_hip_runs_a_stream_read_from_simulator_files(vio, oracle_lib, hip_lib, tmp_path):
    """The file-format path end to end on the GPU: write the simulator's files, read them back, run the HIP backend over
    the stream with triangulated depths and a non-keyframe every fourth frame; same trajectory as the oracle's."""
    st = vio.stream.SyntheticStream(n_frames=22, landmarks_per_frame=20, seed=9)
    vio.stream.write_simulator_files(st, str(tmp_path))
    fs = vio.stream.SimulatorFileStream(str(tmp_path))
    dh = vio.stream.StreamDriver(hip_lib, fs, triangulate=True, nonkey_every=4)
    do = vio.stream.StreamDriver(oracle_lib, fs, triangulate=True, nonkey_every=4)
    th, to = dh.run(), do.run()
    gt = do.ground_truth()
    sh, so = vio.stream.ape_stats(th, gt), vio.stream.ape_stats(to, gt)
    assert so["rmse"] < 0.05 and abs(sh["rmse"] - so["rmse"]) <= 0.01 * so["rmse"] + 1e-6
    assert np.abs(th[:, 1:4] - to[:, 1:4]).max() < 1e-3
    assert dh.flags == do.flags
