"""FeatureManager::triangulate: tracks -> depths.

The pin against the reference's own FeatureManager (compiled from VM/src/feature_manager.cpp) is
tests/test_feature_manager_golden.py.  Here, second opinions and sizes the golden file does not hold: the oracle's
restatement (oracle/vio_oracle.c: vio_triangulate, citing feature_manager.cpp:203-257) against an independent
transcription of the same algorithm with numpy.linalg.svd (LAPACK) on the very matrix svd_A the reference builds, and
the HIP kernel (Jacobi on A^T A, one thread per track) against the oracle through the C ABI up to 50 000 tracks."""
import numpy as np
import pytest


def make_tracks(vio, n, seed, noise=0.0, have_depth_frac=0.2):
    rng = np.random.RandomState(seed)
    synth = vio.synth
    t = [1.0 + 0.1 * k for k in range(11)]
    gt = [synth.motion_model(tt) for tt in t]
    poses = np.zeros((11, 7))
    for k, m in enumerate(gt):
        poses[k, 0:3] = m.twb
        poses[k, 3:7] = synth.rot_to_quat(m.Rwb)
    ext = np.concatenate([synth.T_IC, synth.rot_to_quat(synth.R_IC)])
    sf, off, pts, true_depth = [], [0], [], []
    for _ in range(n):
        s = int(rng.randint(0, 10))
        k = int(rng.randint(1, min(6, 11 - s) + 1))          # 1 .. 6 observations (1: skipped by the rule)
        px = rng.uniform(-0.4, 0.4, 2)
        dep = rng.uniform(3.0, 12.0)
        pw = gt[s].Rwb @ (synth.R_IC @ (np.array([px[0], px[1], 1.0]) * dep) + synth.T_IC) + gt[s].twb
        for j in range(s, s + k):
            pc = synth.R_IC.T @ (gt[j].Rwb.T @ (pw - gt[j].twb) - synth.T_IC)
            pts.append(pc[0:2] / pc[2] + rng.normal(0.0, noise, 2) * (j != s))
        sf.append(s); off.append(off[-1] + k); true_depth.append(dep)
    depth0 = np.where(rng.uniform(size=n) < have_depth_frac, rng.uniform(2.0, 9.0, n), -1.0)
    return (np.array(sf, dtype=np.int32), np.array(off, dtype=np.int64), np.array(pts).reshape(-1, 2), poses, ext,
            depth0, np.array(true_depth))


def numpy_triangulate(vio, sf, off, pts, poses, ext, depth0, init_depth=5.0):
    """feature_manager.cpp:203-257 line by line, with numpy.linalg.svd standing where Eigen::JacobiSVD stands."""
    synth = vio.synth
    ric, tic = synth.quat_to_rot(ext[3:7]), ext[0:3]
    Rs = [synth.quat_to_rot(poses[k, 3:7]) for k in range(11)]
    out = depth0.copy()
    for i in range(len(sf)):
        k = off[i + 1] - off[i]
        if not (k >= 2 and sf[i] < 10 - 2):
            continue
        if out[i] > 0:
            continue
        t0 = poses[sf[i], 0:3] + Rs[sf[i]] @ tic
        R0 = Rs[sf[i]] @ ric
        A = np.zeros((2 * k, 4))
        for j in range(k):
            f = sf[i] + j
            t1 = poses[f, 0:3] + Rs[f] @ tic
            R1 = Rs[f] @ ric
            t = R0.T @ (t1 - t0)
            R = R0.T @ R1
            P = np.hstack([R.T, (-R.T @ t).reshape(3, 1)])
            v = np.array([pts[off[i] + j, 0], pts[off[i] + j, 1], 1.0])
            v /= np.linalg.norm(v)
            A[2 * j] = v[0] * P[2] - v[2] * P[0]
            A[2 * j + 1] = v[1] * P[2] - v[2] * P[1]
        V = np.linalg.svd(A)[2][-1]
        d = V[2] / V[3]
        out[i] = init_depth if d < 0.1 else d
    return out


@pytest.mark.parametrize("noise", [0.0, 1.0 / 460.0])
def test_oracle_triangulate_matches_lapack_svd(vio, oracle_lib, noise):
    sf, off, pts, poses, ext, d0, true = make_tracks(vio, 400, seed=3, noise=noise)
    ctx = oracle_lib.context()
    got = ctx.triangulate(sf, off, pts, poses, ext, d0)
    want = numpy_triangulate(vio, sf, off, pts, poses, ext, d0)
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=0)
    k = np.diff(off)
    done = (k >= 2) & (sf < 8) & (d0 <= 0)
    assert done.sum() > 100
    np.testing.assert_array_equal(got[~done], d0[~done])          # skipped tracks are untouched (:207-211)
    if noise == 0.0:                                              # exact data: the DLT null vector is the true point
        np.testing.assert_allclose(got[done], true[done], rtol=1e-6)


def test_triangulate_edge_cases(vio, oracle_lib):
    sf, off, pts, poses, ext, d0, _ = make_tracks(vio, 50, seed=5)
    ctx = oracle_lib.context()
    # no tracks at all
    assert ctx.triangulate(np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros((0, 2)), poses, ext, np.zeros(0)).size == 0
    # a point behind the camera triangulates to a negative depth -> INIT_DEPTH (:252-255)
    sf1, off1 = np.array([0], np.int32), np.array([0, 3], np.int64)
    p = pts[off[0]:off[0] + 1].repeat(3, axis=0).copy()
    p[1] += [0.3, 0.0]; p[2] -= [0.3, 0.0]                        # inconsistent rays: whatever comes out, the rule holds
    got = ctx.triangulate(sf1, off1, p, poses, ext, np.array([-1.0]), init_depth=7.5)
    assert got[0] == 7.5 or got[0] >= 0.1
    # a track leaving the window is rejected
    with pytest.raises(vio.VioError):
        ctx.triangulate(np.array([9], np.int32), np.array([0, 4], np.int64), np.zeros((4, 2)), poses, ext, np.array([-1.0]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,noise", [(1, 0.0), (777, 1.0 / 460.0), (50000, 1.0 / 460.0)])
def test_hip_triangulate_matches_oracle(vio, hip_lib, oracle_lib, n, noise):
    sf, off, pts, poses, ext, d0, true = make_tracks(vio, n, seed=11 + n, noise=noise)
    got = hip_lib.context().triangulate(sf, off, pts, poses, ext, d0)
    want = oracle_lib.context().triangulate(sf, off, pts, poses, ext, d0)
    # both take the smallest eigenvector of A^T A (Jacobi rotations vs. tridiagonal QL): agreement to rounding
    # amplified by the conditioning of the 4x4 problem
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=0)
    k = np.diff(off)
    done = (k >= 2) & (sf < 8) & (d0 <= 0)
    np.testing.assert_array_equal(got[~done], d0[~done])
    if noise == 0.0 and done.any():
        np.testing.assert_allclose(got[done], true[done], rtol=1e-6)
