"""Finite-difference checks of the factor Jacobians in the oracle (the reference has these checks commented out:
edge_reprojection.cc:110-126, integration_base.h:279-445).  They are what pins the IMU factor, whose reference
translation unit cannot be built here (needs <ceres/ceres.h>)."""
import ctypes as C

import numpy as np

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def pose_plus(oracle_lib, pose, delta):
    out = np.ascontiguousarray(pose, dtype=np.float64).copy()
    d = np.ascontiguousarray(delta, dtype=np.float64)
    f = oracle_lib.dll.vioo_pose_plus
    f.restype = None
    f(dp(out), dp(d))
    return out


def reproj(oracle_lib, pi, pj, ext, lam, a, b, jac=True):
    f = oracle_lib.dll.vioo_reproj_edge
    f.restype = None
    r, Jl, Ji, Jj, Je = np.zeros(2), np.zeros(2), np.zeros(12), np.zeros(12), np.zeros(12)
    args = [np.ascontiguousarray(x, dtype=np.float64) for x in (pi, pj, ext)]
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if jac:
        f(dp(args[0]), dp(args[1]), dp(args[2]), C.c_double(lam), dp(a), dp(b), dp(r), dp(Jl), dp(Ji), dp(Jj), dp(Je))
    else:
        f(dp(args[0]), dp(args[1]), dp(args[2]), C.c_double(lam), dp(a), dp(b), dp(r), None, None, None, None)
    return r, Jl.reshape(2, 1), Ji.reshape(2, 6), Jj.reshape(2, 6), Je.reshape(2, 6)


def test_reprojection_jacobians_match_finite_differences(vio, oracle_lib):
    w = vio.synth.make_window(24, seed=11, ragged=True)
    eps = 1e-7
    worst = 0.0
    for e in range(0, w.lm.size, 3):
        pi, pj, ext = w.poses[w.host[e]], w.poses[w.target[e]], w.ext
        lam = w.inv_depth[w.lm[e]]
        _, Jl, Ji, Jj, Je = reproj(oracle_lib, pi, pj, ext, lam, w.pts_i[e], w.pts_j[e])
        fd = (reproj(oracle_lib, pi, pj, ext, lam + eps, w.pts_i[e], w.pts_j[e], False)[0]
              - reproj(oracle_lib, pi, pj, ext, lam - eps, w.pts_i[e], w.pts_j[e], False)[0]) / (2 * eps)
        worst = max(worst, np.abs(fd - Jl[:, 0]).max() / max(1.0, np.abs(Jl).max()))
        for which, J in ((0, Ji), (1, Jj), (2, Je)):
            for k in range(6):
                d = np.zeros(6)
                d[k] = eps
                ps = [pi, pj, ext]
                pp, pm = list(ps), list(ps)
                pp[which] = pose_plus(oracle_lib, ps[which], d)      # perturb through VertexPose::Plus
                pm[which] = pose_plus(oracle_lib, ps[which], -d)
                fd = (reproj(oracle_lib, pp[0], pp[1], pp[2], lam, w.pts_i[e], w.pts_j[e], False)[0]
                      - reproj(oracle_lib, pm[0], pm[1], pm[2], lam, w.pts_i[e], w.pts_j[e], False)[0]) / (2 * eps)
                worst = max(worst, np.abs(fd - J[:, k]).max() / max(1.0, np.abs(J).max()))
    assert worst < 1e-6, worst      # the survey measured <= 2.2e-9 on the reference itself


def imu(oracle_lib, vio, pre, pi, si, pj, sj, jac=True):
    f = oracle_lib.dll.vioo_imu_edge
    f.restype = None
    g = np.array([0, 0, 9.81])
    r = np.zeros(15)
    Js = [np.zeros(90), np.zeros(135), np.zeros(90), np.zeros(135)]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (pi, si, pj, sj)]
    if jac:
        f(C.byref(pre), dp(g), dp(a[0]), dp(a[1]), dp(a[2]), dp(a[3]), dp(r), dp(Js[0]), dp(Js[1]), dp(Js[2]), dp(Js[3]))
    else:
        f(C.byref(pre), dp(g), dp(a[0]), dp(a[1]), dp(a[2]), dp(a[3]), dp(r), None, None, None, None)
    return r, Js[0].reshape(15, 6), Js[1].reshape(15, 9), Js[2].reshape(15, 6), Js[3].reshape(15, 9)


def test_imu_jacobians_match_finite_differences(vio, oracle_lib):
    """Tolerances per block follow the survey's measurement on the reference: J_sb_i is the one that is only
    approximate, because the d r_R / d bg_i block deliberately uses delta_q instead of corrected_delta_q
    (edge_imu.cc:107-109)."""
    rng = np.random.RandomState(3)
    w = vio.synth.make_window(4, seed=2)
    eps = 1e-7
    for k in (0, 4, 9):
        pre = vio.VioPreint.from_dict(w.preint[k])
        pi, pj = w.poses[k], w.poses[k + 1]
        si, sj = w.speed_bias[k].copy(), w.speed_bias[k + 1].copy()
        si[3:] += rng.normal(0, 1e-3, 6)        # non-zero bias offsets so that the bias-correction terms are live
        sj[3:] += rng.normal(0, 1e-3, 6)
        _, Jpi, Jsi, Jpj, Jsj = imu(oracle_lib, vio, pre, pi, si, pj, sj)

        def fd_pose(which, idx):
            d = np.zeros(6)
            d[idx] = eps
            a, b = [pi, pj], [pi, pj]
            a[which] = pose_plus(oracle_lib, a[which], d)
            b[which] = pose_plus(oracle_lib, b[which], -d)
            return (imu(oracle_lib, vio, pre, a[0], si, a[1], sj, False)[0]
                    - imu(oracle_lib, vio, pre, b[0], si, b[1], sj, False)[0]) / (2 * eps)

        def fd_sb(which, idx):
            a, b = [si.copy(), sj.copy()], [si.copy(), sj.copy()]
            a[which][idx] += eps
            b[which][idx] -= eps
            return (imu(oracle_lib, vio, pre, pi, a[0], pj, a[1], False)[0]
                    - imu(oracle_lib, vio, pre, pi, b[0], pj, b[1], False)[0]) / (2 * eps)

        err_pi = max(np.abs(fd_pose(0, c) - Jpi[:, c]).max() for c in range(6))
        err_pj = max(np.abs(fd_pose(1, c) - Jpj[:, c]).max() for c in range(6))
        err_si = max(np.abs(fd_sb(0, c) - Jsi[:, c]).max() for c in range(9))
        err_sj = max(np.abs(fd_sb(1, c) - Jsj[:, c]).max() for c in range(9))
        assert err_pi < 5e-6 and err_pj < 1e-6 and err_sj < 1e-6, (err_pi, err_pj, err_sj)
        assert err_si < 1e-4, err_si


def test_imu_residual_is_zero_on_the_noise_free_trajectory(vio, oracle_lib):
    """Mid-point pre-integration of noise-free 200 Hz samples reproduces the analytic trajectory to O(dt^2)."""
    w = vio.synth.make_window(4, seed=2)
    for k in range(10):
        pre = vio.VioPreint.from_dict(w.preint[k])
        r = imu(oracle_lib, vio, pre, w.poses_gt[k], w.speed_bias_gt[k], w.poses_gt[k + 1], w.speed_bias_gt[k + 1], False)[0]
        assert np.abs(r[0:3]).max() < 2e-5 and np.abs(r[3:6]).max() < 2e-5 and np.abs(r[6:9]).max() < 5e-4, r
        assert np.abs(r[9:]).max() == 0.0
