"""Synthetic sliding windows for tests and bench.py.

Trajectory: the simulator's analytic MotionModel (A/17-vins-initialization/simulator/src/imu.cpp:76-117,
an ellipse 15x20 m with a sinusoidal z and small roll/pitch) sampled at keyframe times
t_i = t0 + 0.1*i.  IMU: 200 Hz noise-free samples pre-integrated with the mid-point rule of
VM/include/factor/integration_base.h:54-158 (restated here in numpy; this is host-side L2 work in
the reference, `Estimator::processIMU`).  Landmarks: host frame l mod 7, K observations in the
following frames, pixel noise 1/460, Cauchy-weighted information (460/1.5)^2.

Everything is numpy; nothing here touches the oracle.
"""
import math
from types import SimpleNamespace

import numpy as np

from .capi import NUM_FRAMES, WINDOW_SIZE

# VM/config/vio_simulation.yaml:30-42,60-63,79
R_IC = np.array([[0.0, 0.0, -1.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
T_IC = np.array([0.05, 0.04, 0.03])
ACC_N, GYR_N, ACC_W, GYR_W = 0.2687, 0.2121, 7.07e-6, 7.07e-7
G_NORM = 9.81
FOCAL = 460.0


# ---- small rotation helpers (Eigen conventions, quaternion stored x,y,z,w) -------------------
def rot_to_quat(R):
    """Eigen::Quaterniond(Matrix3d) (Shepperd), returns (x,y,z,w)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = math.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * t
    t = 0.5 / t
    q[3] = (R[k, j] - R[j, k]) * t
    q[j] = (R[j, i] + R[i, j]) * t
    q[k] = (R[k, i] + R[i, k]) * t
    return q


def quat_to_rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


# ---- simulator MotionModel (imu.cpp:76-117) --------------------------------------------------
def motion_model(t):
    ex, ey, z, K1, K = 15.0, 20.0, 1.0, 10.0, math.pi / 10
    pos = np.array([ex * math.cos(K * t) + 5, ey * math.sin(K * t) + 5, z * math.sin(K1 * K * t) + 5])
    dp = np.array([-K * ex * math.sin(K * t), K * ey * math.cos(K * t), z * K1 * K * math.cos(K1 * K * t)])
    K2 = K * K
    ddp = np.array([-K2 * ex * math.cos(K * t), -K2 * ey * math.sin(K * t),
                    -z * K1 * K1 * K2 * math.sin(K1 * K * t)])
    kr, kp = 0.1, 0.2
    roll, pitch, yaw = kr * math.cos(t), kp * math.sin(t), K * t
    rates = np.array([-kr * math.sin(t), kp * math.cos(t), K])
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    Rwb = np.array([[cy * cp, cy * sp * sr - sy * cr, sy * sr + cy * cr * sp],
                    [sy * cp, cy * cr + sy * sr * sp, sp * sy * cr - cy * sr],
                    [-sp, cp * sr, cp * cr]])
    E = np.array([[1, 0, -sp], [0, cr, sr * cp], [0, -sr, cr * cp]])
    gyro = E @ rates
    gn = np.array([0, 0, -G_NORM])
    acc = Rwb.T @ (ddp - gn)
    return SimpleNamespace(Rwb=Rwb, twb=pos, vel=dp, gyro=gyro, acc=acc)


# ---- IntegrationBase mid-point pre-integration (integration_base.h:14-158) --------------------
def preintegrate(acc0, gyr0, ba, bg, dts, accs, gyrs, acc_n=ACC_N, gyr_n=GYR_N, acc_w=ACC_W, gyr_w=GYR_W):
    noise = np.zeros((18, 18))
    noise[0:3, 0:3] = acc_n * acc_n * np.eye(3)
    noise[3:6, 3:6] = gyr_n * gyr_n * np.eye(3)
    noise[6:9, 6:9] = acc_n * acc_n * np.eye(3)
    noise[9:12, 9:12] = gyr_n * gyr_n * np.eye(3)
    noise[12:15, 12:15] = acc_w * acc_w * np.eye(3)
    noise[15:18, 15:18] = gyr_w * gyr_w * np.eye(3)
    jac, cov = np.eye(15), np.zeros((15, 15))
    dp, dv, dq, sum_dt = np.zeros(3), np.zeros(3), np.array([0.0, 0.0, 0.0, 1.0]), 0.0
    a0, g0 = np.array(acc0, dtype=float), np.array(gyr0, dtype=float)
    I3 = np.eye(3)
    for dt, a1, g1 in zip(dts, accs, gyrs):
        Rd = quat_to_rot(dq)
        un_acc_0 = Rd @ (a0 - ba)
        un_gyr = 0.5 * (g0 + g1) - bg
        rq = quat_mul(dq, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        Rr = quat_to_rot(rq)    # un-normalised on purpose: the reference rotates with result_delta_q before
                                # delta_q.normalize() (integration_base.h:66-67,142); same polynomial as Eigen's q*v
        un_acc_1 = Rr @ (a1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        Rwx, Ra0, Ra1 = skew(un_gyr), skew(a0 - ba), skew(a1 - ba)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rd @ Ra0 * dt * dt + -0.25 * Rr @ Ra1 @ (I3 - Rwx * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I3 - Rwx * dt
        F[3:6, 12:15] = -1.0 * I3 * dt
        F[6:9, 3:6] = -0.5 * Rd @ Ra0 * dt + -0.5 * Rr @ Ra1 @ (I3 - Rwx * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rd * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rd * dt
        V[6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        jac = F @ jac
        cov = F @ cov @ F.T + V @ noise @ V.T
        dp, dv = rp, rv
        dq = rq / np.linalg.norm(rq)
        sum_dt += dt
        a0, g0 = np.array(a1, dtype=float), np.array(g1, dtype=float)
    return {"sum_dt": sum_dt, "delta_p": dp, "delta_q": dq, "delta_v": dv, "linearized_ba": np.array(ba, dtype=float),
            "linearized_bg": np.array(bg, dtype=float), "jacobian": jac, "covariance": cov}


class Window(SimpleNamespace):
    """Flat arrays in exactly the shapes the C ABI takes (see include/vio_backend.h)."""

    def copy(self):
        d = {}
        for k, v in self.__dict__.items():
            d[k] = v.copy() if isinstance(v, np.ndarray) else (list(v) if isinstance(v, list) else v)
        return Window(**d)


def make_window(n_landmarks, seed=42, obs_per_landmark=4, t0=1.0, imu_rate=200, frame_dt=0.1,
                pos_noise=0.02, rot_noise=0.005, depth_noise=0.05, pixel_noise=1.0 / FOCAL, ragged=False,
                outlier_fraction=0.0):
    """Build one 11-frame window.  `ragged=True` draws K per landmark from 1..min(10-host, 10) and hosts
    from 0..8 so that every (host, track-length) pattern a VINS window can contain shows up."""
    rng = np.random.RandomState(seed)
    times = [t0 + frame_dt * i for i in range(NUM_FRAMES)]
    gt = [motion_model(t) for t in times]
    poses_gt = np.zeros((NUM_FRAMES, 7))
    sb_gt = np.zeros((NUM_FRAMES, 9))
    for i, m in enumerate(gt):
        poses_gt[i, 0:3] = m.twb
        poses_gt[i, 3:7] = rot_to_quat(m.Rwb)
        sb_gt[i, 0:3] = m.vel
    ext = np.concatenate([T_IC, rot_to_quat(R_IC)])

    # IMU: samples at t_i + k/imu_rate, k = 0..n (integration_base.h:30-36: ctor takes sample 0)
    n_sub = int(round(frame_dt * imu_rate))
    dt = frame_dt / n_sub
    preint = []
    for i in range(WINDOW_SIZE):
        ms = [motion_model(times[i] + k * dt) for k in range(n_sub + 1)]
        preint.append(preintegrate(ms[0].acc, ms[0].gyro, np.zeros(3), np.zeros(3), [dt] * n_sub,
                                   [m.acc for m in ms[1:]], [m.gyro for m in ms[1:]]))

    # landmarks + observations, grouped by landmark as estimator.cpp:975-1016 emits them
    N = int(n_landmarks)
    if ragged:
        host_of = rng.randint(0, WINDOW_SIZE - 1, size=N)
        k_of = np.array([rng.randint(1, WINDOW_SIZE - h + 1) for h in host_of])
    else:
        host_of = np.arange(N) % max(1, min(7, WINDOW_SIZE + 1 - int(obs_per_landmark)))    # host + K <= 10
        k_of = np.full(N, int(obs_per_landmark))
    px = rng.uniform(-0.5, 0.5, size=(N, 2))
    depth = rng.uniform(4.0, 10.0, size=N)
    Rg = np.stack([m.Rwb for m in gt])
    Pg = np.stack([m.twb for m in gt])
    # vectorised over landmarks; observation order = landmark-major, then target frame (estimator.cpp:975-1016)
    pc = np.concatenate([px, np.ones((N, 1))], axis=1) * depth[:, None]
    pb = pc @ R_IC.T + T_IC
    pw = np.einsum("nij,nj->ni", Rg[host_of], pb) + Pg[host_of]
    lm = np.repeat(np.arange(N, dtype=np.int32), k_of)
    first = np.concatenate([[0], np.cumsum(k_of)[:-1]]) if N else np.zeros(0, dtype=np.int64)
    within = np.arange(lm.size) - np.repeat(first, k_of)
    host = host_of[lm].astype(np.int32)
    target = (host + 1 + within).astype(np.int32)
    pbj = np.einsum("nji,nj->ni", Rg[target], pw[lm] - Pg[target])
    pcj = (pbj - T_IC) @ R_IC
    pcj[:, 2] = np.maximum(pcj[:, 2], 0.5)      # behind / too close: keep the edge count deterministic
    pts_j = pcj[:, 0:2] / pcj[:, 2:3] + rng.normal(0.0, pixel_noise, size=(lm.size, 2))
    pts_i = px[lm].copy()
    if outlier_fraction > 0 and lm.size:
        bad = rng.rand(lm.size) < outlier_fraction
        pts_j[bad] += rng.normal(0.0, 0.05, size=(int(bad.sum()), 2))

    # initial state = GT + noise
    poses = poses_gt.copy()
    poses[:, 0:3] += rng.normal(0.0, pos_noise, size=(NUM_FRAMES, 3))
    for i in range(NUM_FRAMES):
        th = rng.normal(0.0, rot_noise, size=3)
        dq = np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0])
        dq /= np.linalg.norm(dq)
        poses[i, 3:7] = quat_mul(poses_gt[i, 3:7], dq)
    inv_depth = 1.0 / (depth * (1.0 + depth_noise * rng.normal(size=N)))
    return Window(poses=poses, speed_bias=sb_gt.copy(), ext=ext, inv_depth=inv_depth, lm=lm, host=host, target=target,
                  pts_i=pts_i, pts_j=pts_j, preint=preint, prior=None, poses_gt=poses_gt, speed_bias_gt=sb_gt,
                  inv_depth_gt=1.0 / depth, seed=seed, n_landmarks=N, n_observations=int(lm.size))


def make_window_xyz(n_landmarks, seed=42, xyz_noise=0.05, **kw):
    """The same window with its landmarks as world points (VertexPointXYZ) and one EdgeReprojectionXYZ per observation
    (VM/src/backend/edge_reprojection.cc:130-180): a landmark hosted in frame h and tracked into h+1..h+K of
    make_window() is observed K+1 times here — the host observation is an observation like the others.  Observations
    are listed landmark-major, frames ascending.  Fields: xyz [N][3], lm / frame / pts [M]."""
    w = make_window(n_landmarks, seed=seed, **kw)
    rng = np.random.RandomState(seed + 7919)
    N = w.n_landmarks
    host_of = np.zeros(N, dtype=np.int64)
    px = np.zeros((N, 2))
    host_of[w.lm] = w.host
    px[w.lm] = w.pts_i
    Rg = np.stack([quat_to_rot(q) for q in w.poses_gt[:, 3:7]])
    Pg = w.poses_gt[:, 0:3]
    pc = np.concatenate([px, np.ones((N, 1))], axis=1) / w.inv_depth_gt[:, None]
    pb = pc @ R_IC.T + T_IC
    xyz_gt = np.einsum("nij,nj->ni", Rg[host_of], pb) + Pg[host_of]
    k_of = np.bincount(w.lm, minlength=N)
    first = np.concatenate([[0], np.cumsum(k_of)[:-1]]) if N else np.zeros(0, dtype=np.int64)
    M = int(w.lm.size + N)
    lm = np.repeat(np.arange(N, dtype=np.int32), k_of + 1)
    pos = np.concatenate([[0], np.cumsum(k_of + 1)[:-1]]) if N else np.zeros(0, dtype=np.int64)
    frame = np.zeros(M, dtype=np.int32)
    pts = np.zeros((M, 2))
    frame[pos] = host_of
    pts[pos] = px + rng.normal(0.0, kw.get("pixel_noise", 1.0 / FOCAL), size=(N, 2))
    rest = np.ones(M, dtype=bool)
    rest[pos] = False
    frame[rest] = w.target
    pts[rest] = w.pts_j
    xyz = xyz_gt + rng.normal(0.0, xyz_noise, size=(N, 3))
    return Window(poses=w.poses, speed_bias=w.speed_bias, ext=w.ext, xyz=xyz, lm=lm, frame=frame, pts=pts, preint=w.preint,
                  prior=None, poses_gt=w.poses_gt, speed_bias_gt=w.speed_bias_gt, xyz_gt=xyz_gt, seed=seed, n_landmarks=N,
                  n_observations=M)


def shard_window(w, rank, world):
    """Block-partition the landmarks (with all their observations) over `world` ranks
    (SURVEY.md section 8e): poses, speed-biases, extrinsic, pre-integrations and prior are replicated."""
    N = w.n_landmarks
    lo = (N * rank) // world
    hi = (N * (rank + 1)) // world
    keep = (w.lm >= lo) & (w.lm < hi)
    s = w.copy()
    if getattr(w, "xyz", None) is not None:         # XYZ landmarks (make_window_xyz)
        s.xyz, s.xyz_gt = w.xyz[lo:hi].copy(), w.xyz_gt[lo:hi].copy()
        s.lm = (w.lm[keep] - lo).astype(np.int32)
        s.frame, s.pts = w.frame[keep].copy(), w.pts[keep].copy()
        s.n_landmarks, s.n_observations = hi - lo, int(s.lm.size)
        s.landmark_range = (lo, hi)
        return s
    s.inv_depth = w.inv_depth[lo:hi].copy()
    s.inv_depth_gt = w.inv_depth_gt[lo:hi].copy()
    s.lm = (w.lm[keep] - lo).astype(np.int32)
    s.host = w.host[keep].copy()
    s.target = w.target[keep].copy()
    s.pts_i = w.pts_i[keep].copy()
    s.pts_j = w.pts_j[keep].copy()
    s.n_landmarks = hi - lo
    s.n_observations = int(s.lm.size)
    s.landmark_range = (lo, hi)
    return s


def algorithmic_bytes(n_landmarks, n_observations):
    """B_alg = 44*M + 16*N + B_win per GN iteration."""
    b_win = 1464 + 10 * (10 + 225 + 225 + 6 + 1) * 8 + (171 * 171 + 171) * 8 + 171 * 8
    return 44 * n_observations + 16 * n_landmarks + b_win
