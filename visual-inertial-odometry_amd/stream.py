"""Sliding-window stream driver over synthetic data.

Host-side control flow of `Estimator::processImage` -> `backendOptimization` -> `slideWindow` for a stream of
keyframes, with any library that exports the C ABI as the backend:

    per new keyframe:   predict its state from the last one with the pre-integrated IMU (processIMU,
                        VM/src/estimator.cpp:105-139), vector2double (:505-547), Solve(10) with the prior
                        (problemSolve :902-1073), double2vector's yaw/position re-anchoring (:549-600),
                        MargOldFrame (:693-829), slide the window by one frame (slideWindowOld :1144-1200).

Simplifications against the reference (this is the "next" row, not the graded hot path): every frame is a keyframe
(always MARGIN_OLD), and a landmark hosted in the marginalised frame leaves the window instead of being re-hosted
(`removeBackShiftDepth`, feature_manager.cpp:276-312).  Poses are written in TUM format (`stamp px py pz qx qy qz qw`,
System.cpp:438) so that `evo_ape tum` can be run on the output.
"""
import math

import numpy as np

from . import synth
from .capi import MARG_OLD, NUM_FRAMES, WINDOW_SIZE


def r2ypr(R):
    """Utility::R2ypr (VM/include/utility/utility.h:68-84), degrees."""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = math.atan2(n[1], n[0])
    p = math.atan2(-n[2], n[0] * math.cos(y) + n[1] * math.sin(y))
    r = math.atan2(a[0] * math.sin(y) - a[1] * math.cos(y), -o[0] * math.sin(y) + o[1] * math.cos(y))
    return np.array([y, p, r]) / math.pi * 180.0


def ypr2r(ypr):
    """Utility::ypr2R (utility.h:86-110), degrees."""
    y, p, r = (v / 180.0 * math.pi for v in ypr)
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    return Rz @ Ry @ Rx


def anchor_gauge(poses_before, poses_after, sb_after):
    """double2vector's re-anchoring of yaw and position to the pre-solve frame 0 (estimator.cpp:551-600)."""
    R0_before = synth.quat_to_rot(poses_before[0, 3:7])
    R0_after = synth.quat_to_rot(poses_after[0, 3:7] / np.linalg.norm(poses_after[0, 3:7]))
    o0, o00 = r2ypr(R0_before), r2ypr(R0_after)
    rot_diff = ypr2r([o0[0] - o00[0], 0, 0])
    if abs(abs(o0[1]) - 90) < 1.0 or abs(abs(o00[1]) - 90) < 1.0:
        rot_diff = R0_before @ R0_after.T
    poses, sb = poses_after.copy(), sb_after.copy()
    for i in range(NUM_FRAMES):
        q = poses_after[i, 3:7] / np.linalg.norm(poses_after[i, 3:7])
        poses[i, 3:7] = synth.rot_to_quat(rot_diff @ synth.quat_to_rot(q))
        poses[i, 0:3] = rot_diff @ (poses_after[i, 0:3] - poses_after[0, 0:3]) + poses_before[0, 0:3]
        sb[i, 0:3] = rot_diff @ sb_after[i, 0:3]
    return poses, sb


class SyntheticStream:
    """Ground truth + measurements of `n_frames` keyframes on the simulator's trajectory."""

    def __init__(self, n_frames=30, landmarks_per_frame=30, track_len=5, seed=0, t0=1.0, frame_dt=0.1, imu_rate=200,
                 pixel_noise=1.0 / synth.FOCAL):
        rng = np.random.RandomState(seed)
        self.n_frames, self.frame_dt, self.t0 = n_frames, frame_dt, t0
        self.times = [t0 + frame_dt * k for k in range(n_frames)]
        gt = [synth.motion_model(t) for t in self.times]
        self.R = np.stack([m.Rwb for m in gt])
        self.P = np.stack([m.twb for m in gt])
        self.V = np.stack([m.vel for m in gt])
        self.Q = np.stack([synth.rot_to_quat(m.Rwb) for m in gt])
        self.ext = np.concatenate([synth.T_IC, synth.rot_to_quat(synth.R_IC)])
        n_sub = int(round(frame_dt * imu_rate))
        dt = frame_dt / n_sub
        self.preint = []            # preint[k]: frame k -> k+1
        for k in range(n_frames - 1):
            ms = [synth.motion_model(self.times[k] + j * dt) for j in range(n_sub + 1)]
            self.preint.append(synth.preintegrate(ms[0].acc, ms[0].gyro, np.zeros(3), np.zeros(3), [dt] * n_sub,
                                                  [m.acc for m in ms[1:]], [m.gyro for m in ms[1:]]))
        # landmarks: hosted in frame h, seen in h+1 .. h+track_len
        self.lm_host, self.lm_px, self.lm_depth, self.lm_obs = [], [], [], []
        for h in range(n_frames - 1):
            for _ in range(landmarks_per_frame):
                px = rng.uniform(-0.5, 0.5, 2)
                depth = rng.uniform(4.0, 10.0)
                pw = self.R[h] @ (synth.R_IC @ (np.array([px[0], px[1], 1.0]) * depth) + synth.T_IC) + self.P[h]
                obs = {}
                for j in range(h + 1, min(h + 1 + track_len, n_frames)):
                    pc = synth.R_IC.T @ (self.R[j].T @ (pw - self.P[j]) - synth.T_IC)
                    pc[2] = max(pc[2], 0.5)
                    obs[j] = pc[0:2] / pc[2] + rng.normal(0.0, pixel_noise, 2)
                self.lm_host.append(h)
                self.lm_px.append(px)
                self.lm_depth.append(depth)
                self.lm_obs.append(obs)
        self.init_noise = rng.normal(size=(len(self.lm_host),))


class StreamDriver:
    def __init__(self, lib, stream, ctx_kwargs=None, pos_noise=0.02, rot_noise=0.005, depth_noise=0.05, seed=1,
                 triangulate=False):
        """triangulate=True: a landmark's first depth comes from FeatureManager::triangulate (vio_triangulate, on the
        current pose estimates, feature_manager.cpp:203-257) the first time it enters a solve, as in
        Estimator::solveOdometry (estimator.cpp:489-503), instead of from the perturbed ground truth."""
        self.lib, self.s = lib, stream
        self.ctx = lib.context(**(ctx_kwargs or {}))
        rng = np.random.RandomState(seed)
        st = stream
        self.start = 0
        self.poses = np.zeros((NUM_FRAMES, 7))
        self.sb = np.zeros((NUM_FRAMES, 9))
        for i in range(NUM_FRAMES):
            th = rng.normal(0.0, rot_noise, 3)
            dq = np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0])
            dq /= np.linalg.norm(dq)
            self.poses[i, 0:3] = st.P[i] + rng.normal(0.0, pos_noise, 3)
            self.poses[i, 3:7] = synth.quat_mul(st.Q[i], dq)
            self.sb[i, 0:3] = st.V[i]
        self.ext = st.ext.copy()
        self.inv_depth = 1.0 / (np.array(st.lm_depth) * (1.0 + depth_noise * st.init_noise))
        self.triangulate = triangulate
        self.have_depth = np.zeros(len(st.lm_host), dtype=bool) if triangulate else np.ones(len(st.lm_host), dtype=bool)
        self.prior = None
        self.trajectory = []        # (stamp, pose[7]) of the newest frame after every solve
        self.reports = []

    def window_arrays(self):
        s, st = self.start, self.s
        ids, lm, host, target, pi, pj = [], [], [], [], [], []
        for l, h in enumerate(st.lm_host):
            # estimator.cpp:979-981: used_num >= 2 and start_frame < WINDOW_SIZE - 2, restricted to this window
            if h < s or h - s >= WINDOW_SIZE - 2:
                continue
            obs = [(j, o) for j, o in sorted(st.lm_obs[l].items()) if j <= s + WINDOW_SIZE]
            if not obs:
                continue
            k = len(ids)
            ids.append(l)
            for j, o in obs:
                lm.append(k); host.append(h - s); target.append(j - s); pi.append(st.lm_px[l]); pj.append(o)
        w = synth.Window(poses=self.poses.copy(), speed_bias=self.sb.copy(), ext=self.ext.copy(),
                         inv_depth=self.inv_depth[ids].copy(), lm=np.array(lm, dtype=np.int32),
                         host=np.array(host, dtype=np.int32), target=np.array(target, dtype=np.int32),
                         pts_i=np.array(pi).reshape(-1, 2), pts_j=np.array(pj).reshape(-1, 2),
                         preint=[st.preint[s + k] for k in range(WINDOW_SIZE)], prior=self.prior,
                         n_landmarks=len(ids), n_observations=len(lm))
        return w, np.array(ids, dtype=np.int64)

    def triangulate_new(self):
        """f_manager.triangulate(Ps, tic, ric) of solveOdometry: depths of the tracks that have none yet."""
        s, st = self.start, self.s
        todo, sf, off, pts = [], [], [0], []
        for l, h in enumerate(st.lm_host):
            if self.have_depth[l] or h < s or h - s >= WINDOW_SIZE - 2:
                continue
            obs = [o for j, o in sorted(st.lm_obs[l].items()) if j <= s + WINDOW_SIZE]
            if not obs:
                continue
            todo.append(l); sf.append(h - s); pts.append(st.lm_px[l]); pts.extend(obs); off.append(off[-1] + 1 + len(obs))
        if not todo:
            return 0
        depth = self.ctx.triangulate(np.array(sf, dtype=np.int32), np.array(off, dtype=np.int64), np.array(pts).reshape(-1, 2),
                                     self.poses, self.ext, -np.ones(len(todo)))
        self.inv_depth[todo] = 1.0 / depth
        self.have_depth[todo] = True
        return len(todo)

    def step(self):
        """One keyframe: solve, re-anchor, marginalise the oldest frame, slide.  Returns False at the end."""
        st = self.s
        if self.triangulate:
            self.triangulate_new()
        w, ids = self.window_arrays()
        self.ctx.load(w)
        rep = self.ctx.solve(10)
        poses, sb, _ = self.ctx.get_window()
        invd = self.ctx.get_landmarks()
        if self.prior is not None:      # estimator.cpp:1040-1049: b/err prior come back updated, H/Jt stay
            b, e = self.ctx.get_prior()
            self.prior = dict(self.prior, b=b[:156].copy(), err=e.copy())
        self.poses, self.sb = anchor_gauge(w.poses, poses, sb)
        self.inv_depth[ids] = invd
        newest = self.start + WINDOW_SIZE
        self.trajectory.append((st.times[newest], self.poses[WINDOW_SIZE].copy()))
        self.reports.append(rep)
        # MargOldFrame on the re-anchored states (backendOptimization, estimator.cpp:1086-1092)
        w2, _ = self.window_arrays()
        self.ctx.load(w2)
        self.prior = self.ctx.marginalize(MARG_OLD)
        if newest + 1 >= st.n_frames:
            return False
        # slideWindowOld + processIMU prediction of the new frame
        self.poses[:-1], self.sb[:-1] = self.poses[1:].copy(), self.sb[1:].copy()
        pre = st.preint[newest]
        dt = pre["sum_dt"]
        Ri = synth.quat_to_rot(self.poses[WINDOW_SIZE - 1, 3:7])
        g = np.array([0.0, 0.0, synth.G_NORM])
        Pi, Vi = self.poses[WINDOW_SIZE - 1, 0:3], self.sb[WINDOW_SIZE - 1, 0:3]
        self.poses[WINDOW_SIZE, 0:3] = Pi + Vi * dt - 0.5 * g * dt * dt + Ri @ pre["delta_p"]
        self.poses[WINDOW_SIZE, 3:7] = synth.quat_mul(self.poses[WINDOW_SIZE - 1, 3:7], pre["delta_q"])
        self.sb[WINDOW_SIZE, 0:3] = Vi - g * dt + Ri @ pre["delta_v"]
        self.sb[WINDOW_SIZE, 3:9] = self.sb[WINDOW_SIZE - 1, 3:9]
        self.start += 1
        return True

    def run(self):
        while self.step():
            pass
        return np.array([np.concatenate([[t], p]) for t, p in self.trajectory])

    def ground_truth(self):
        st = self.s
        return np.array([np.concatenate([[t], st.P[k], st.Q[k]])
                         for k, t in enumerate(st.times) if k >= WINDOW_SIZE])[:len(self.trajectory)]


def ate_rmse(traj, gt):
    """Translation APE without alignment (the windows are anchored to the initial frame), RMSE in metres."""
    d = traj[:, 1:4] - gt[:, 1:4]
    return float(np.sqrt((d * d).sum(axis=1).mean()))


def write_tum(path, traj):
    with open(path, "w") as f:
        for row in traj:
            f.write(" ".join("%.9f" % v for v in row) + "\n")
