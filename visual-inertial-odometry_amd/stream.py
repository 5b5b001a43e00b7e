"""Sliding-window stream driver over synthetic data.

Host-side control flow of `Estimator::processImage` -> `solveOdometry` -> `slideWindow` for a stream of frames, with
any library that exports the C ABI as the backend:

    per new frame:   predict its state from the last one with the pre-integrated IMU (processIMU,
                     VM/src/estimator.cpp:105-139), [f_manager.triangulate], vector2double (:505-547), Solve(10) with the
                     prior (problemSolve :902-1073), double2vector's yaw/position re-anchoring (:549-600), then
      keyframe       MargOldFrame (:693-829), slideWindowOld (:1187-1199): the oldest frame leaves; landmarks hosted
                     in it move to their next frame with the depth carried over (removeBackShiftDepth,
                     feature_manager.cpp:276-312) or die with it
      non-keyframe   MargNewFrame (:830-901), slideWindowNew (:1201-1206): the second-newest frame leaves, its
                     observations are dropped (removeFront, feature_manager.cpp:331-350), its IMU samples are appended
                     to the interval before it (pre_integrations[frame_count - 1]->push_back, :1163-1176)

Which frames are keyframes is the front-end's call (the parallax test of addFeatureCheckParallax); here it is a
parameter (`nonkey_every`).  Poses are written in TUM format (`stamp px py pz qx qy qz qw`, System.cpp:438) so that
`evo_ape tum` can be run on the output.
"""
import math

import numpy as np

from . import synth
from .capi import MARG_OLD, MARG_SECOND_NEW, NUM_FRAMES, WINDOW_SIZE


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
    R0_after = synth.quat_to_rot(poses_after[0, 3:7])       # not normalised here (:557-561), normalised below (:579)
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
        self.preint, self.imu = [], []      # [k]: frame k -> k+1: pre-integration and the raw samples it was made of
        for k in range(n_frames - 1):
            ms = [synth.motion_model(self.times[k] + j * dt) for j in range(n_sub + 1)]
            self.imu.append(dict(acc0=ms[0].acc, gyr0=ms[0].gyro, dt=[dt] * n_sub, acc=[m.acc for m in ms[1:]],
                                 gyr=[m.gyro for m in ms[1:]]))
            self.preint.append(synth.preintegrate(ms[0].acc, ms[0].gyro, np.zeros(3), np.zeros(3), [dt] * n_sub,
                                                  [m.acc for m in ms[1:]], [m.gyro for m in ms[1:]]))
        make_tracks(self, rng, landmarks_per_frame, track_len, pixel_noise)


IMU_COLUMNS = ("timestamp", "q_w", "q_x", "q_y", "q_z", "p_x", "p_y", "p_z", "gyro_x", "gyro_y", "gyro_z", "acc_x", "acc_y", "acc_z")
KEYFRAME_COLUMNS = ("timestamp", "id", "x", "y", "z", "lambda", "x_normalized", "y_normalized", "u", "v")


def write_simulator_files(stream, out_dir, fx=synth.FOCAL, cx=255.0, cy=255.0):
    """Write a stream in the file formats of the reference's simulator: `imu_pose.txt`
    (timestamp, ground-truth q and p, gyro, acc: simulator/src/utilities.cpp:150-171; the runner reads timestamp, gyro
    and acc by name and ignores the rest, run_vio_simulation.cpp:43-49) and one `keyframe/all_points_<n>.txt` per frame
    (timestamp, id, world point, lambda, normalised point, pixel: utilities.cpp:37-63, read at
    run_vio_simulation.cpp:163-170).  Three extra columns v_x, v_y, v_z carry the ground-truth velocity."""
    import os
    os.makedirs(os.path.join(out_dir, "keyframe"), exist_ok=True)
    st = stream
    with open(os.path.join(out_dir, "imu_pose.txt"), "w") as f:
        f.write(",".join(IMU_COLUMNS + ("v_x", "v_y", "v_z")) + "\n")
        rows = []
        for k, iv in enumerate(st.imu):
            t = st.times[k]
            if k == 0:
                rows.append((t, iv["gyr0"], iv["acc0"]))
            for dt, a, g in zip(iv["dt"], iv["acc"], iv["gyr"]):
                t = t + dt
                rows.append((t, g, a))
            rows[-1] = (st.times[k + 1], rows[-1][1], rows[-1][2])      # the frame stamp itself, not an accumulated sum
        for t, g, a in rows:
            m = synth.motion_model(t)
            q = synth.rot_to_quat(m.Rwb)                                 # (x, y, z, w)
            vals = [t, q[3], q[0], q[1], q[2], *m.twb, *g, *a, *m.vel]
            f.write(",".join(repr(float(v)) for v in vals) + "\n")
    for k in range(st.n_frames):
        with open(os.path.join(out_dir, "keyframe", "all_points_%d.txt" % k), "w") as f:
            f.write(",".join(KEYFRAME_COLUMNS) + "\n")
            for l, h in enumerate(st.lm_host):
                if h == k:
                    xy = np.asarray(st.lm_px[l], dtype=np.float64)
                elif k in st.lm_obs[l]:
                    xy = np.asarray(st.lm_obs[l][k], dtype=np.float64)
                else:
                    continue
                pw = st.R[h] @ (synth.R_IC @ (np.array([st.lm_px[l][0], st.lm_px[l][1], 1.0]) * st.lm_depth[l]) + synth.T_IC) + st.P[h]
                vals = [st.times[k], l, *pw, 1.0, xy[0], xy[1], fx * xy[0] + cx, fx * xy[1] + cy]
                f.write(",".join(str(v) if isinstance(v, int) else repr(float(v)) for v in vals) + "\n")


def _read_csv(path, wanted, optional=()):
    """Columns by name, extra columns ignored (io::ignore_extra_column of the reference's CSV reader)."""
    with open(path) as f:
        header = [h.strip() for h in f.readline().strip().split(",")]
        missing = [w for w in wanted if w not in header]
        if missing:
            raise ValueError("%s: missing column(s) %s" % (path, ", ".join(missing)))
        cols = [header.index(w) for w in wanted]
        opt = [header.index(w) if w in header else -1 for w in optional]
        data, extra = [], []
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split(",")
            data.append([float(parts[c]) for c in cols])
            extra.append([float(parts[c]) if c >= 0 else np.nan for c in opt])
    return np.array(data, dtype=np.float64).reshape(-1, len(wanted)), np.array(extra, dtype=np.float64).reshape(-1, len(optional))


def cut_imu_intervals(t_imu, meas, times, noise=None):
    """IMU samples (stamps t_imu, rows acc | gyro) cut into per-frame intervals the way System::ProcessBackEnd does
    (System.cpp:363-401): every sample up to the image stamp, then one sample linearly interpolated to the stamp itself, which also
    opens the next interval (Estimator::processIMU: acc_0 / gyr_0).  Returns (raw intervals, their pre-integrations at zero bias)."""
    def at(t):
        j = int(np.clip(np.searchsorted(t_imu, t), 1, len(t_imu) - 1))
        w = (t - t_imu[j - 1]) / (t_imu[j] - t_imu[j - 1])
        w = min(max(w, 0.0), 1.0)
        return (1.0 - w) * meas[j - 1] + w * meas[j]

    ivs, pres = [], []
    for k in range(len(times) - 1):
        ta, tb = times[k], times[k + 1]
        first = at(ta)
        dts, accs, gyrs, cur = [], [], [], ta
        for j in np.nonzero((t_imu > ta) & (t_imu <= tb))[0]:
            if t_imu[j] - cur > 0:
                dts.append(float(t_imu[j] - cur)); accs.append(meas[j, 0:3].copy()); gyrs.append(meas[j, 3:6].copy())
                cur = float(t_imu[j])
        if tb - cur > 1e-12:
            last = at(tb)
            dts.append(float(tb - cur)); accs.append(last[0:3].copy()); gyrs.append(last[3:6].copy())
        ivs.append(dict(acc0=first[0:3].copy(), gyr0=first[3:6].copy(), dt=dts, acc=accs, gyr=gyrs))
        pres.append(synth.preintegrate(first[0:3], first[3:6], np.zeros(3), np.zeros(3), dts, accs, gyrs, **(noise or {})))
    return ivs, pres


def make_tracks(st, rng, landmarks_per_frame, track_len, pixel_noise, ric=None, tic=None):
    """Synthetic vision on a stream's ground-truth poses: landmarks hosted in frame h, seen in h+1 .. h+track_len."""
    ric = synth.R_IC if ric is None else ric
    tic = synth.T_IC if tic is None else tic
    st.lm_host, st.lm_px, st.lm_depth, st.lm_obs = [], [], [], []
    for h in range(st.n_frames - 1):
        for _ in range(landmarks_per_frame):
            px = rng.uniform(-0.5, 0.5, 2)
            depth = rng.uniform(4.0, 10.0)
            pw = st.R[h] @ (ric @ (np.array([px[0], px[1], 1.0]) * depth) + tic) + st.P[h]
            obs = {}
            for j in range(h + 1, min(h + 1 + track_len, st.n_frames)):
                pc = ric.T @ (st.R[j].T @ (pw - st.P[j]) - tic)
                pc[2] = max(pc[2], 0.5)
                obs[j] = pc[0:2] / pc[2] + rng.normal(0.0, pixel_noise, 2)
            st.lm_host.append(h)
            st.lm_px.append(px)
            st.lm_depth.append(depth)
            st.lm_obs.append(obs)
    st.init_noise = rng.normal(size=(len(st.lm_host),))


class RealImuStream:
    """Real inertial data, synthetic vision: the nearest runnable stand-in for the reference's EuRoC runs (VM/test/run_euroc.cpp:26-110
    feeds MH_05_imu0.txt and the images of MH_05_cam0.txt; the images and the OpenCV front-end are not in the tree, the IMU file
    and the stamps are).  `data` = tests/golden/mh05_imu_stretch.npz: the samples are cut at the frame stamps as System::ProcessBackEnd
    does and pre-integrated with the noise parameters of euroc_config.yaml; the "ground truth" is the trajectory those
    pre-integrations define from a gravity-aligned start at rest (P_j = P_i + V_i dt - g dt^2 / 2 + R_i delta_p, ...: the IMU
    factors' own model, integration_base.h:160-186, so the inertial residuals vanish on it), and the landmarks are drawn in front of
    its camera poses with the sequence's extrinsic.  Same attributes as SyntheticStream: StreamDriver runs on it."""

    def __init__(self, data, n_frames=None, landmarks_per_frame=30, track_len=5, seed=0, pixel_noise=1.0 / 460.0):
        rng = np.random.RandomState(seed)
        cam_t = np.asarray(data["cam_t"], dtype=np.float64)
        self.n_frames = int(n_frames or len(cam_t))
        self.times = [float(t) for t in cam_t[:self.n_frames]]
        self.t0, self.frame_dt = self.times[0], (self.times[-1] - self.times[0]) / max(1, self.n_frames - 1)
        self.noise = dict(acc_n=float(data["acc_n"]), gyr_n=float(data["gyr_n"]), acc_w=float(data["acc_w"]), gyr_w=float(data["gyr_w"]))
        self.g_norm = float(data["g_norm"])
        self.ric, self.tic = np.asarray(data["ric"], dtype=np.float64), np.asarray(data["tic"], dtype=np.float64)
        self.ext = np.concatenate([self.tic, synth.rot_to_quat(self.ric)])
        meas = np.concatenate([np.asarray(data["imu_acc"]), np.asarray(data["imu_gyr"])], axis=1)
        self.imu, self.preint = cut_imu_intervals(np.asarray(data["imu_t"], dtype=np.float64), meas, self.times, self.noise)
        # start: at rest, the first accelerometer sample straight up (R_0 a_0 || +z: the specific force of a body at rest is -g)
        a0 = self.imu[0]["acc0"] / np.linalg.norm(self.imu[0]["acc0"])
        z = np.array([0.0, 0.0, 1.0])
        v = np.cross(a0, z)
        c = float(a0 @ z)
        K = synth.skew(v)
        R0 = np.eye(3) + K + K @ K / (1.0 + c)
        g = np.array([0.0, 0.0, self.g_norm])
        self.R, self.P, self.V = np.zeros((self.n_frames, 3, 3)), np.zeros((self.n_frames, 3)), np.zeros((self.n_frames, 3))
        self.Q = np.zeros((self.n_frames, 4))
        self.R[0], self.Q[0] = R0, synth.rot_to_quat(R0)
        for k, pre in enumerate(self.preint):
            dt = pre["sum_dt"]
            self.P[k + 1] = self.P[k] + self.V[k] * dt - 0.5 * g * dt * dt + self.R[k] @ pre["delta_p"]
            self.V[k + 1] = self.V[k] - g * dt + self.R[k] @ pre["delta_v"]
            q = synth.quat_mul(self.Q[k], pre["delta_q"])
            self.Q[k + 1] = q / np.linalg.norm(q)
            self.R[k + 1] = synth.quat_to_rot(self.Q[k + 1])
        self.has_ground_truth = True
        make_tracks(self, rng, landmarks_per_frame, track_len, pixel_noise, self.ric, self.tic)


class SimulatorFileStream:
    """A stream read from the reference simulator's files (see write_simulator_files): the same attributes as
    SyntheticStream, so StreamDriver runs on either.  IMU samples are cut into per-frame intervals the way
    System::ProcessBackEnd does (System.cpp:363-401): every sample up to the image stamp, then one sample linearly
    interpolated to the stamp itself, which also opens the next interval (Estimator::processIMU, acc_0 / gyr_0)."""

    def __init__(self, directory, imu_file="imu_pose.txt", keyframe_dir="keyframe", max_frames=None, seed=0):
        import os
        import re
        imu, extra = _read_csv(os.path.join(directory, imu_file),
                               ("timestamp", "gyro_x", "gyro_y", "gyro_z", "acc_x", "acc_y", "acc_z"),
                               ("q_w", "q_x", "q_y", "q_z", "p_x", "p_y", "p_z", "v_x", "v_y", "v_z"))
        files = []
        for name in os.listdir(os.path.join(directory, keyframe_dir)):
            m = re.fullmatch(r"all_points_(.*)\.txt", name)
            if m:
                files.append((int(m.group(1)), os.path.join(directory, keyframe_dir, name)))
        files.sort()                                        # by sequence id, run_vio_simulation.cpp:141-147
        if max_frames:
            files = files[:max_frames]
        frames = []
        for _, path in files:
            kf, world = _read_csv(path, ("timestamp", "id", "x_normalized", "y_normalized"), ("x", "y", "z"))
            if len(kf):
                frames.append((kf, world))
        self.n_frames = len(frames)
        self.times = [float(kf[0, 0]) for kf, _ in frames]
        self.t0, self.frame_dt = self.times[0], (self.times[-1] - self.times[0]) / max(1, self.n_frames - 1)
        self.ext = np.concatenate([synth.T_IC, synth.rot_to_quat(synth.R_IC)])
        t_imu = imu[:, 0]

        def at(t, cols):                                    # linear interpolation of imu/extra columns at stamp t
            j = int(np.clip(np.searchsorted(t_imu, t), 1, len(t_imu) - 1))
            w = (t - t_imu[j - 1]) / (t_imu[j] - t_imu[j - 1])
            w = min(max(w, 0.0), 1.0)
            return (1.0 - w) * cols[j - 1] + w * cols[j]

        # ground truth at the frame stamps (initial guesses and the ATE reference); velocity by central differences
        # of the position when the file has no v columns
        have_gt = not np.isnan(extra[:, 0:7]).any()
        self.P, self.Q, self.R, self.V = (np.zeros((self.n_frames, 3)), np.zeros((self.n_frames, 4)),
                                          np.zeros((self.n_frames, 3, 3)), np.zeros((self.n_frames, 3)))
        for k, t in enumerate(self.times):
            if not have_gt:
                self.Q[k] = (0, 0, 0, 1)
                self.R[k] = np.eye(3)
                continue
            g = at(t, extra)
            q = np.array([g[1], g[2], g[3], g[0]])
            self.Q[k] = q / np.linalg.norm(q)
            self.R[k] = synth.quat_to_rot(self.Q[k])
            self.P[k] = g[4:7]
            if not np.isnan(extra[:, 7:10]).any():
                self.V[k] = g[7:10]
            else:
                h = 2.0 * np.median(np.diff(t_imu))
                self.V[k] = (at(t + h, extra)[4:7] - at(t - h, extra)[4:7]) / (2.0 * h)
        self.has_ground_truth = bool(have_gt)
        # IMU intervals
        meas = imu[:, [4, 5, 6, 1, 2, 3]]                  # acc, gyro
        self.imu, self.preint = cut_imu_intervals(t_imu, meas, self.times)
        # tracks: a feature id is hosted by the first frame that sees it
        index = {}
        self.lm_host, self.lm_px, self.lm_depth, self.lm_obs, self.lm_id = [], [], [], [], []
        for k, (kf, world) in enumerate(frames):
            for row, pw in zip(kf, world):
                fid = int(row[1])
                if fid not in index:
                    index[fid] = len(self.lm_host)
                    self.lm_host.append(k)
                    self.lm_px.append(row[2:4].copy())
                    depth = np.nan
                    if have_gt and not np.isnan(pw).any():
                        pc = synth.R_IC.T @ (self.R[k].T @ (pw - self.P[k]) - synth.T_IC)
                        depth = float(pc[2])
                    self.lm_depth.append(depth)
                    self.lm_obs.append({})
                    self.lm_id.append(fid)
                else:
                    self.lm_obs[index[fid]][k] = row[2:4].copy()
        self.init_noise = np.random.RandomState(seed).normal(size=(len(self.lm_host),))


class StreamDriver:
    def __init__(self, lib, stream, ctx_kwargs=None, pos_noise=0.02, rot_noise=0.005, depth_noise=0.05, seed=1,
                 triangulate=False, nonkey_every=0):
        """triangulate=True: a landmark's first depth comes from FeatureManager::triangulate (vio_triangulate, on the
        current pose estimates, feature_manager.cpp:203-257) the first time it enters a solve, as in
        Estimator::solveOdometry (estimator.cpp:489-503), instead of from the perturbed ground truth.
        nonkey_every=n > 0: every n-th frame is not a keyframe: when it is the second-newest frame of the window it is
        marginalised (MARGIN_SECOND_NEW) instead of the oldest one."""
        self.lib, self.s = lib, stream
        self.noise = dict(getattr(stream, "noise", None) or {})      # sensor noise of re-integrated intervals (default: synth's)
        self.g_norm = float(getattr(stream, "g_norm", synth.G_NORM))
        kw = dict(ctx_kwargs or {})
        if hasattr(stream, "g_norm"):
            kw.setdefault("gravity", (0.0, 0.0, self.g_norm))
        self.ctx = lib.context(**kw)
        rng = np.random.RandomState(seed)
        st = stream
        self.frames = list(range(NUM_FRAMES))              # global index of the frame in each window slot
        self.intervals = [dict(st.imu[k]) for k in range(WINDOW_SIZE)]     # raw IMU between consecutive window frames
        self.preint = [st.preint[k] for k in range(WINDOW_SIZE)]
        self.next_frame = NUM_FRAMES
        self.nonkey_every = nonkey_every
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
        # tracks (FeaturePerId): landmark -> list of (global frame, normalised point), the first one is the host
        self.tracks = {}
        for f in self.frames:
            self.add_frame_observations(f)
        self.depth = {}                                    # landmark -> estimated depth in its current host frame
        self.init_depth = np.array(st.lm_depth) * (1.0 + depth_noise * st.init_noise)
        self.triangulate = triangulate
        self.prior = None
        self.trajectory = []        # (stamp, pose[7]) of the newest frame after every solve
        self.reports = []
        self.flags = []             # marginalisation flag of every step
        self.n_triangulated = 0

    # ---- feature bookkeeping (FeatureManager) ---------------------------------------------------------
    def add_frame_observations(self, f):
        st = self.s
        for l, h in enumerate(st.lm_host):
            if h == f:
                self.tracks[l] = [(f, np.asarray(st.lm_px[l], dtype=np.float64))]
            elif l in self.tracks and f in st.lm_obs[l] and self.tracks[l][-1][0] == self.prev_of(f):
                self.tracks[l].append((f, np.asarray(st.lm_obs[l][f], dtype=np.float64)))

    def prev_of(self, f):
        """The window frame in front of global frame f (tracks are consecutive in window frames)."""
        fr = self.frames if f in self.frames else self.frames + [f]
        i = fr.index(f)
        return fr[i - 1] if i > 0 else None

    def usable(self):
        """(landmark, window index of its host) of the tracks the optimiser takes: used_num >= 2 and
        start_frame < WINDOW_SIZE - 2 (estimator.cpp:979-981)."""
        out = []
        for l, tr in self.tracks.items():
            start = self.frames.index(tr[0][0])
            if len(tr) >= 2 and start < WINDOW_SIZE - 2:
                out.append((l, start))
        return out

    def ensure_depths(self):
        """First depth of the tracks that have none: triangulation on the current poses, or perturbed ground truth."""
        todo = [(l, start) for l, start in self.usable() if l not in self.depth]
        if not todo:
            return
        if not self.triangulate:
            for l, _ in todo:
                self.depth[l] = float(self.init_depth[l])
            return
        sf, off, pts = [], [0], []
        for l, start in todo:
            sf.append(start); pts.extend(p for _, p in self.tracks[l]); off.append(off[-1] + len(self.tracks[l]))
        d = self.ctx.triangulate(np.array(sf, dtype=np.int32), np.array(off, dtype=np.int64), np.array(pts).reshape(-1, 2),
                                 self.poses, self.ext, -np.ones(len(todo)))
        for (l, _), v in zip(todo, d):
            self.depth[l] = float(v)
        self.n_triangulated += len(todo)

    def window_arrays(self):
        ids, lm, host, target, pi, pj, invd = [], [], [], [], [], [], []
        for l, start in self.usable():
            k = len(ids)
            ids.append(l)
            invd.append(1.0 / self.depth[l])
            tr = self.tracks[l]
            for j in range(1, len(tr)):
                lm.append(k); host.append(start); target.append(start + j); pi.append(tr[0][1]); pj.append(tr[j][1])
        w = synth.Window(poses=self.poses.copy(), speed_bias=self.sb.copy(), ext=self.ext.copy(),
                         inv_depth=np.array(invd), lm=np.array(lm, dtype=np.int32),
                         host=np.array(host, dtype=np.int32), target=np.array(target, dtype=np.int32),
                         pts_i=np.array(pi).reshape(-1, 2), pts_j=np.array(pj).reshape(-1, 2),
                         preint=list(self.preint), prior=self.prior,
                         n_landmarks=len(ids), n_observations=len(lm))
        return w, ids

    @property
    def inv_depth(self):
        """Inverse depths by landmark id (NaN where the landmark never got one) — what the tests compare."""
        out = np.full(len(self.s.lm_host), np.nan)
        for l, d in self.depth.items():
            out[l] = 1.0 / d
        return out

    @property
    def have_depth(self):
        out = np.zeros(len(self.s.lm_host), dtype=bool)
        out[list(self.depth.keys())] = True
        return out

    # ---- one frame ------------------------------------------------------------------------------------
    def step(self):
        """Solve the current window, marginalise, slide, take in the next frame.  Returns False at the end."""
        st = self.s
        self.ensure_depths()
        w, ids = self.window_arrays()
        self.ctx.load(w)
        rep = self.ctx.solve(10)
        poses, sb, _ = self.ctx.get_window()
        invd = self.ctx.get_landmarks()
        if self.prior is not None:      # estimator.cpp:1040-1049: b/err prior come back updated, H/Jt stay
            b, e = self.ctx.get_prior()
            self.prior = dict(self.prior, b=b[:156].copy(), err=e.copy())
        self.poses, self.sb = anchor_gauge(w.poses, poses, sb)
        for l, v in zip(ids, invd):
            self.depth[l] = 1.0 / v
        newest = self.frames[WINDOW_SIZE]
        self.trajectory.append((st.times[newest], self.poses[WINDOW_SIZE].copy()))
        self.reports.append(rep)
        second_new = self.frames[WINDOW_SIZE - 1]
        margin_old = not (self.nonkey_every and second_new % self.nonkey_every == self.nonkey_every - 1)
        self.flags.append(MARG_OLD if margin_old else MARG_SECOND_NEW)
        # backendOptimization marginalises on the re-anchored states (estimator.cpp:1086-1102)
        w2, _ = self.window_arrays()
        self.ctx.load(w2)
        self.prior = self.ctx.marginalize(MARG_OLD if margin_old else MARG_SECOND_NEW)
        if self.next_frame >= st.n_frames:
            return False
        if margin_old:
            self.slide_window_old()
        else:
            self.slide_window_new()
        self.take_next_frame()
        return True

    def slide_window_old(self):
        """slideWindowOld (estimator.cpp:1144-1199) + removeBackShiftDepth (feature_manager.cpp:276-312)."""
        gone = self.frames[0]
        R0 = synth.quat_to_rot(self.poses[0, 3:7]); R1 = synth.quat_to_rot(self.poses[1, 3:7])
        ric, tic = synth.quat_to_rot(self.ext[3:7]), self.ext[0:3]
        mR, mP, nR, nP = R0 @ ric, self.poses[0, 0:3] + R0 @ tic, R1 @ ric, self.poses[1, 0:3] + R1 @ tic
        for l in list(self.tracks.keys()):
            tr = self.tracks[l]
            if tr[0][0] != gone:
                continue
            uv = np.array([tr[0][1][0], tr[0][1][1], 1.0])
            del tr[0]
            if len(tr) < 2:
                del self.tracks[l]
                self.depth.pop(l, None)
                continue
            if l in self.depth:
                pj = nR.T @ (mR @ (uv * self.depth[l]) + mP - nP)
                self.depth[l] = float(pj[2]) if pj[2] > 0 else 5.0       # INIT_DEPTH
        self.frames.pop(0)
        self.intervals.pop(0)
        self.preint.pop(0)
        self.poses[:-1], self.sb[:-1] = self.poses[1:].copy(), self.sb[1:].copy()

    def slide_window_new(self):
        """slideWindowNew (estimator.cpp:1201-1206, the frame_count - 1 branch of :1163-1186) + removeFront."""
        gone = self.frames[WINDOW_SIZE - 1]
        for l in list(self.tracks.keys()):
            tr = [o for o in self.tracks[l] if o[0] != gone]
            if not tr:
                del self.tracks[l]
                self.depth.pop(l, None)
            else:
                self.tracks[l] = tr
        # the IMU samples of the newest interval are appended to the one before it
        a, b = self.intervals[WINDOW_SIZE - 2], self.intervals[WINDOW_SIZE - 1]
        merged = dict(acc0=a["acc0"], gyr0=a["gyr0"], dt=a["dt"] + b["dt"], acc=a["acc"] + b["acc"], gyr=a["gyr"] + b["gyr"])
        self.intervals[WINDOW_SIZE - 2:] = [merged]
        self.preint[WINDOW_SIZE - 2:] = [synth.preintegrate(merged["acc0"], merged["gyr0"], np.zeros(3), np.zeros(3),
                                                           merged["dt"], merged["acc"], merged["gyr"], **self.noise)]
        self.frames.pop(WINDOW_SIZE - 1)
        self.poses[WINDOW_SIZE - 1], self.sb[WINDOW_SIZE - 1] = self.poses[WINDOW_SIZE].copy(), self.sb[WINDOW_SIZE].copy()

    def take_next_frame(self):
        """The next image: processIMU's propagation of the newest state, then the frame's observations."""
        st = self.s
        f = self.next_frame
        self.next_frame += 1
        last = self.frames[-1]
        # raw samples from `last` to f (consecutive global frames unless frames were skipped: they never are here)
        iv = dict(st.imu[last])
        pre = st.preint[last]
        self.frames.append(f)
        self.intervals.append(iv)
        self.preint.append(pre)
        dt = pre["sum_dt"]
        i = WINDOW_SIZE - 1
        Ri = synth.quat_to_rot(self.poses[i, 3:7])
        g = np.array([0.0, 0.0, self.g_norm])
        Pi, Vi = self.poses[i, 0:3], self.sb[i, 0:3]
        self.poses[WINDOW_SIZE, 0:3] = Pi + Vi * dt - 0.5 * g * dt * dt + Ri @ pre["delta_p"]
        self.poses[WINDOW_SIZE, 3:7] = synth.quat_mul(self.poses[i, 3:7], pre["delta_q"])
        self.sb[WINDOW_SIZE, 0:3] = Vi - g * dt + Ri @ pre["delta_v"]
        self.sb[WINDOW_SIZE, 3:9] = self.sb[i, 3:9]
        self.add_frame_observations(f)

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


def read_tum(path):
    """TUM trajectory file: `stamp px py pz qx qy qz qw` per line (System.cpp:438), '#' comments allowed."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                rows.append([float(v) for v in line.replace(",", " ").split()[:8]])
    return np.array(rows, dtype=np.float64).reshape(-1, 8)


def associate(est, gt, max_diff=0.01):
    """Pair every estimated stamp with the closest ground-truth stamp not further than max_diff (one to one, in time
    order) — the association of `evo_ape tum` (README.md:169 of the reference's assignment 17)."""
    ti, tj = est[:, 0], gt[:, 0]
    j = np.clip(np.searchsorted(tj, ti), 1, len(tj) - 1)
    j = np.where(np.abs(tj[j - 1] - ti) <= np.abs(tj[j] - ti), j - 1, j)
    ok = np.abs(tj[j] - ti) <= max_diff
    # one to one: keep the first estimate claiming a ground-truth sample
    _, first = np.unique(np.where(ok, j, -1), return_index=True)
    keep = np.zeros(len(ti), dtype=bool)
    keep[first] = True
    keep &= ok
    return np.nonzero(keep)[0], j[keep]


def umeyama_se3(x, y):
    """Rotation R and translation t minimising sum |R x_k + t - y_k|^2 (Umeyama 1991, no scale): `evo_ape -a`."""
    mx, my = x.mean(axis=0), y.mean(axis=0)
    S = (y - my).T @ (x - mx) / len(x)
    U, _, Vt = np.linalg.svd(S)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1.0
    R = U @ D @ Vt
    return R, my - R @ mx


def ape_stats(est, gt, align=True, max_diff=0.01):
    """Translation APE statistics as `evo_ape tum gt est -va` prints them (rmse, mean, median, std, min, max, sse)."""
    ie, ig = associate(est, gt, max_diff)
    x, y = est[ie, 1:4], gt[ig, 1:4]
    if align:
        R, t = umeyama_se3(x, y)
        x = x @ R.T + t
    e = np.linalg.norm(x - y, axis=1)
    return {"rmse": float(np.sqrt((e * e).mean())), "mean": float(e.mean()), "median": float(np.median(e)),
            "std": float(e.std()), "min": float(e.min()), "max": float(e.max()), "sse": float((e * e).sum()),
            "pairs": int(len(e))}


def write_tum(path, traj):
    with open(path, "w") as f:
        for row in traj:
            f.write(" ".join("%.9f" % v for v in row) + "\n")
