#!/usr/bin/env python3
"""Generates tests/golden/feature_manager.npz and tests/golden/gauge.npz from the COMPILED REFERENCE
(oracle/_ref/libvio_ref.so: VM/src/feature_manager.cpp + VM/include/utility/utility.h + vendored Eigen 3.3.4, driven by
oracle/ref_feature_manager.cpp).

Run only in the container that mounts /root/reference:   python tests/golden/make_golden_feature_manager.py
The files hold inputs and the reference's outputs for them; no reference source travels with them.

feature_manager.npz
  tri<k>_*        FeatureManager::triangulate (feature_manager.cpp:203-257): tracks in CSR form, poses, ext, depths in/out
  sc_<name>_*     an operation sequence (tests/fm_util.py: scenario_ops) on a track list: tracks in, ops, tracks out,
                  getFeatureCount(), getDepthVector()
  par_*           addFeatureCheckParallax (:55-115) over 14 images: keyframe decisions, last_track_num, final tracks
gauge.npz
  r2ypr / ypr2r / quat_to_rot / rot_to_quat / normalize_angle vectors (utility.h:68-139, Eigen's conversions), and
  d2v_*: whole double2vector / vector2double cases (estimator.cpp:505-600) assembled from those reference primitives
  in the reference's order of operations.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
import fm_util as fu  # noqa: E402
from test_triangulate import make_tracks  # noqa: E402

vio = load_package()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
dll = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so"))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# =========================================== FeatureManager ===========================================================
fm = {}
# ---- triangulate
tri_cases = [dict(n=400, seed=3, noise=0.0), dict(n=400, seed=3, noise=1.0 / 460.0), dict(n=1500, seed=21, noise=2.0 / 460.0),
             dict(n=60, seed=8, noise=30.0 / 460.0)]          # the last one: gross noise -> depths < 0.1 -> INIT_DEPTH
for k, c in enumerate(tri_cases):
    sf, off, pts, poses, ext, d0, _ = make_tracks(vio, c["n"], seed=c["seed"], noise=c["noise"])
    init_depth = 5.0 if k != 3 else 7.5
    r = fu.RefFeatureManager(dll, init_depth=init_depth)
    r.add_tracks([dict(id=i, start=int(sf[i]), depth=float(d0[i]), pts=pts[off[i]:off[i + 1]]) for i in range(len(sf))])
    r.apply((1,), poses, ext)
    out = np.array([t["depth"] for t in r.tracks()])
    p = "tri%d_" % k
    fm.update({p + "start": sf, p + "off": off, p + "pts": pts, p + "poses": poses, p + "ext": ext, p + "depth_in": d0,
               p + "depth_out": out, p + "init_depth": np.float64(init_depth)})
    done = (np.diff(off) >= 2) & (sf < 8) & (d0 <= 0)
    print("tri%d: %d tracks, %d triangulated, %d fell back to INIT_DEPTH" % (k, len(sf), done.sum(), (out[done] == init_depth).sum()))
fm["n_tri"] = np.int32(len(tri_cases))

# ---- operation sequences
for name in fu.SCENARIOS:
    sf, off, pts, poses, ext, d0, _ = make_tracks(vio, 160, seed=7, noise=1.0 / 460.0,
                                                  have_depth_frac=0.5 if name == "frame_chain" else 1.0)
    tracks = [dict(id=100 + i, start=int(sf[i]), depth=float(d0[i]), pts=pts[off[i]:off[i + 1]]) for i in range(len(sf))]
    r = fu.RefFeatureManager(dll)
    r.add_tracks(tracks)
    ops = fu.scenario_ops(vio, name, tracks, poses, ext)
    for op in ops:
        r.apply(op, poses, ext)
    p = "sc_%s_" % name
    fm.update(fu.tracks_to_arrays(tracks, p + "in_"))
    fm.update(fu.ops_to_arrays(ops, p + "op_"))
    fm.update(fu.tracks_to_arrays(r.tracks(), p + "out_"))
    fm.update({p + "poses": poses, p + "ext": ext, p + "count": np.int32(r.count()), p + "depvec": r.depth_vector()})
    print("%-16s %d tracks -> %d, %d usable" % (name, len(tracks), len(r.tracks()), r.count()))

# ---- addFeatureCheckParallax: 14 images; ids persist for a few frames, some move a lot (keyframe), some barely
rng = np.random.RandomState(5)
r = fu.RefFeatureManager(dll)
ids_all, pts_all, img_off, frame_counts, keyflags, last_tracks = [], [], [0], [], [], []
alive = {}
next_id = 0
for f in range(14):
    frame_count = min(f, fu.WINDOW_SIZE)
    motion = 0.08 if f % 3 else 0.004                       # small motion -> parallax below MIN_PARALLAX -> not a keyframe
    for i in list(alive):
        alive[i] = alive[i] + rng.normal(0.0, motion, 2)
        if rng.uniform() < 0.15:
            del alive[i]
    while len(alive) < 40:
        alive[next_id] = rng.uniform(-0.5, 0.5, 2)
        next_id += 1
    ids = np.array(sorted(alive), dtype=np.int32)
    pts = np.stack([alive[i] for i in ids])
    key, last = r.add_image(frame_count, ids, pts)
    ids_all.append(ids); pts_all.append(pts); img_off.append(img_off[-1] + len(ids))
    frame_counts.append(frame_count); keyflags.append(key); last_tracks.append(last)
    if frame_count == fu.WINDOW_SIZE:                        # slide as Estimator::slideWindow does before the next image
        if key:
            r.apply((5,))
        else:
            r.apply((6, fu.WINDOW_SIZE))
fm.update({"par_ids": np.concatenate(ids_all), "par_pts": np.concatenate(pts_all), "par_off": np.array(img_off, dtype=np.int64),
           "par_frame_count": np.array(frame_counts, dtype=np.int32), "par_key": np.array(keyflags, dtype=np.int32),
           "par_last_track_num": np.array(last_tracks, dtype=np.int32)})
fm.update(fu.tracks_to_arrays(r.tracks(), "par_out_"))
print("parallax: keyframe decisions", "".join("K" if k else "-" for k in keyflags), " last_track_num", last_tracks)
save("feature_manager", **fm)

# =========================================== gauge helpers ============================================================
for fn in ("vior_r2ypr", "vior_ypr2r", "vior_quat_to_rot", "vior_rot_to_quat"):
    getattr(dll, fn).restype = None
dll.vior_normalize_angle.restype = C.c_double


def ref_r2ypr(R):
    out = np.zeros(3)
    dll.vior_r2ypr(dp(np.ascontiguousarray(R, dtype=np.float64)), dp(out))
    return out


def ref_ypr2r(ypr):
    out = np.zeros((3, 3))
    dll.vior_ypr2r(dp(np.ascontiguousarray(ypr, dtype=np.float64)), dp(out))
    return out


def ref_q2r(q, normalize):
    out = np.zeros((3, 3))
    dll.vior_quat_to_rot(dp(np.ascontiguousarray(q, dtype=np.float64)), C.c_int(int(normalize)), dp(out))
    return out


def ref_r2q(R):
    out = np.zeros(4)
    dll.vior_rot_to_quat(dp(np.ascontiguousarray(R, dtype=np.float64)), dp(out))
    return out


rng = np.random.RandomState(9)
g = {}
ypr_in = np.concatenate([rng.uniform(-180, 180, (60, 3)) * [1, 0.5, 1],
                         # around the singularity of double2vector's test |pitch| within 1 degree of 90
                         np.stack([rng.uniform(-180, 180, 12), np.array([89.5, -89.5, 90.0, -90.0, 88.9, -88.9, 89.0, -89.0, 91.0, 89.999, 45.0, 0.0]),
                                   rng.uniform(-180, 180, 12)], axis=1)])
R_from_ypr = np.stack([ref_ypr2r(v) for v in ypr_in])
g["ypr_in"], g["ypr2r_out"] = ypr_in, R_from_ypr
g["r2ypr_out"] = np.stack([ref_r2ypr(R) for R in R_from_ypr])
q_in = rng.normal(size=(80, 4))
q_in[:40] /= np.linalg.norm(q_in[:40], axis=1, keepdims=True)      # unit and non-unit quaternions
g["q_in"] = q_in
g["q2r_plain"] = np.stack([ref_q2r(q, 0) for q in q_in])
g["q2r_normalized"] = np.stack([ref_q2r(q, 1) for q in q_in])
# rotation -> quaternion: every branch of Eigen's selection (trace > 0, largest diagonal element 0 / 1 / 2)
R_in = np.concatenate([g["q2r_normalized"], np.stack([ref_ypr2r(v) for v in ([179.0, 0, 0], [0, 179.0, 0], [0, 0, 179.0], [120, 10, 170], [0, 0, 0])])])
g["r_in"] = R_in
g["r2q_out"] = np.stack([ref_r2q(R) for R in R_in])
ang = np.concatenate([rng.uniform(-1000, 1000, 40), [0.0, 180.0, -180.0, 360.0, 540.0, -540.0]])
g["angle_in"], g["angle_out"] = ang, np.array([dll.vior_normalize_angle(C.c_double(a)) for a in ang])

# double2vector (estimator.cpp:551-600) out of the reference primitives, in its order of operations.  Matrix products
# of 3x3 operands are the only arithmetic done here (numpy); everything else is a reference call.
NF = 11
cases = []
for c in range(8):
    near_singular = c >= 6
    ypr0 = np.array([rng.uniform(-170, 170), 89.6 if near_singular else rng.uniform(-40, 40), rng.uniform(-30, 30)])
    Rs0 = ref_ypr2r(ypr0)
    Ps0 = rng.uniform(-5, 5, 3)
    poses = np.zeros((NF, 7))
    sb = rng.normal(size=(NF, 9))
    for i in range(NF):
        ypr = ypr0 + rng.normal(0, 3.0, 3) + [2.0 * i, 0, 0]
        if c == 7:
            ypr[1] = 20.0                                      # the solved frames are far from the singularity; only Rs[0] is near
        q = ref_r2q(ref_ypr2r(ypr))
        poses[i, 3:7] = q * (1.0 + 1e-9 * rng.normal())        # the solver does not renormalise (vertex_pose.cc:12)
        poses[i, 0:3] = Ps0 + rng.normal(0, 1.0, 3) + [0.3 * i, 0, 0]
    origin_R0 = ref_r2ypr(Rs0)
    R00 = ref_q2r(poses[0, 3:7], 0)
    origin_R00 = ref_r2ypr(R00)
    y_diff = origin_R0[0] - origin_R00[0]
    rot_diff = ref_ypr2r(np.array([y_diff, 0.0, 0.0]))
    singular = abs(abs(origin_R0[1]) - 90) < 1.0 or abs(abs(origin_R00[1]) - 90) < 1.0
    if singular:
        rot_diff = Rs0 @ R00.T
    Rs = np.stack([rot_diff @ ref_q2r(poses[i, 3:7], 1) for i in range(NF)])
    Ps = np.stack([rot_diff @ (poses[i, 0:3] - poses[0, 0:3]) + Ps0 for i in range(NF)])
    Vs = np.stack([rot_diff @ sb[i, 0:3] for i in range(NF)])
    # vector2double of the result (:505-528)
    para_pose = np.zeros((NF, 7))
    for i in range(NF):
        para_pose[i, 0:3] = Ps[i]
        para_pose[i, 3:7] = ref_r2q(Rs[i])
    cases.append(dict(Rs0=Rs0, Ps0=Ps0, para_Pose=poses, para_SpeedBias=sb, Rs=Rs, Ps=Ps, Vs=Vs, singular=np.int32(singular),
                      para_Pose_again=para_pose))
    print("d2v case %d: y_diff %.3f deg, singular branch %d" % (c, y_diff, singular))
for key in cases[0]:
    g["d2v_" + key] = np.stack([c[key] for c in cases])
save("gauge", **g)
