"""The C++ FeatureManager mirror (visual-inertial-odometry_amd/host/feature_manager.{h,cpp}): depth
vector in/out, failure removal, the three window-shift rules and triangulation.  The program is compiled with g++
against the C ABI; its results are compared with a plain-Python restatement of the reference's rules
(VM/src/feature_manager.cpp:37-52,141-200,276-350) written here, and for triangulate with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_triangulate import make_tracks

HOST_DIR = os.path.join(ROOT, "visual-inertial-odometry_amd", "host")
CSRC = os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc")
WINDOW_SIZE, INIT_DEPTH = 10, 5.0


def build(tmp_path):
    exe = str(tmp_path / "feature_manager_main")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "feature_manager_main.cpp"),
                           os.path.join(HOST_DIR, "feature_manager.cpp"), "-L" + CSRC, "-lvio_hip", "-Wl,-rpath," + CSRC,
                           "-o", exe])
    return exe


def write_input(path, tracks, poses, ext, ops):
    with open(path, "wb") as f:
        f.write(struct.pack("<q", len(tracks)))
        for t in tracks:
            f.write(struct.pack("<iiid", t["id"], t["start"], len(t["pts"]), t["depth"]))
            f.write(np.asarray(t["pts"], dtype=np.float64).tobytes())
        f.write(np.asarray(poses, dtype=np.float64).tobytes())
        f.write(np.asarray(ext, dtype=np.float64).tobytes())
        f.write(struct.pack("<i", len(ops)))
        for op in ops:
            f.write(struct.pack("<i", op[0]))
            if op[0] in (2, 7):
                f.write(struct.pack("<q", len(op[1])))
                f.write(np.asarray(op[1], dtype=np.float64).tobytes())
            elif op[0] == 4:
                f.write(np.concatenate([np.ravel(a) for a in op[1:]]).astype(np.float64).tobytes())
            elif op[0] == 6:
                f.write(struct.pack("<i", op[1]))


def read_output(path):
    b = open(path, "rb").read()
    o = 0
    cnt, = struct.unpack_from("<i", b, o); o += 4
    nd, = struct.unpack_from("<q", b, o); o += 8
    dep = np.frombuffer(b, dtype=np.float64, count=nd, offset=o).copy(); o += 8 * nd
    n, = struct.unpack_from("<q", b, o); o += 8
    tracks = []
    for _ in range(n):
        fid, start, k = struct.unpack_from("<iii", b, o); o += 12
        depth, = struct.unpack_from("<d", b, o); o += 8
        flag, = struct.unpack_from("<i", b, o); o += 4
        pts = np.frombuffer(b, dtype=np.float64, count=2 * k, offset=o).reshape(k, 2).copy(); o += 16 * k
        tracks.append(dict(id=fid, start=start, depth=depth, flag=flag, pts=pts))
    return cnt, dep, tracks


# ---- the reference's rules, restated in Python -----------------------------------------------------------
def usable(t):
    return len(t["pts"]) >= 2 and t["start"] < WINDOW_SIZE - 2


def py_apply(tracks, ops):
    tracks = [dict(t, pts=np.array(t["pts"], dtype=np.float64).reshape(-1, 2), flag=0) for t in tracks]
    for op in ops:
        if op[0] in (2, 7):
            x = iter(op[1])
            for t in tracks:
                if usable(t):
                    t["depth"] = 1.0 / next(x)
                    if op[0] == 2:
                        t["flag"] = 2 if t["depth"] < 0 else 1
        elif op[0] == 3:
            tracks = [t for t in tracks if t["flag"] != 2]
        elif op[0] == 4:
            mR, mP, nR, nP = (np.asarray(a, dtype=np.float64) for a in op[1:])
            out = []
            for t in tracks:
                if t["start"] != 0:
                    t["start"] -= 1
                    out.append(t)
                    continue
                uv = np.array([t["pts"][0, 0], t["pts"][0, 1], 1.0])
                t["pts"] = t["pts"][1:]
                if len(t["pts"]) < 2:
                    continue
                pj = nR.reshape(3, 3).T @ (mR.reshape(3, 3) @ (uv * t["depth"]) + mP - nP)
                t["depth"] = pj[2] if pj[2] > 0 else INIT_DEPTH
                out.append(t)
            tracks = out
        elif op[0] == 5:
            out = []
            for t in tracks:
                if t["start"] != 0:
                    t["start"] -= 1
                    out.append(t)
                else:
                    t["pts"] = t["pts"][1:]
                    if len(t["pts"]):
                        out.append(t)
            tracks = out
        elif op[0] == 6:
            fc, out = op[1], []
            for t in tracks:
                if t["start"] == fc:
                    t["start"] -= 1
                    out.append(t)
                    continue
                j = WINDOW_SIZE - 1 - t["start"]
                if t["start"] + len(t["pts"]) - 1 < fc - 1:
                    out.append(t)
                    continue
                t["pts"] = np.delete(t["pts"], j, axis=0)
                if len(t["pts"]):
                    out.append(t)
            tracks = out
    return tracks


def scenario(vio, n, seed):
    sf, off, pts, poses, ext, d0, _ = make_tracks(vio, n, seed=seed, noise=1.0 / 460.0, have_depth_frac=1.0)
    tracks = [dict(id=100 + i, start=int(sf[i]), depth=float(d0[i]), pts=pts[off[i]:off[i + 1]]) for i in range(n)]
    return tracks, poses, ext


def compare(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["id"] == w["id"] and g["start"] == w["start"] and g["flag"] == w["flag"]
        np.testing.assert_array_equal(g["pts"], w["pts"])
        np.testing.assert_allclose(g["depth"], w["depth"], rtol=1e-14)


@pytest.mark.parametrize("ops_kind", ["depth_vector", "shift_old", "shift_old_init", "shift_new"])
def test_feature_manager_host_rules(vio, tmp_path, ops_kind):
    """No GPU needed: every operation except triangulate is host arithmetic."""
    exe = build(tmp_path)
    tracks, poses, ext = scenario(vio, 120, seed=7)
    rng = np.random.RandomState(1)
    n_use = sum(usable(t) for t in tracks)
    R = [vio.synth.quat_to_rot(poses[k, 3:7]) for k in range(11)]
    ric, tic = vio.synth.quat_to_rot(ext[3:7]), ext[0:3]
    if ops_kind == "depth_vector":
        x = rng.uniform(0.05, 0.5, n_use)
        x[::7] *= -1.0                                   # negative inverse depths: solve_flag 2, removed
        ops = [(2, x), (3,)]
    elif ops_kind == "shift_old":                        # slideWindowOld (estimator.cpp:1187-1199) with depth shift
        ops = [(4, R[0] @ ric, poses[0, 0:3] + R[0] @ tic, R[1] @ ric, poses[1, 0:3] + R[1] @ tic)]
    elif ops_kind == "shift_old_init":                   # before initialisation: removeBack
        ops = [(7, rng.uniform(0.1, 0.4, n_use)), (5,)]
    else:
        ops = [(6, WINDOW_SIZE)]                         # slideWindowNew
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    write_input(inp, tracks, poses, ext, ops)
    subprocess.check_call([exe, str(inp), str(out)])
    cnt, dep, got = read_output(out)
    want = py_apply(tracks, ops)
    compare(got, want)
    assert cnt == sum(usable(t) for t in want)
    np.testing.assert_allclose(dep, [1.0 / t["depth"] for t in want if usable(t)], rtol=1e-14)
    assert len(want) < len(tracks) or ops_kind == "shift_new"     # the scenarios do drop tracks


@pytest.mark.gpu
def test_feature_manager_triangulate_on_gpu(vio, oracle_lib, tmp_path):
    exe = build(tmp_path)
    sf, off, pts, poses, ext, d0, _ = make_tracks(vio, 300, seed=19, noise=1.0 / 460.0)
    tracks = [dict(id=i, start=int(sf[i]), depth=float(d0[i]), pts=pts[off[i]:off[i + 1]]) for i in range(len(sf))]
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    write_input(inp, tracks, poses, ext, [(1,)])
    subprocess.check_call([exe, str(inp), str(out)])
    _, _, got = read_output(out)
    want = oracle_lib.context().triangulate(sf, off, pts, poses, ext, d0)
    np.testing.assert_allclose([t["depth"] for t in got], want, rtol=1e-7)
