"""Helpers for the FeatureManager / gauge tests: the ctypes face of the compiled reference's FeatureManager
(oracle/_ref/libvio_ref.so: oracle/ref_feature_manager.cpp), track (de)serialisation for the golden file, scenarios.

Operation codes (the same the C++ test driver tests/cpp/feature_manager_main.cpp reads):
  1 triangulate   2 setDepth(x)   3 removeFailures   4 removeBackShiftDepth(margR, margP, newR, newP)   5 removeBack
  6 removeFront(frame_count)   7 clearDepth(x)
"""
import ctypes as C

import numpy as np

WINDOW_SIZE = 10


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class RefFeatureManager:
    """The reference's FeatureManager, compiled from VM/src/feature_manager.cpp (only where /root/reference exists)."""

    def __init__(self, dll, init_depth=5.0, min_parallax=10.0 / 460.0):
        self.d = dll
        for name, res in (("vior_fm_create", C.c_void_p), ("vior_fm_feature_count", C.c_int), ("vior_fm_depth_vector", C.c_int),
                          ("vior_fm_num_tracks", C.c_int), ("vior_fm_get_track", C.c_int), ("vior_fm_last_track_num", C.c_int),
                          ("vior_fm_add_feature_check_parallax", C.c_int)):
            getattr(dll, name).restype = res
        for name in ("vior_fm_destroy", "vior_fm_add_track", "vior_fm_set_depth", "vior_fm_clear_depth", "vior_fm_remove_failures",
                     "vior_fm_triangulate", "vior_fm_remove_back_shift_depth", "vior_fm_remove_back", "vior_fm_remove_front",
                     "vior_fm_config"):
            getattr(dll, name).restype = None
        dll.vior_fm_config(C.c_double(init_depth), C.c_double(min_parallax))
        self.h = C.c_void_p(dll.vior_fm_create())

    def __del__(self):
        if getattr(self, "h", None):
            self.d.vior_fm_destroy(self.h)
            self.h = None

    def add_tracks(self, tracks):
        for t in tracks:
            p = np.ascontiguousarray(t["pts"], dtype=np.float64).reshape(-1, 2)
            self.d.vior_fm_add_track(self.h, C.c_int(t["id"]), C.c_int(t["start"]), C.c_int(len(p)), _dp(p),
                                     C.c_double(t["depth"]), C.c_int(t.get("flag", 0)))

    def add_image(self, frame_count, ids, pts):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
        key = self.d.vior_fm_add_feature_check_parallax(self.h, C.c_int(frame_count), C.c_int(ids.size),
                                                        ids.ctypes.data_as(C.POINTER(C.c_int)), _dp(pts))
        return bool(key), int(self.d.vior_fm_last_track_num(self.h))

    def count(self):
        return int(self.d.vior_fm_feature_count(self.h))

    def depth_vector(self):
        out = np.zeros(max(self.count(), 1))
        n = self.d.vior_fm_depth_vector(self.h, _dp(out))
        return out[:n].copy()

    def apply(self, op, poses=None, ext=None):
        code = op[0]
        if code == 1:
            self.d.vior_fm_triangulate(self.h, _dp(np.ascontiguousarray(poses, dtype=np.float64)),
                                       _dp(np.ascontiguousarray(ext, dtype=np.float64)))
        elif code in (2, 7):
            x = np.ascontiguousarray(op[1], dtype=np.float64)
            (self.d.vior_fm_set_depth if code == 2 else self.d.vior_fm_clear_depth)(self.h, C.c_int(x.size), _dp(x))
        elif code == 3:
            self.d.vior_fm_remove_failures(self.h)
        elif code == 4:
            a = [np.ascontiguousarray(np.ravel(v), dtype=np.float64) for v in op[1:]]
            self.d.vior_fm_remove_back_shift_depth(self.h, _dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]))
        elif code == 5:
            self.d.vior_fm_remove_back(self.h)
        elif code == 6:
            self.d.vior_fm_remove_front(self.h, C.c_int(op[1]))
        else:
            raise ValueError(code)

    def tracks(self):
        out = []
        for i in range(int(self.d.vior_fm_num_tracks(self.h))):
            fid, start, flag, dep = C.c_int(), C.c_int(), C.c_int(), C.c_double()
            n = self.d.vior_fm_get_track(self.h, C.c_int(i), C.byref(fid), C.byref(start), C.byref(dep), C.byref(flag), None, C.c_int(0))
            pts = np.zeros((n, 2))
            self.d.vior_fm_get_track(self.h, C.c_int(i), C.byref(fid), C.byref(start), C.byref(dep), C.byref(flag), _dp(pts), C.c_int(n))
            out.append(dict(id=fid.value, start=start.value, depth=dep.value, flag=flag.value, pts=pts))
        return out


def usable(t):
    return len(t["pts"]) >= 2 and t["start"] < WINDOW_SIZE - 2


def tracks_to_arrays(tracks, prefix):
    off = np.concatenate([[0], np.cumsum([len(t["pts"]) for t in tracks])]).astype(np.int64)
    pts = np.concatenate([np.asarray(t["pts"], dtype=np.float64).reshape(-1, 2) for t in tracks]) if tracks else np.zeros((0, 2))
    return {prefix + "id": np.array([t["id"] for t in tracks], dtype=np.int32),
            prefix + "start": np.array([t["start"] for t in tracks], dtype=np.int32),
            prefix + "depth": np.array([t["depth"] for t in tracks], dtype=np.float64),
            prefix + "flag": np.array([t.get("flag", 0) for t in tracks], dtype=np.int32),
            prefix + "off": off, prefix + "pts": pts}


def arrays_to_tracks(z, prefix):
    off = z[prefix + "off"]
    return [dict(id=int(z[prefix + "id"][i]), start=int(z[prefix + "start"][i]), depth=float(z[prefix + "depth"][i]),
                 flag=int(z[prefix + "flag"][i]), pts=z[prefix + "pts"][off[i]:off[i + 1]].copy()) for i in range(len(off) - 1)]


def ops_to_arrays(ops, prefix):
    """An operation list as (codes, one flat payload array, payload offsets)."""
    codes, payload, off = [], [], [0]
    for op in ops:
        codes.append(op[0])
        if op[0] in (2, 7):
            payload.append(np.asarray(op[1], dtype=np.float64).ravel())
        elif op[0] == 4:
            payload.append(np.concatenate([np.ravel(a) for a in op[1:]]).astype(np.float64))
        elif op[0] == 6:
            payload.append(np.array([float(op[1])]))
        else:
            payload.append(np.zeros(0))
        off.append(off[-1] + payload[-1].size)
    return {prefix + "codes": np.array(codes, dtype=np.int32), prefix + "payload": np.concatenate(payload) if payload else np.zeros(0),
            prefix + "poff": np.array(off, dtype=np.int64)}


def arrays_to_ops(z, prefix):
    ops = []
    for k, code in enumerate(z[prefix + "codes"]):
        p = z[prefix + "payload"][z[prefix + "poff"][k]:z[prefix + "poff"][k + 1]]
        code = int(code)
        if code in (2, 7):
            ops.append((code, p.copy()))
        elif code == 4:
            ops.append((code, p[0:9].reshape(3, 3).copy(), p[9:12].copy(), p[12:21].reshape(3, 3).copy(), p[21:24].copy()))
        elif code == 6:
            ops.append((code, int(p[0])))
        else:
            ops.append((code,))
    return ops


def scenario_ops(vio, kind, tracks, poses, ext, seed=1):
    """The operation sequences Estimator runs around a solve (estimator.cpp:614-617 double2vector -> setDepth,
    :157-166 removeFailures, :1187-1199 slideWindowOld, :1202-1205 slideWindowNew)."""
    rng = np.random.RandomState(seed)
    n_use = sum(usable(t) for t in tracks)
    R = [vio.synth.quat_to_rot(poses[k, 3:7]) for k in range(11)]
    ric, tic = vio.synth.quat_to_rot(ext[3:7]), ext[0:3]
    if kind == "depth_vector":
        x = rng.uniform(0.05, 0.5, n_use)
        x[::7] *= -1.0                                   # negative inverse depths: solve_flag 2, removed
        return [(2, x), (3,)]
    if kind == "shift_old":                              # slideWindowOld with depth shift
        return [(4, R[0] @ ric, poses[0, 0:3] + R[0] @ tic, R[1] @ ric, poses[1, 0:3] + R[1] @ tic)]
    if kind == "shift_old_init":                         # before initialisation: removeBack
        return [(7, rng.uniform(0.1, 0.4, n_use)), (5,)]
    if kind == "shift_new":
        return [(6, WINDOW_SIZE)]                        # slideWindowNew
    if kind == "behind_camera":                          # a re-hosted point lands behind the new host: INIT_DEPTH
        return [(4, R[0] @ ric, poses[0, 0:3] + R[0] @ tic, -(R[1] @ ric), poses[1, 0:3] + R[1] @ tic)]
    if kind == "frame_chain":                            # triangulate, solve result in, failures out, slide, again
        x = rng.uniform(0.05, 0.5, n_use)
        x[3::11] *= -1.0
        return [(1,), (2, x), (3,), (4, R[0] @ ric, poses[0, 0:3] + R[0] @ tic, R[1] @ ric, poses[1, 0:3] + R[1] @ tic), (1,)]
    raise ValueError(kind)


SCENARIOS = ("depth_vector", "shift_old", "shift_old_init", "shift_new", "behind_camera", "frame_chain")


def compare_tracks(got, want, depth_rtol=1e-14):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["id"] == w["id"] and g["start"] == w["start"] and g["flag"] == w["flag"], (g, w)
        np.testing.assert_array_equal(g["pts"], w["pts"])
        np.testing.assert_allclose(g["depth"], w["depth"], rtol=depth_rtol)
