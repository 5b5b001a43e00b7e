"""ctypes binding of the C ABI declared in include/vio_backend.h.

The product library is `csrc/libvio_hip.so` (prefix ``vio_``).  The same binding class is
parametrised by (path, prefix) so that the tests can drive the CPU oracle (``vioo_``) and the
compiled reference harness (``vior_``) through the identical surface; nothing in this package loads
those two libraries.
"""
import ctypes as C
import os

import numpy as np

WINDOW_SIZE = 10          # VM/include/parameters.h:35
NUM_FRAMES = WINDOW_SIZE + 1
POSE_DIM = 6 + 15 * NUM_FRAMES     # 171
PRIOR_DIM = POSE_DIM - 15          # 156
CAM_DIM = 6 + 6 * NUM_FRAMES       # 72

LOSS_TRIVIAL, LOSS_HUBER, LOSS_CAUCHY, LOSS_TUKEY = 0, 1, 2, 3
MARG_OLD, MARG_SECOND_NEW = 0, 1
ITEMS_LATENCY, ITEMS_THROUGHPUT = 0, 1
ORDER_EIGEN, ORDER_CHAIN = 0, 1          # vio_solve_order: elimination order of the damped pose solve

STATUS = {0: "VIO_OK", -1: "VIO_ERR_BAD_ARG", -2: "VIO_ERR_HIP", -3: "VIO_ERR_NOT_FINITE",
          -4: "VIO_ERR_EMPTY", -5: "VIO_ERR_UNSUPPORTED", -6: "VIO_ERR_NO_DEVICE"}


class VioConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("ext_fixed", C.c_int32), ("loss_type", C.c_int32),
                ("item_policy", C.c_int32), ("loss_delta", C.c_double), ("reproj_sqrt_info", C.c_double),
                ("gravity", C.c_double * 3), ("stream", C.c_void_p), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32)]


class VioPreint(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4),
                ("delta_v", C.c_double * 3), ("linearized_ba", C.c_double * 3),
                ("linearized_bg", C.c_double * 3), ("jacobian", C.c_double * 225),
                ("covariance", C.c_double * 225)]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        v = np.frombuffer(p, dtype=np.float64)       # the struct is 467 contiguous doubles: filled through a view (a ctypes array
        v[0] = float(d["sum_dt"])                    # assigned from a list costs 10 us per pre-integration, 0.1 ms per frame)
        o = 1
        for name, n in (("delta_p", 3), ("delta_q", 4), ("delta_v", 3), ("linearized_ba", 3),
                        ("linearized_bg", 3), ("jacobian", 225), ("covariance", 225)):
            arr = np.asarray(d[name], dtype=np.float64).reshape(-1)
            assert arr.size == n, name
            v[o:o + n] = arr
            o += n
        return p


class VioSolveReport(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("accepted", C.c_int32),
                ("stop_reason", C.c_int32), ("initial_chi2", C.c_double), ("final_chi2", C.c_double),
                ("final_lambda", C.c_double), ("solve_ms", C.c_double), ("hessian_ms", C.c_double),
                ("chi2_trace", C.c_double * 128), ("lambda_trace", C.c_double * 128)]


class VioError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: %s %s" % (where, STATUS.get(status, status), detail))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class VioLib:
    """One shared library exporting the vio_backend.h surface under `prefix`."""

    SYMBOLS = ["create", "destroy", "last_error", "default_config", "set_window", "set_landmarks",
               "set_observations", "set_imu", "set_prior", "solve", "linearize", "init_lm",
               "solve_linear", "update_states", "rollback_states", "chi2", "eval_step", "gn_iteration",
               "synchronize", "marginalize", "get_window", "get_landmarks", "get_prior", "get_delta",
               "get_schur_system", "get_landmark_system", "get_pose_gradient", "exchange_buffers",
               "set_exchange_hook", "bind_exchange_buffers", "set_landmarks_xyz", "set_observations_xyz",
               "get_landmarks_xyz", "set_config"]
    # outside the backend proper: the compiled-reference harness (vior_) has no FeatureManager
    # ... nor the receive side of the sharded exchange (it never shards)
    # ... nor the two-halves marginalisation (vio_marginalize_begin / _end)
    # ... nor the in-place observation list (vio_map_observations / vio_commit_observations)
    # ... nor the ten IMU edges in one call (vio_set_imu_all)
    # ... nor the plan built ahead of the solve (vio_prepare)
    OPTIONAL = ["triangulate", "gather_buffers", "bind_gather_buffers", "marginalize_begin", "marginalize_end",
                "map_observations", "commit_observations", "set_imu_all", "prepare"]

    # exported by the HIP library only (measurement, caller-owned exchange buffers)
    HIP_ONLY = ["profile_begin", "profile_begin_sampled", "profile_end", "kernel_name", "preintegrate",
                "comm_unique_id", "comm_init", "comm_destroy", "comm_info", "get_stream", "batch_gn_iteration", "batch_solve", "get_host_timing",
                "set_solve_order", "get_solve_order"]
    # diagnostic entry points: only in a build with -DVIO_DEBUG_ENTRY_POINTS (csrc/diag/libvio_hip_debug.so, the tests')
    HIP_DEBUG = ["debug_chain_solve"]
    KERNELS = ["k_linearize", "k_reduce", "k_assemble", "k_pose_solve", "k_backsub", "k_lm_decide"]

    def __init__(self, path, prefix="vio_"):
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found — build it first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.prefix = prefix
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        self.fn = {}
        for s in self.SYMBOLS:
            self.fn[s] = getattr(self.dll, prefix + s)     # raises AttributeError on a missing export
        for s in self.OPTIONAL:
            if hasattr(self.dll, prefix + s):
                self.fn[s] = getattr(self.dll, prefix + s)
                self.fn[s].restype = C.c_int
        if prefix == "vio_":
            for s in self.HIP_ONLY:
                self.fn[s] = getattr(self.dll, prefix + s)
                self.fn[s].restype = C.c_int
            self.fn["kernel_name"].restype = C.c_char_p
            for s in self.HIP_DEBUG:
                if hasattr(self.dll, prefix + s):
                    self.fn[s] = getattr(self.dll, prefix + s)
                    self.fn[s].restype = C.c_int
        self.fn["last_error"].restype = C.c_char_p
        self.fn["destroy"].restype = None
        self.fn["default_config"].restype = None
        for s in self.SYMBOLS:
            if s not in ("last_error", "destroy", "default_config"):
                self.fn[s].restype = C.c_int

    def preintegrate(self, acc0, gyr0, ba, bg, dts, accs, gyrs, acc_n, gyr_n, acc_w, gyr_w):
        """IntegrationBase on the host through the ABI (vio_preintegrate); returns a VioPreint."""
        out = VioPreint()
        a = [_f64(x).reshape(-1) for x in (acc0, gyr0, ba, bg, dts, accs, gyrs)]
        fn = self.raw("preintegrate") if "preintegrate" not in self.fn else self.fn["preintegrate"]
        fn.restype = C.c_int
        st = fn(_dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), C.c_int32(a[4].size), _dp(a[4]), _dp(a[5]), _dp(a[6]),
                C.c_double(acc_n), C.c_double(gyr_n), C.c_double(acc_w), C.c_double(gyr_w), C.byref(out))
        if st != 0:
            raise VioError(st, self.prefix + "preintegrate")
        return out

    def comm_unique_id(self):
        """ncclGetUniqueId through the ABI (rank 0): 128 bytes to hand to every rank's VioContext.comm_init."""
        buf = (C.c_char * 128)()
        st = self.fn["comm_unique_id"](buf)
        if st != 0:
            raise VioError(st, self.prefix + "comm_unique_id", "(is librccl.so loadable?)")
        return bytes(buf.raw)

    def batch_gn_iteration(self, ctxs, lam):
        """One fixed-lambda GN iteration of every window of `ctxs` in one launch per kernel (vio_batch_gn_iteration)."""
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        st = self.fn["batch_gn_iteration"](arr, C.c_int32(len(ctxs)), C.c_double(lam))
        if st != 0:
            msg = self.fn["last_error"](ctxs[0].h)
            raise VioError(st, self.prefix + "batch_gn_iteration", (msg or b"").decode(errors="replace"))

    def batch_solve(self, ctxs, iterations=10):
        """Problem::Solve of every window of `ctxs` with one launch per kernel for the batch (vio_batch_solve); the reports."""
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        reps = (VioSolveReport * len(ctxs))()
        st = self.fn["batch_solve"](arr, C.c_int32(len(ctxs)), C.c_int32(iterations), reps)
        if st != 0:
            msg = self.fn["last_error"](ctxs[0].h)
            raise VioError(st, self.prefix + "batch_solve", (msg or b"").decode(errors="replace"))
        return list(reps)

    def has(self, name):
        return hasattr(self.dll, self.prefix + name)

    def raw(self, name):
        return getattr(self.dll, self.prefix + name)

    def default_config(self):
        cfg = VioConfig()
        self.fn["default_config"](C.byref(cfg))
        return cfg

    def context(self, cfg=None, **overrides):
        cfg = cfg or self.default_config()
        # (diagnostics: VIO_DEFAULT_ITEM_POLICY=1 makes every context of a tool a throughput-policy one — the randomised sweeps of
        # tools/fuzz_*.py through the half-width kernels — unless the caller says otherwise)
        if "item_policy" not in overrides and os.environ.get("VIO_DEFAULT_ITEM_POLICY"):
            cfg.item_policy = int(os.environ["VIO_DEFAULT_ITEM_POLICY"])
        for k, v in overrides.items():
            if k == "gravity":
                cfg.gravity[:] = list(v)
            else:
                setattr(cfg, k, v)
        return VioContext(self, cfg)


class VioContext:
    """Owns one vio_ctx handle.  Methods mirror the reference's Problem call sequence."""

    def __init__(self, lib, cfg):
        self.lib = lib
        self.cfg = cfg
        self.h = C.c_void_p()
        self.n = 0
        self.m = 0
        self.lm_dim = 1     # 1: inverse depths, 3: XYZ landmarks (set_landmarks / set_landmarks_xyz switch it)
        st = lib.fn["create"](C.byref(cfg), C.byref(self.h))
        if st != 0:
            raise VioError(st, lib.prefix + "create")

    def set_config(self, **overrides):
        """vio_set_config: ext_fixed / loss_type / loss_delta / reproj_sqrt_info / gravity on the living context."""
        for k, v in overrides.items():
            if k == "gravity":
                self.cfg.gravity[:] = list(v)
            else:
                setattr(self.cfg, k, v)
        self._ck(self.lib.fn["set_config"](self.h, C.byref(self.cfg)), "set_config")

    def close(self):
        if self.h:
            self.lib.fn["destroy"](self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, where):
        if st != 0:
            msg = self.lib.fn["last_error"](self.h)
            raise VioError(st, self.lib.prefix + where, (msg or b"").decode(errors="replace"))

    # ---- graph construction ---------------------------------------------------------------
    def set_window(self, poses, speed_bias, ext):
        p, s, e = _f64(poses, (NUM_FRAMES, 7)), _f64(speed_bias, (NUM_FRAMES, 9)), _f64(ext, (7,))
        self._ck(self.lib.fn["set_window"](self.h, _dp(p), _dp(s), _dp(e)), "set_window")

    def set_landmarks(self, inv_depth):
        d = _f64(inv_depth).reshape(-1)
        self.n, self.lm_dim = d.size, 1
        self._ck(self.lib.fn["set_landmarks"](self.h, C.c_int64(d.size), _dp(d)), "set_landmarks")

    def set_landmarks_xyz(self, xyz):
        """VertexPointXYZ x N: world points [N][3]; the context now holds XYZ landmarks."""
        d = _f64(xyz).reshape(-1, 3)
        self.n, self.lm_dim = d.shape[0], 3
        self._ck(self.lib.fn["set_landmarks_xyz"](self.h, C.c_int64(self.n), _dp(d)), "set_landmarks_xyz")

    def set_observations_xyz(self, lm, frame, pts):
        """EdgeReprojectionXYZ x M: (landmark, observing frame, normalised observation)."""
        lm = np.ascontiguousarray(lm, dtype=np.int32)
        frame = np.ascontiguousarray(frame, dtype=np.int32)
        self.m = lm.size
        p = _f64(pts, (self.m, 2))
        self._ck(self.lib.fn["set_observations_xyz"](self.h, C.c_int64(self.m), _ip(lm), _ip(frame), _dp(p)),
                 "set_observations_xyz")

    def set_observations(self, lm, host, target, pts_i, pts_j):
        lm = np.ascontiguousarray(lm, dtype=np.int32)
        host = np.ascontiguousarray(host, dtype=np.int32)
        target = np.ascontiguousarray(target, dtype=np.int32)
        self.m = lm.size
        pi, pj = _f64(pts_i, (self.m, 2)), _f64(pts_j, (self.m, 2))
        self._ck(self.lib.fn["set_observations"](self.h, C.c_int64(self.m), _ip(lm), _ip(host), _ip(target),
                                                 _dp(pi), _dp(pj)), "set_observations")

    def map_observations(self, m):
        """vio_map_observations: the library's own arrays for m edges as numpy views (lm, host, target [m] int32; pts_i, pts_j [m][2]
        float64) to be filled in place, then commit_observations()."""
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        lm, host, target, pi, pj = ip(), ip(), ip(), dp(), dp()
        self._ck(self.lib.fn["map_observations"](self.h, C.c_int64(m), C.byref(lm), C.byref(host), C.byref(target), C.byref(pi), C.byref(pj)),
                 "map_observations")
        self.m = int(m)
        if m == 0:
            z = np.zeros(0, dtype=np.int32)
            return z, z.copy(), z.copy(), np.zeros((0, 2)), np.zeros((0, 2))
        view = np.ctypeslib.as_array
        return (view(lm, (m,)), view(host, (m,)), view(target, (m,)), view(pi, (2 * m,)).reshape(m, 2), view(pj, (2 * m,)).reshape(m, 2))

    def commit_observations(self):
        self._ck(self.lib.fn["commit_observations"](self.h), "commit_observations")

    def set_imu(self, k, pre):
        if pre is None:
            self._ck(self.lib.fn["set_imu"](self.h, C.c_int32(k), None), "set_imu")
        else:
            p = pre if isinstance(pre, VioPreint) else VioPreint.from_dict(pre)
            self._ck(self.lib.fn["set_imu"](self.h, C.c_int32(k), C.byref(p)), "set_imu")

    def set_imu_all(self, pres):
        """vio_set_imu_all: the ten edges (dicts, VioPreint or None) in one crossing of the boundary; libraries without it: ten calls."""
        pres = list(pres)
        if len(pres) != WINDOW_SIZE:           # (a shorter list would silently pass NULL — "no edge" — for the missing ones: ADVICE r05)
            raise ValueError("set_imu_all takes the window's %d IMU edges (None for a missing one), got %d" % (WINDOW_SIZE, len(pres)))
        if "set_imu_all" not in self.lib.fn:
            for k, pre in enumerate(pres):
                self.set_imu(k, pre)
            return
        keep = [None if p is None else (p if isinstance(p, VioPreint) else VioPreint.from_dict(p)) for p in pres]
        arr = (C.POINTER(VioPreint) * WINDOW_SIZE)(*[C.pointer(p) if p is not None else C.POINTER(VioPreint)() for p in keep])
        self._ck(self.lib.fn["set_imu_all"](self.h, arr), "set_imu_all")

    def set_prior(self, prior):
        if prior is None:
            self._ck(self.lib.fn["set_prior"](self.h, C.c_int32(0), None, None, None, None), "set_prior")
            return
        H, b = _f64(prior["H"], (PRIOR_DIM, PRIOR_DIM)), _f64(prior["b"], (PRIOR_DIM,))
        err, jt = _f64(prior["err"], (PRIOR_DIM,)), _f64(prior["jt_inv"], (PRIOR_DIM, PRIOR_DIM))
        self._ck(self.lib.fn["set_prior"](self.h, C.c_int32(PRIOR_DIM), _dp(H), _dp(b), _dp(err), _dp(jt)),
                 "set_prior")

    def load(self, w):
        """Upload a synth.Window (or any object/dict with the same fields)."""
        g = (lambda k: w[k]) if isinstance(w, dict) else (lambda k: getattr(w, k))
        self.set_window(g("poses"), g("speed_bias"), g("ext"))
        xyz = w.get("xyz") if isinstance(w, dict) else getattr(w, "xyz", None)
        if xyz is not None:                                   # synth.make_window_xyz
            self.set_landmarks_xyz(xyz)
            self.set_observations_xyz(g("lm"), g("frame"), g("pts"))
        else:
            self.set_landmarks(g("inv_depth"))
            self.set_observations(g("lm"), g("host"), g("target"), g("pts_i"), g("pts_j"))
        pres = list(g("preint"))
        if len(pres) == WINDOW_SIZE:
            self.set_imu_all(pres)
        else:
            for k, pre in enumerate(pres):
                self.set_imu(k, pre)
        self.set_prior(g("prior"))

    # ---- solve ----------------------------------------------------------------------------
    def solve(self, iterations=10):
        rep = VioSolveReport()
        self._ck(self.lib.fn["solve"](self.h, C.c_int32(iterations), C.byref(rep)), "solve")
        return rep

    def prepare(self):
        """vio_prepare: the graph's plan built and uploaded now (a no-op for libraries without it)."""
        if "prepare" in self.lib.fn:
            self._ck(self.lib.fn["prepare"](self.h), "prepare")

    def linearize(self):
        self._ck(self.lib.fn["linearize"](self.h), "linearize")

    def init_lm(self):
        chi, lam = C.c_double(), C.c_double()
        self._ck(self.lib.fn["init_lm"](self.h, C.byref(chi), C.byref(lam)), "init_lm")
        return chi.value, lam.value

    def solve_linear(self, lam):
        self._ck(self.lib.fn["solve_linear"](self.h, C.c_double(lam)), "solve_linear")

    def update_states(self):
        self._ck(self.lib.fn["update_states"](self.h), "update_states")

    def rollback_states(self):
        self._ck(self.lib.fn["rollback_states"](self.h), "rollback_states")

    def chi2(self):
        chi = C.c_double()
        self._ck(self.lib.fn["chi2"](self.h, C.byref(chi)), "chi2")
        return chi.value

    def eval_step(self):
        ok, chi, lam = C.c_int32(), C.c_double(), C.c_double()
        self._ck(self.lib.fn["eval_step"](self.h, C.byref(ok), C.byref(chi), C.byref(lam)), "eval_step")
        return bool(ok.value), chi.value, lam.value

    def gn_iteration(self, lam):
        self._ck(self.lib.fn["gn_iteration"](self.h, C.c_double(lam)), "gn_iteration")

    def synchronize(self):
        self._ck(self.lib.fn["synchronize"](self.h), "synchronize")

    def get_stream(self):
        st = C.c_void_p()
        self._ck(self.lib.fn["get_stream"](self.h, C.byref(st)), "get_stream")
        return st.value

    def marginalize(self, kind, allow_nonfinite=False):
        """The new prior.  allow_nonfinite: VIO_ERR_NOT_FINITE (a landmark block without an inverse) hands back what the reference
        leaves in that case — H_prior 0, the rest NaN — instead of raising."""
        H = np.zeros((PRIOR_DIM, PRIOR_DIM))
        b, err = np.zeros(PRIOR_DIM), np.zeros(PRIOR_DIM)
        jt = np.zeros((PRIOR_DIM, PRIOR_DIM))
        st = self.lib.fn["marginalize"](self.h, C.c_int32(kind), _dp(H), _dp(b), _dp(err), _dp(jt))
        if not (st == -3 and allow_nonfinite):
            self._ck(st, "marginalize")
        return {"H": H, "b": b, "err": err, "jt_inv": jt}

    def marginalize_begin(self, kind):
        """Device part of the marginalisation; its dense tail goes on in the background (vio_marginalize_begin)."""
        self._ck(self.lib.fn["marginalize_begin"](self.h, C.c_int32(kind)), "marginalize_begin")

    def marginalize_end(self, allow_nonfinite=False):
        H = np.zeros((PRIOR_DIM, PRIOR_DIM))
        b, err = np.zeros(PRIOR_DIM), np.zeros(PRIOR_DIM)
        jt = np.zeros((PRIOR_DIM, PRIOR_DIM))
        st = self.lib.fn["marginalize_end"](self.h, _dp(H), _dp(b), _dp(err), _dp(jt))
        if not (st == -3 and allow_nonfinite):
            self._ck(st, "marginalize_end")
        return {"H": H, "b": b, "err": err, "jt_inv": jt}

    # ---- read back ------------------------------------------------------------------------
    def get_window(self):
        p, s, e = np.zeros((NUM_FRAMES, 7)), np.zeros((NUM_FRAMES, 9)), np.zeros(7)
        self._ck(self.lib.fn["get_window"](self.h, _dp(p), _dp(s), _dp(e)), "get_window")
        return p, s, e

    def get_landmarks(self):
        d = np.zeros(max(self.n, 1))
        self._ck(self.lib.fn["get_landmarks"](self.h, C.c_int64(self.n), _dp(d)), "get_landmarks")
        return d[:self.n]

    def get_landmarks_xyz(self):
        d = np.zeros((max(self.n, 1), 3))
        self._ck(self.lib.fn["get_landmarks_xyz"](self.h, C.c_int64(self.n), _dp(d)), "get_landmarks_xyz")
        return d[:self.n]

    def get_prior(self):
        b, err = np.zeros(POSE_DIM), np.zeros(PRIOR_DIM)
        self._ck(self.lib.fn["get_prior"](self.h, _dp(b), _dp(err)), "get_prior")
        return b, err

    def get_delta(self):
        d = self.lm_dim
        dp, dl = np.zeros(POSE_DIM), np.zeros(max(self.n, 1) * d)
        self._ck(self.lib.fn["get_delta"](self.h, _dp(dp), C.c_int64(self.n), _dp(dl)), "get_delta")
        return dp, (dl[:self.n] if d == 1 else dl[:self.n * d].reshape(self.n, d))

    def get_schur_system(self):
        H, b = np.zeros((POSE_DIM, POSE_DIM)), np.zeros(POSE_DIM)
        self._ck(self.lib.fn["get_schur_system"](self.h, _dp(H), _dp(b)), "get_schur_system")
        return H, b

    def get_landmark_system(self):
        d = self.lm_dim
        h, b = np.zeros(max(self.n, 1) * d * d), np.zeros(max(self.n, 1) * d)
        self._ck(self.lib.fn["get_landmark_system"](self.h, C.c_int64(self.n), _dp(h), _dp(b)),
                 "get_landmark_system")
        if d == 1:
            return h[:self.n], b[:self.n]
        return h[:self.n * d * d].reshape(self.n, d, d), b[:self.n * d].reshape(self.n, d)

    def get_pose_gradient(self):
        b, d = np.zeros(POSE_DIM), np.zeros(POSE_DIM)
        self._ck(self.lib.fn["get_pose_gradient"](self.h, _dp(b), _dp(d)), "get_pose_gradient")
        return b, d

    def bind_exchange_buffers(self, reduced_ptr, scalars_ptr):
        self._ck(self.lib.fn["bind_exchange_buffers"](self.h, C.c_void_p(reduced_ptr), C.c_void_p(scalars_ptr)),
                 "bind_exchange_buffers")

    def gather_buffers(self):
        """Receive side of the sharded exchange (vio_gather_buffers): pointers of [shard_count][n_reduced], [shard_count][n_scalars]."""
        p0, p1 = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.fn["gather_buffers"](self.h, C.byref(p0), C.byref(p1)), "gather_buffers")
        return p0.value, p1.value

    def bind_gather_buffers(self, gathered_ptr, gathered_scalars_ptr):
        self._ck(self.lib.fn["bind_gather_buffers"](self.h, C.c_void_p(gathered_ptr), C.c_void_p(gathered_scalars_ptr)),
                 "bind_gather_buffers")

    def set_exchange_hook(self, fn):
        """fn(which) -> 0 on success; kept alive on the context."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
        self._hook = proto(lambda user, which: int(fn(which))) if fn is not None else None
        self._ck(self.lib.fn["set_exchange_hook"](self.h, self._hook if self._hook else C.cast(None, proto), None),
                 "set_exchange_hook")

    def triangulate(self, start_frame, obs_offset, pts, poses, ext, depth, init_depth=5.0):
        """FeatureManager::triangulate for tracks in CSR form; returns the updated depth vector (entries > 0 are kept)."""
        sf = np.ascontiguousarray(start_frame, dtype=np.int32)
        off = np.ascontiguousarray(obs_offset, dtype=np.int64)
        p = _f64(pts).reshape(-1, 2) if len(pts) else np.zeros((0, 2))
        d = np.array(depth, dtype=np.float64).copy()
        assert off.size == sf.size + 1 and d.size == sf.size and (sf.size == 0 or off[-1] == p.shape[0])
        self._ck(self.lib.fn["triangulate"](self.h, C.c_int64(sf.size), _ip(sf), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                            _dp(p), _dp(_f64(poses, (NUM_FRAMES, 7))), _dp(_f64(ext, (7,))),
                                            C.c_double(init_depth), _dp(d)), "triangulate")
        return d

    def comm_init(self, id128, rank, nranks):
        """Native RCCL exchange: id128 = the 128 bytes of rank 0's VioLib.comm_unique_id(), same on every rank."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._ck(self.lib.fn["comm_init"](self.h, buf, C.c_int32(rank), C.c_int32(nranks)), "comm_init")

    def comm_info(self):
        """(ranks RCCL's communicator spans, this rank's index in it); (0, -1) without a native communicator"""
        n, r = C.c_int32(), C.c_int32()
        self._ck(self.lib.fn["comm_info"](self.h, C.byref(n), C.byref(r)), "comm_info")
        return n.value, r.value

    def comm_destroy(self):
        self._ck(self.lib.fn["comm_destroy"](self.h), "comm_destroy")

    def profile_begin(self, which):
        self._ck(self.lib.fn["profile_begin"](self.h, C.c_int32(which)), "profile_begin")

    def profile_begin_sampled(self, which, every):
        self._ck(self.lib.fn["profile_begin_sampled"](self.h, C.c_int32(which), C.c_int32(every)), "profile_begin_sampled")

    def profile_end(self):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.lib.fn["profile_end"](self.h, C.byref(ms), C.byref(n)), "profile_end")
        return ms.value, n.value

    def set_solve_order(self, order):
        """vio_set_solve_order: ORDER_EIGEN (Eigen's LDLT pivot order) or ORDER_CHAIN (the static structured order, default)."""
        self._ck(self.lib.fn["set_solve_order"](self.h, C.c_int32(order)), "set_solve_order")

    def get_solve_order(self):
        """(requested, effective): a prior outside the chain pattern is solved in Eigen's order whatever was asked for."""
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.lib.fn["get_solve_order"](self.h, C.byref(a), C.byref(b)), "get_solve_order")
        return a.value, b.value

    def debug_chain_solve(self, H, b, lam, dump=False):
        """vio_debug_chain_solve: (H + lam I)^-1 b by the chain-order kernel alone; with dump, also the factor as left in LDS."""
        H, b = _f64(H, (POSE_DIM, POSE_DIM)), _f64(b, (POSE_DIM,))
        x = np.zeros(POSE_DIM)
        d = np.zeros(1 << 15) if dump else None
        self._ck(self.lib.fn["debug_chain_solve"](self.h, _dp(H), _dp(b), C.c_double(lam), _dp(x), _dp(d) if dump else None), "debug_chain_solve")
        return (x, d) if dump else x

    def host_timing(self):
        """vio_get_host_timing as a dict (microseconds; `marg_live_rows` is a count)."""
        out = (C.c_double * 8)()
        self._ck(self.lib.fn["get_host_timing"](self.h, out), "get_host_timing")
        keys = ("activate_pull_us", "activate_plan_us", "activate_push_us", "marg_device_us", "marg_tail_us", "marg_live_rows", "marg_prepare_us")
        return {k: out[i] for i, k in enumerate(keys)}

    def exchange_buffers(self):
        p0, n0, p1, n1 = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._ck(self.lib.fn["exchange_buffers"](self.h, C.byref(p0), C.byref(n0), C.byref(p1), C.byref(n1)),
                 "exchange_buffers")
        return (p0.value, n0.value), (p1.value, n1.value)
