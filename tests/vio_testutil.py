"""Helpers shared by the tests: window (de)serialisation for golden fixtures, comparison metrics."""
import numpy as np

PREINT_FIELDS = ("sum_dt", "delta_p", "delta_q", "delta_v", "linearized_ba", "linearized_bg", "jacobian", "covariance")
PRIOR_FIELDS = ("H", "b", "err", "jt_inv")


def window_to_arrays(w, prefix="in_"):
    d = {}
    xyz = getattr(w, "xyz", None) is not None       # synth.make_window_xyz: VertexPointXYZ landmarks
    for k in (("poses", "speed_bias", "ext", "xyz", "lm", "frame", "pts") if xyz else
              ("poses", "speed_bias", "ext", "inv_depth", "lm", "host", "target", "pts_i", "pts_j")):
        d[prefix + k] = np.asarray(getattr(w, k))
    if any(p is None for p in w.preint):            # estimator.cpp:956-970: an IMU edge with sum_dt > 10 is not added
        d[prefix + "pre_valid"] = np.array([p is not None for p in w.preint], dtype=np.int32)
    size = {"sum_dt": 1, "delta_p": 3, "delta_q": 4, "delta_v": 3, "linearized_ba": 3, "linearized_bg": 3, "jacobian": 225, "covariance": 225}
    for f in PREINT_FIELDS:
        d[prefix + "pre_" + f] = np.stack([np.zeros(size[f]) if p is None else np.asarray(p[f], dtype=np.float64).reshape(-1)
                                           for p in w.preint])
    if w.prior is not None:
        for f in PRIOR_FIELDS:
            d[prefix + "prior_" + f] = np.asarray(w.prior[f])
    return d


def arrays_to_window(vio, z, prefix="in_"):
    pre = []
    n_edges = z[prefix + "pre_sum_dt"].shape[0]
    valid = z[prefix + "pre_valid"] if prefix + "pre_valid" in z else np.ones(n_edges, dtype=np.int32)
    for k in range(n_edges):
        if not valid[k]:
            pre.append(None)
            continue
        p = {}
        for f in PREINT_FIELDS:
            a = z[prefix + "pre_" + f][k]
            p[f] = float(a.reshape(-1)[0]) if f == "sum_dt" else a.copy()
        pre.append(p)
    prior = None
    if prefix + "prior_H" in z:
        prior = {f: z[prefix + "prior_" + f].copy() for f in PRIOR_FIELDS}
    lm = z[prefix + "lm"]
    if prefix + "xyz" in z:
        return vio.synth.Window(poses=z[prefix + "poses"].copy(), speed_bias=z[prefix + "speed_bias"].copy(),
                                ext=z[prefix + "ext"].copy(), xyz=z[prefix + "xyz"].copy(), lm=lm.copy(),
                                frame=z[prefix + "frame"].copy(), pts=z[prefix + "pts"].copy(), preint=pre, prior=prior,
                                n_landmarks=int(z[prefix + "xyz"].shape[0]), n_observations=int(lm.size))
    return vio.synth.Window(poses=z[prefix + "poses"].copy(), speed_bias=z[prefix + "speed_bias"].copy(),
                            ext=z[prefix + "ext"].copy(), inv_depth=z[prefix + "inv_depth"].copy(),
                            lm=lm.copy(), host=z[prefix + "host"].copy(), target=z[prefix + "target"].copy(),
                            pts_i=z[prefix + "pts_i"].copy(), pts_j=z[prefix + "pts_j"].copy(), preint=pre, prior=prior,
                            n_landmarks=int(z[prefix + "inv_depth"].size), n_observations=int(lm.size))


def rel_max(a, b):
    """max |a-b| relative to max |b|."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


def abs_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) if a.size else 0.0


def scaled_sym_err(A, B):
    """max_ij |A-B|_ij / sqrt(|B_ii| |B_jj|): the metric that makes sense for the badly scaled pose Hessian
    (IMU bias information ~1e16 next to visual entries ~1e5)."""
    d = np.sqrt(np.maximum(np.abs(np.diag(B)), 1e-300))
    return float((np.abs(A - B) / np.outer(d, d)).max())


CAM_IDX = list(range(6)) + [6 + 15 * f + k for f in range(11) for k in range(6)]


def run_stepwise(ctx, lam=None):
    """linearize -> init_lm -> solve_linear(lambda0) -> update -> chi2 -> eval_step; returns everything observable."""
    out = {}
    ctx.linearize()
    out["Hs"], out["bs"] = ctx.get_schur_system()
    out["hll"], out["bl"] = ctx.get_landmark_system()
    out["bpp"], out["diag"] = ctx.get_pose_gradient()
    chi0, lam0 = ctx.init_lm()
    out["chi0"], out["lambda0"] = np.float64(chi0), np.float64(lam0)
    ctx.solve_linear(lam0 if lam is None else lam)
    out["dx_pose"], out["dx_lm"] = ctx.get_delta()
    ctx.update_states()
    out["poses1"], out["sb1"], out["ext1"] = ctx.get_window()
    out["invd1"] = ctx.get_landmarks() if ctx.lm_dim == 1 else ctx.get_landmarks_xyz()
    out["bprior1"], out["errprior1"] = ctx.get_prior()
    out["chi1"] = np.float64(ctx.chi2())
    ok, chi, lam1 = ctx.eval_step()
    out["accepted"], out["chi_after"], out["lambda1"] = np.int32(ok), np.float64(chi), np.float64(lam1)
    return out


def run_solve(ctx, iterations=10):
    rep = ctx.solve(iterations)
    out = {}
    out["posesF"], out["sbF"], out["extF"] = ctx.get_window()
    out["invdF"] = ctx.get_landmarks() if ctx.lm_dim == 1 else ctx.get_landmarks_xyz()
    out["bpriorF"], out["errpriorF"] = ctx.get_prior()
    out["final_chi2"], out["final_lambda"] = np.float64(rep.final_chi2), np.float64(rep.final_lambda)
    out["iterations"] = np.int32(rep.iterations)
    out["chi2_trace"] = np.array(list(rep.chi2_trace[:max(rep.iterations, 1)]))
    out["lambda_trace"] = np.array(list(rep.lambda_trace[:max(rep.iterations, 1)]))
    return out, rep


def check_mapping_protocol(vio, lib, pytest):
    """The ways out of a mapping (include/vio_backend.h), the same on the oracle and on the HIP library: vio_set_observations drops what
    was written in place and sets its own list (map -> set -> solve gives the bits of set -> solve); vio_set_landmarks with another
    count invalidates the mapping: the commit is refused, the context has no list, and a fresh map -> commit works again; an
    invalid first edge (landmark -1) is refused without a look at the edge before it."""
    import numpy as np
    w = vio.synth.make_window(90, seed=15, ragged=True)
    a, b = lib.context(), lib.context()
    a.load(w)
    b.set_window(w.poses, w.speed_bias, w.ext)
    b.set_landmarks(w.inv_depth)
    lm, host, target, pi, pj = b.map_observations(7)
    lm[:] = 3                                   # rubbish that is never committed
    b.set_observations(w.lm, w.host, w.target, w.pts_i, w.pts_j)
    for k, p in enumerate(w.preint):
        b.set_imu(k, p)
    b.set_prior(None)
    ra, rb = a.solve(10), b.solve(10)
    assert ra.final_chi2 == rb.final_chi2 and ra.iterations == rb.iterations
    assert np.array_equal(a.get_landmarks(), b.get_landmarks())
    with pytest.raises(vio.VioError):
        b.commit_observations()                 # the set ended the mapping
    # another landmark count while mapped
    w2 = vio.synth.make_window(40, seed=16)
    lm, host, target, pi, pj = b.map_observations(w2.n_observations)
    b.set_landmarks(w2.inv_depth)
    with pytest.raises(vio.VioError) as e:
        b.commit_observations()
    assert e.value.status == -1 and "invalidated" in str(e.value)
    with pytest.raises(vio.VioError):
        b.solve(10)                             # no list (and no silent vision-less solve)
    b.set_window(w2.poses, w2.speed_bias, w2.ext)
    lm, host, target, pi, pj = b.map_observations(w2.n_observations)
    lm[:], host[:], target[:], pi[:], pj[:] = w2.lm, w2.host, w2.target, w2.pts_i, w2.pts_j
    b.commit_observations()
    a.load(w2)
    ra, rb = a.solve(10), b.solve(10)
    assert ra.final_chi2 == rb.final_chi2 and np.array_equal(a.get_landmarks(), b.get_landmarks())
    # an invalid FIRST edge
    bad = w2.lm.copy()
    bad[0] = -1
    with pytest.raises(vio.VioError):
        b.set_observations(bad, w2.host, w2.target, w2.pts_i, w2.pts_j)
