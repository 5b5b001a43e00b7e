"""Round 6's two shortcuts of the frame loop leave the results as they are:
  * vio_solve after vio_linearize on the same state and graph starts from that system instead of forming it again (csrc/vio_api.cpp: lin_fresh) —
    and must form it again when anything came between the two calls;
  * vio_prepare builds and uploads the new graph's plan ahead of the call that needs it (the background-marginalisation frame loop)."""
import numpy as np
import pytest

import vio_testutil as tu

pytestmark = pytest.mark.gpu


def solve_digest(c, its=10):
    rep = c.solve(its)
    p, s, e = c.get_window()
    return (rep.iterations, rep.trials, rep.final_chi2, rep.final_lambda, p.tobytes(), s.tobytes(), e.tobytes(), np.asarray(c.get_landmarks()).tobytes(),
            np.asarray(rep.chi2_trace[:rep.iterations + 1]).tobytes(), np.asarray(rep.lambda_trace[:rep.iterations + 1]).tobytes())


@pytest.mark.parametrize("n,seed,ragged", [(150, 300, True), (2000, 14, False)])
def test_solve_after_linearize_is_the_solve(vio, hip_lib, n, seed, ragged):
    cp = hip_lib.context()
    cp.load(vio.synth.make_window(200, seed=71, t0=0.9))
    cp.solve(10)
    prior = cp.marginalize(vio.MARG_OLD)
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    w.prior = prior
    a, b, c = hip_lib.context(), hip_lib.context(), hip_lib.context()
    a.load(w)
    ref = solve_digest(a)
    b.load(w)
    b.linearize()
    b.synchronize()
    assert solve_digest(b) == ref                       # starts from vio_linearize's system: the same bits
    # something between the two calls: the solve must not start from the stale system
    c.load(w)
    c.linearize()
    p, s, e = c.get_window()
    p2 = p.copy()
    p2[3, 0:3] += 0.01
    c.set_window(p2, s, e)
    w2 = w.copy()
    w2.poses = p2
    d = hip_lib.context()
    d.load(w2)
    assert solve_digest(c) == solve_digest(d)
    # ... and a second solve on a solved context linearises again: its state is the first solve's result, not what vio_linearize saw
    b.linearize()
    first = solve_digest(b, 2)
    second = solve_digest(b, 2)
    assert second[2] <= first[2] * (1 + 1e-9) and second != first


def test_prepare_leaves_the_frame_loop_as_it_is(vio, hip_lib):
    wins = [vio.synth.make_window(150, seed=300 + r, t0=1.0 + 0.1 * r, ragged=True) for r in range(5)]

    def run(with_prepare):
        c = hip_lib.context()
        prior, out = None, []
        for r, w in enumerate(wins):
            c.set_window(w.poses, w.speed_bias, w.ext)
            c.set_landmarks(w.inv_depth)
            c.set_observations(w.lm, w.host, w.target, w.pts_i, w.pts_j)
            c.set_imu_all(w.preint)
            if with_prepare:
                c.prepare()
            if r > 0:
                prior = c.marginalize_end()
            c.set_prior(prior)
            out.append(solve_digest(c))
            c.marginalize_begin(vio.MARG_OLD)
        last = c.marginalize_end()
        return out, {k: np.asarray(v).tobytes() for k, v in last.items()}

    assert run(True) == run(False)
