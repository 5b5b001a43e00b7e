"""vio_batch_gn_iteration: B independent windows advanced by one launch per kernel (grid.y = window).  Every window is an
ordinary context; the batched sequence runs the same kernel bodies on the same data, so the result must equal B separate
vio_gn_iteration runs bit for bit — also when a getter looks at one window in the middle of the sequence."""
import numpy as np
import pytest

import vio_testutil as tu

pytestmark = pytest.mark.gpu


def windows(vio, oracle_lib):
    ws = [vio.synth.make_window(700, seed=1, ragged=True), vio.synth.make_window(2500, seed=2), vio.synth.make_window(300, seed=3, t0=1.1),
          vio.synth.make_window(64, seed=4), vio.synth.make_window(1, seed=5)]
    # the third window carries a marginalisation prior (of the oracle: test infrastructure makes the input, not the result)
    co = oracle_lib.context()
    wp = vio.synth.make_window(120, seed=9)
    co.load(wp)
    co.solve(10)
    p, s, e = co.get_window()
    wp.poses, wp.speed_bias, wp.ext, wp.inv_depth = p, s, e, co.get_landmarks()
    co.load(wp)
    ws[2].prior = co.marginalize(vio.MARG_OLD)
    return ws


def state_of(ctx, xyz=False):
    p, s, e = ctx.get_window()
    return p, s, e, (ctx.get_landmarks_xyz() if xyz else ctx.get_landmarks()), ctx.chi2()


def test_batched_gn_equals_separate_runs(vio, hip_lib, oracle_lib):
    ws = windows(vio, oracle_lib)
    lam = 5e5
    lead = hip_lib.context()
    batch = [lead] + [hip_lib.context(stream=lead.get_stream()) for _ in ws[1:]]
    solo = [hip_lib.context() for _ in ws]
    for c, w in zip(batch, ws):
        c.load(w)
    for c, w in zip(solo, ws):
        c.load(w)
    for it in range(5):
        hip_lib.batch_gn_iteration(batch, lam)
        for c in solo:
            c.gn_iteration(lam)
        if it == 1:             # a getter in the middle: settles that window's pending step test the classic way
            a, b = state_of(batch[1]), state_of(solo[1])
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
    for c, r in zip(batch, solo):
        for x, y in zip(state_of(c), state_of(r)):
            np.testing.assert_array_equal(x, y)
    # a window re-loaded in the middle of a batch's life, a smaller batch, a bigger one
    batch[3].load(ws[3])
    solo[3].load(ws[3])
    for _ in range(2):
        hip_lib.batch_gn_iteration(batch[:4], lam)
        for c in solo[:4]:
            c.gn_iteration(lam)
    hip_lib.batch_gn_iteration(batch, lam)
    for c in solo:
        c.gn_iteration(lam)
    for c, r in zip(batch, solo):
        for x, y in zip(state_of(c), state_of(r)):
            np.testing.assert_array_equal(x, y)


def test_solo_calls_on_batch_members_between_batch_calls(vio, hip_lib, oracle_lib):
    """A member's LmState.cur can move outside the batch — vio_gn_iteration, vio_solve, the stepwise flips on that one context —
    without its tables changing; the batch's cached device array (every window's cur at build time, flipped by the parity of the
    batch's own iteration count) must notice and be rebuilt.  Same bits as separate runs given the same call sequence."""
    ws = windows(vio, oracle_lib)[:4]
    lam = 5e5
    lead = hip_lib.context()
    batch = [lead] + [hip_lib.context(stream=lead.get_stream()) for _ in ws[1:]]
    solo = [hip_lib.context() for _ in ws]
    for c, r, w in zip(batch, solo, ws):
        c.load(w)
        r.load(w)

    def both(fn_batch, fn_solo):
        fn_batch()
        fn_solo()

    def batch_step():
        hip_lib.batch_gn_iteration(batch, lam)
        for r in solo:
            r.gn_iteration(lam)

    batch_step()
    # one solo iteration on EVERY member — all steps still pending, all generations equal — then the batch again
    for c, r in zip(batch, solo):
        c.gn_iteration(lam)
        r.gn_iteration(lam)
    batch_step()
    batch_step()
    # one member only, an odd number of solo iterations; another through the stepwise entry points; a third through vio_solve
    for _ in range(3):
        batch[1].gn_iteration(lam)
        solo[1].gn_iteration(lam)
    for c in (batch[2], solo[2]):
        c.linearize()
        c.init_lm()
        c.solve_linear(lam)
        c.update_states()
        c.eval_step()
    ra, rb = batch[3].solve(3), solo[3].solve(3)
    assert (ra.iterations, ra.trials, ra.final_chi2) == (rb.iterations, rb.trials, rb.final_chi2)
    batch_step()
    batch_step()
    for c, r in zip(batch, solo):
        for x, y in zip(state_of(c), state_of(r)):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("kind", ["invdepth", "xyz"])
def test_throughput_item_policy(vio, hip_lib, oracle_lib, kind):
    """vio_config.item_policy = VIO_ITEMS_THROUGHPUT: the largest items half the LDS holds, half-width workgroups (k_linearize_h /
    k_linearize_xyz_h), two to a CU — what a batch wants.  Another grouping of the same sums: batch and solo runs of such contexts
    agree bit for bit, and with the default policy to rounding."""
    xyz = kind == "xyz"
    make = vio.synth.make_window if kind == "invdepth" else vio.synth.make_window_xyz
    ws = [make(3000, seed=41), make(1200, seed=42, ragged=True)]
    lam = 5e5 if kind == "invdepth" else 2e5
    pol = vio.capi.ITEMS_THROUGHPUT
    lead = hip_lib.context(item_policy=pol)
    batch = [lead, hip_lib.context(stream=lead.get_stream(), item_policy=pol)]
    solo = [hip_lib.context(item_policy=pol) for _ in ws]
    dflt = [hip_lib.context() for _ in ws]
    for c, r, d, w in zip(batch, solo, dflt, ws):
        c.load(w)
        r.load(w)
        d.load(w)
    for _ in range(3):
        hip_lib.batch_gn_iteration(batch, lam)
        for r in solo + dflt:
            r.gn_iteration(lam)
    for c, r, d in zip(batch, solo, dflt):
        for x, y, z in zip(state_of(c, xyz), state_of(r, xyz), state_of(d, xyz)):
            np.testing.assert_array_equal(x, y)
            assert np.abs(np.asarray(x) - np.asarray(z)).max() <= 1e-9 * max(1.0, np.abs(np.asarray(z)).max())
    # the policy can be changed on a living context (the plan is rebuilt)
    dflt[0].set_config(item_policy=pol)
    dflt[0].load(ws[0])
    solo[0].load(ws[0])
    dflt[0].gn_iteration(lam)
    solo[0].gn_iteration(lam)
    for x, y in zip(state_of(dflt[0], xyz), state_of(solo[0], xyz)):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("policy", [0, 1])
def test_batch_with_a_free_extrinsic(vio, hip_lib, policy):
    """Windows whose plans carry an extrinsic block (vio_config.ext_fixed = 0) run on the kernels that read the block layout from the
    item (k_linearize_gb / k_linearize_ghb), also when only one window of the batch has one: bit-identical to separate runs."""
    ws = [vio.synth.make_window(900, seed=51, ragged=True), vio.synth.make_window(1500, seed=52)]
    lam = 5e5
    lead = hip_lib.context(item_policy=policy, ext_fixed=0)
    batch = [lead, hip_lib.context(stream=lead.get_stream(), item_policy=policy, ext_fixed=1)]
    solo = [hip_lib.context(item_policy=policy, ext_fixed=0), hip_lib.context(item_policy=policy, ext_fixed=1)]
    for c, r, w in zip(batch, solo, ws):
        c.load(w)
        r.load(w)
    for _ in range(3):
        hip_lib.batch_gn_iteration(batch, lam)
        for r in solo:
            r.gn_iteration(lam)
    for c, r in zip(batch, solo):
        for x, y in zip(state_of(c), state_of(r)):
            np.testing.assert_array_equal(x, y)
    reps_b = hip_lib.batch_solve(batch, 5)
    reps_s = [r.solve(5) for r in solo]
    for a, b in zip(reps_b, reps_s):
        assert (a.iterations, a.trials, a.final_chi2) == (b.iterations, b.trials, b.final_chi2)
    for c, r in zip(batch, solo):
        for x, y in zip(state_of(c), state_of(r)):
            np.testing.assert_array_equal(x, y)


def test_batched_gn_xyz_windows(vio, hip_lib):
    """XYZ-landmark windows in one batch (k_linearize_xyz_b): equal to separate runs bit for bit, ragged sizes included."""
    ws = [vio.synth.make_window_xyz(600, seed=11, ragged=True), vio.synth.make_window_xyz(2000, seed=12), vio.synth.make_window_xyz(40, seed=13),
          vio.synth.make_window_xyz(1, seed=14, obs_per_landmark=4)]
    lam = 2e5
    lead = hip_lib.context()
    batch = [lead] + [hip_lib.context(stream=lead.get_stream()) for _ in ws[1:]]
    solo = [hip_lib.context() for _ in ws]
    for c, r, w in zip(batch, solo, ws):
        c.load(w)
        r.load(w)
    for it in range(4):
        hip_lib.batch_gn_iteration(batch, lam)
        for r in solo:
            r.gn_iteration(lam)
        if it == 1:
            np.testing.assert_array_equal(batch[0].get_landmarks_xyz(), solo[0].get_landmarks_xyz())
    for c, r in zip(batch, solo):
        for x, y in zip(c.get_window(), r.get_window()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(c.get_landmarks_xyz(), r.get_landmarks_xyz())
        assert c.chi2() == r.chi2()


@pytest.mark.parametrize("kind", ["invdepth", "xyz"])
def test_batched_lm_solve_equals_separate_solves(vio, hip_lib, oracle_lib, kind):
    """vio_batch_solve: Problem::Solve of several windows with one launch per kernel for all of them.  Every window has its own
    LmState — windows that converge early stop while the others go on, rejected trials re-solve with their own lambda — and the
    result equals vio_solve on each, bit for bit (same kernel bodies, same data); reports included."""
    if kind == "xyz":
        ws = [vio.synth.make_window_xyz(500, seed=31, ragged=True), vio.synth.make_window_xyz(1500, seed=32), vio.synth.make_window_xyz(60, seed=33),
              vio.synth.make_window_xyz(8, seed=34, obs_per_landmark=4)]
        get = lambda c: c.get_landmarks_xyz()     # noqa: E731
        # a prior on the small windows of a batch with wider ones (its first-order update is spread over the window's own workgroups)
        ws[2].prior = ws[3].prior = windows(vio, oracle_lib)[2].prior
    else:
        ws = windows(vio, oracle_lib)
        ws.append(vio.synth.make_window(900, seed=8, ragged=True, outlier_fraction=0.1))       # rejected trials on the way
        get = lambda c: c.get_landmarks()         # noqa: E731
    lead = hip_lib.context()
    batch = [lead] + [hip_lib.context(stream=lead.get_stream()) for _ in ws[1:]]
    solo = [hip_lib.context() for _ in ws]
    for c, r, w in zip(batch, solo, ws):
        c.load(w)
        r.load(w)
    reps = hip_lib.batch_solve(batch, 40)
    for c, r, rb in zip(batch, solo, reps):
        rs = r.solve(40)
        assert (rb.iterations, rb.trials, rb.accepted, rb.stop_reason) == (rs.iterations, rs.trials, rs.accepted, rs.stop_reason)
        assert rb.final_chi2 == rs.final_chi2 and rb.final_lambda == rs.final_lambda and rb.initial_chi2 == rs.initial_chi2
        np.testing.assert_array_equal(np.array(rb.chi2_trace[:]), np.array(rs.chi2_trace[:]))
        for x, y in zip(c.get_window(), r.get_window()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(get(c), get(r))
    assert len({rb.iterations for rb in reps}) > 1 or len({rb.trials for rb in reps}) > 1       # the windows did not run in lockstep
    # the contexts go on as usual: a second batched solve, a single one, a marginalisation
    reps2 = hip_lib.batch_solve(batch[:3], 4)
    for c, r, rb in zip(batch[:3], solo[:3], reps2):
        rs = r.solve(4)
        assert rb.iterations == rs.iterations and rb.final_chi2 == rs.final_chi2
        np.testing.assert_array_equal(get(c), get(r))
    if kind != "xyz":
        ma, mb = batch[1].marginalize(vio.MARG_OLD), solo[1].marginalize(vio.MARG_OLD)
        np.testing.assert_array_equal(ma["H"], mb["H"])


def test_batch_argument_checks(vio, hip_lib):
    a, b = hip_lib.context(), hip_lib.context()            # two streams
    w = vio.synth.make_window(50, seed=1)
    a.load(w)
    b.load(w)
    with pytest.raises(vio.VioError):
        hip_lib.batch_gn_iteration([a, b], 1e3)
    c = hip_lib.context(stream=a.get_stream())
    c.load(vio.synth.make_window_xyz(50, seed=1))
    with pytest.raises(vio.VioError):
        hip_lib.batch_gn_iteration([a, c], 1e3)            # one kind of landmark per batch
    d = hip_lib.context(stream=a.get_stream())
    d.load(w)
    with pytest.raises(vio.VioError) as ei:
        hip_lib.batch_gn_iteration([a, d, a], 1e3)         # the same context twice: two windows of the grid on one set of buffers
    assert "twice" in str(ei.value)
    with pytest.raises(vio.VioError):
        hip_lib.batch_solve([a, d, d], 3)
    # a member's failure is reported where the caller looks: on the leader
    e = hip_lib.context(stream=a.get_stream())
    e.set_window(w.poses, w.speed_bias, w.ext)
    e.set_landmarks(np.array([0.2, 0.3]))
    e.set_observations([0], [0], [1], [[0.0, 0.0]], [[0.01, 0.0]])      # landmark 1 has no observation: its plan cannot be built
    with pytest.raises(vio.VioError) as ei:
        hip_lib.batch_gn_iteration([a, e], 1e3)
    assert "window 1" in str(ei.value)
