"""Known-answer tests the reference itself publishes for this path."""
import ctypes as C

import numpy as np

dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def test_marginalize_kat_from_the_reference_readme(oracle_lib):
    """A/15-vio-backend/backend/problem.cc:571-659 (TestMarginalize): information matrix of three chained
    variables with sigma 0.1/0.2/0.3; marginalising variable 1 must print
        26.5306 -8.1633 / -8.1633 10.2041      (A/15-vio-backend/README.md:80-90)."""
    d1, d2, d3 = 0.1 * 0.1, 0.2 * 0.2, 0.3 * 0.3
    H = np.array([[1 / d1, -1 / d1, 0], [-1 / d1, 1 / d1 + 1 / d2 + 1 / d3, -1 / d3], [0, -1 / d3, 1 / d3]])
    # "move row/col 1 to the bottom right" (the reference swaps variable 1 and 2)
    P = [0, 2, 1]
    Hm = np.ascontiguousarray(H[np.ix_(P, P)])
    np.testing.assert_allclose(Hm, [[100, 0, -100], [0, 11.1111, -11.1111], [-100, -11.1111, 136.1111]], atol=6e-5)
    out = np.zeros((2, 2))
    f = oracle_lib.dll.vioo_schur_pinv
    f.restype = None
    f(C.c_int(3), C.c_int(1), dp(Hm), None, dp(out), None)
    np.testing.assert_allclose(out, [[26.5306, -8.1633], [-8.1633, 10.2041]], atol=5e-5)
    # closed form: the chain 1/(d1+d2) and 1/d3 coupling
    np.testing.assert_allclose(out, H[np.ix_([0, 2], [0, 2])] - np.outer(H[[0, 2], 1], H[1, [0, 2]]) / H[1, 1], rtol=1e-13)


def test_lm_lambda_schedule_kat(oracle_lib, vio):
    """Nielsen schedule of IsGoodStepInLM (problem.cc:559-567): an accepted step scales lambda by
    max(1/3, min(2/3, 1-(2 rho-1)^3)); a rejected one multiplies by ni and doubles ni.  The reference's own
    curve-fitting logs show exactly these factors (A/13-vio-bundle-adjustment/doc/data/*/curve_fitting_LM_log__nielsen.csv:
    x1/3 .. x2/3 per accepted step)."""
    w = vio.synth.make_window(40, seed=8)
    ctx = oracle_lib.context()
    ctx.load(w)
    ctx.linearize()
    chi0, lam0 = ctx.init_lm()
    assert abs(lam0 - 1e-5 * 5e10) < 1e-6          # capped max diagonal (problem.cc:518-520)
    ctx.solve_linear(lam0)
    ctx.update_states()
    ok, chi1, lam1 = ctx.eval_step()
    assert ok and chi1 < chi0
    assert 1 / 3 - 1e-12 <= lam1 / lam0 <= 2 / 3 + 1e-12
    # force a rejection with an absurd step: solve with a negative damping that flips the step direction
    ctx.linearize()
    ctx.solve_linear(-1e18)
    ctx.update_states()
    ok2, chi2, lam2 = ctx.eval_step()
    assert not ok2 and chi2 == chi1
    assert lam2 == -1e18 * 2.0                      # currentLambda_ *= ni_ with ni_ = 2
