"""The reference's own marginalisation KAT (Problem::TestMarginalize, A/15-vio-backend/backend/problem.cc:571-650; its printout
is published in A/15-vio-backend/README.md:80-90): three scalar variables in a chain with variances 0.1^2, 0.2^2, 0.3^2,

    H = [ 100  -100      0     ]        marginalise the middle one:      [ 26.5306  -8.1633 ]
        [-100   136.1111 -11.1111]                                       [ -8.1633  10.2041 ]
        [ 0    -11.1111   11.1111]

through the C ABI.  MargNewFrame (estimator.cpp:830-901) marginalises pose 9 + speed-bias 9 out of the prior alone, so the KAT is
a prior whose only entries are those nine: the middle variable on a coordinate of frame 9, the other two on coordinates of frames
1 and 2.  What comes back is Problem::Marginalize's whole dense path (problem.cc:717-779: permutation to the bottom right,
pseudo-inverse of the marginalised block through its eigen-decomposition, Schur complement, eigen-decomposition of the result,
H_prior rebuilt as J^T J, small entries zeroed) on that input."""
import numpy as np
import pytest

D1, D2, D3 = 0.1 * 0.1, 0.2 * 0.2, 0.3 * 0.3
IDX_MID, IDX_A, IDX_B = 6 + 15 * 9 + 2, 6 + 15 * 1 + 0, 6 + 15 * 2 + 4        # frame 9 (marginalised), frames 1 and 2 (kept)


def kat_prior():
    H = np.zeros((156, 156))
    a, m, b = IDX_A, IDX_MID, IDX_B
    H[a, a] = 1 / D1
    H[a, m] = H[m, a] = -1 / D1
    H[m, m] = 1 / D1 + 1 / D2 + 1 / D3
    H[m, b] = H[b, m] = -1 / D3
    H[b, b] = 1 / D3
    return dict(H=H, b=np.zeros(156), err=np.zeros(156), jt_inv=np.zeros((156, 156)))


def check(vio, lib):
    w = vio.synth.make_window(8, seed=3)
    w.prior = kat_prior()
    c = lib.context()
    c.load(w)
    out = c.marginalize(vio.MARG_SECOND_NEW)
    H = out["H"]
    mm = 1 / D1 + 1 / D2 + 1 / D3
    want = np.array([[1 / D1 - (1 / D1) ** 2 / mm, -(1 / D1) * (1 / D3) / mm], [-(1 / D1) * (1 / D3) / mm, 1 / D3 - (1 / D3) ** 2 / mm]])
    got = H[np.ix_([IDX_A, IDX_B], [IDX_A, IDX_B])]
    assert np.abs(want - np.array([[26.5306, -8.1633], [-8.1633, 10.2041]])).max() <= 5.1e-5      # the README's printout
    assert np.abs(got - want).max() <= 1e-10
    rest = H.copy()
    rest[np.ix_([IDX_A, IDX_B], [IDX_A, IDX_B])] = 0.0
    assert np.abs(rest).max() == 0.0                              # nothing else: entries below 1e-9 are zeroed (problem.cc:778)
    assert np.abs(out["b"]).max() == 0.0 and np.abs(out["err"]).max() == 0.0
    # J^T J = H: jt_inv is the pseudo-inverse's factor, S^-1/2 V^T — on the two kept directions (J^T)^-1 (H) (J)^-1 = I
    J = out["jt_inv"]
    P = J @ H @ J.T
    assert abs(np.trace(P) - 2.0) <= 1e-9


def test_oracle_reproduces_the_references_marginalisation_kat(vio, oracle_lib):
    check(vio, oracle_lib)


@pytest.mark.ref
def test_compiled_reference_reproduces_it(vio, ref_lib):
    check(vio, ref_lib)


@pytest.mark.gpu
def test_hip_reproduces_the_references_marginalisation_kat(vio, hip_lib):
    check(vio, hip_lib)
