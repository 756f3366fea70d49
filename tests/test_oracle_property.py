"""Property tests (hypothesis): the two independent restatements of the reference path agree on random problems of
random shape, and the C oracle's block-COO Jacobian equals complex-step derivatives of the numpy restatement knot
by knot, with the jump mask (quirk Q1) applied."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import np_oracle as NP
from oracle import oracle as O


@st.composite
def problems(draw):
    N = draw(st.integers(2, 14))
    kt = draw(st.integers(1, N + 1))
    im = draw(st.sampled_from([1, 2]))
    seed = draw(st.integers(0, 2**31 - 1))
    return N, kt, im, seed


@settings(max_examples=40, deadline=None)
@given(problems())
def test_restatements_agree_and_jacobian_is_the_derivative(p):
    N, kt, im, seed = p
    rng = np.random.default_rng(seed)
    x0, xf = rng.normal(size=15), rng.normal(size=15)
    cost = rng.normal(size=(N, 41))
    Z = rng.normal(size=20 * N - 5)
    Z[15:20 * (N - 1):20] *= 40.0  # forces of realistic size
    Z[16:20 * (N - 1):20] *= 40.0
    Z[19::20] = rng.uniform(0.001, 0.02, size=N - 1)
    nlp = O.OracleNLP(N, kt, im, x0, xf, cost)
    c = nlp.eval_c(Z)
    cn = NP.eval_c(N, kt, im, x0, xf, Z)
    assert c.shape == cn.shape == (18 * N - kt + 16,)
    assert np.max(np.abs(c - cn)) <= 1e-12 * max(1.0, np.max(np.abs(cn)))
    assert abs(nlp.eval_f(Z) - NP.eval_f(N, cost, Z)) <= 1e-10 * max(1.0, abs(nlp.eval_f(Z)))
    vals = nlp.jac_c_coo(Z)
    mode, jump = NP.knot_modes(N, kt, im)
    for k in range(N - 1):
        J = vals[300 * k : 300 * (k + 1)].reshape(20, 15).T
        Jc = NP.step_jacobian_complex(int(mode[k]), Z[20 * k : 20 * k + 15], Z[20 * k + 15 : 20 * k + 20])
        if jump[k]:
            Jc = NP.JUMP_DIAG[:, None] * Jc
        assert np.max(np.abs(J - Jc)) <= 1e-11 * max(1.0, np.max(np.abs(Jc)))
    rows, cols = nlp.jac_structure()
    assert rows.min() >= 0 and rows.max() < nlp.m_nlp and cols.min() >= 0 and cols.max() < nlp.n_nlp
    assert len(set(zip(rows.tolist(), cols.tolist()))) == nlp.nnz
