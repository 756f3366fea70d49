"""Jacobian VALUES have no shipped known answer ("parity unpinned"): they are checked as derivatives
of the pinned constraint function (complex-step on an independent numpy restatement, central
differences on the C oracle itself) and for the reference's quirks Q1/Q3/Q5 by source reading."""
import numpy as np
import pytest

from oracle import np_oracle as NP
from oracle import oracle as O

PATTERN_MODE1 = [  # SURVEY.md 8.0, rows r1..r15, columns x1..x15,u1..u5
    "x......x.......x.x.x", ".x......x.......x.xx", "xxxxxxxxxx..xx.xxxxx", "...x................",
    "....x...............", ".....x......x....x.x", "......x......x....xx", ".......x.......x.x.x",
    "........x.......x.xx", "xx.xxxxxxx..xx.xxxxx", "..........x.........", "...........x........",
    "............x....x.x", ".............x....xx", "..............x....x"]


def _rand_xu(rng):
    x = rng.normal(size=15)
    u = np.concatenate([rng.normal(size=4) * 60.0, [rng.uniform(0.001, 0.02)]])
    return x, u


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_step_jacobian_vs_complex_step(mode):
    rng = np.random.default_rng(mode)
    for _ in range(20):
        x, u = _rand_xu(rng)
        J = O.contact_jacobian(mode, x, u)
        Jc = NP.step_jacobian_complex(mode, x, u)
        nz = Jc != 0
        assert np.array_equal(J != 0, nz)
        assert np.max(np.abs(J[nz] - Jc[nz]) / np.abs(Jc[nz])) <= 1e-12


def test_structural_sparsity_counts():
    rng = np.random.default_rng(0)
    x, u = _rand_xu(rng)
    nnz = {m: int(np.count_nonzero(O.contact_jacobian(m, x, u))) for m in (1, 2, 3)}
    assert nnz == {1: 71, 2: 71, 3: 57}
    pat = np.array([[ch == "x" for ch in row] for row in PATTERN_MODE1])
    assert np.array_equal(O.contact_jacobian(1, x, u) != 0, pat)


def test_golden_single_knot_vectors(golden_dir):
    import os

    g = np.load(os.path.join(golden_dir, "single_knot.npz"))
    for x, u, mode, xn, J in zip(g["x"], g["u"], g["mode"], g["xn"], g["J"]):
        assert np.array_equal(O.contact_dynamics_rk4(int(mode), x, u), xn)
        assert np.array_equal(O.contact_jacobian(int(mode), x, u), J)
        assert np.max(np.abs(NP.rk4(int(mode), x, u) - xn)) <= 1e-13


def _small_problem(N=9, k_trans=4, init_mode=1, seed=0):
    rng = np.random.default_rng(seed)
    Xref, Uref = O.reference_trajectory(N, k_trans, np.zeros(15), init_mode, 0.009)
    cost = O.lqr_cost_table(np.array([10.0] * 14 + [0.0]), np.array([1e-3, 1e-2, 1e-3, 1e-2, 0.0]),
                            np.array([10.0] * 14 + [0.0]), Xref, Uref)
    nlp = O.OracleNLP(N, k_trans, init_mode, rng.normal(size=15), rng.normal(size=15), cost)
    Z = rng.normal(size=nlp.n_nlp)
    Z[19::20] = rng.uniform(0.001, 0.02, size=N - 1)
    return nlp, Z


@pytest.mark.parametrize("init_mode", [1, 2])
def test_dense_jacobian_is_the_derivative_of_eval_c_except_quirk_Q1(init_mode):
    nlp, Z = _small_problem(init_mode=init_mode)
    D = nlp.jac_c_dense(Z, fill=0.0)
    eps = 1e-6
    fd = np.zeros_like(D)
    for j in range(nlp.n_nlp):
        e = np.zeros(nlp.n_nlp)
        e[j] = eps
        fd[:, j] = (nlp.eval_c(Z + e) - nlp.eval_c(Z - e)) / (2 * eps)
    diff = np.abs(D - fd)
    # Q1: the jump mask zeroes row 15 of the block at knot k_trans-1 although the clock passes through
    ci = nlp.cinds()
    r = ci[2][0] - 1 + 15 * (nlp.k_trans - 2) + 14
    cols = [20 * (nlp.k_trans - 2) + 14, 20 * (nlp.k_trans - 2) + 19]
    assert np.allclose(fd[r, cols], 1.0, atol=1e-7) and np.all(D[r, cols] == 0)
    diff[r, cols] = 0
    assert diff.max() <= 5e-6 * max(1.0, np.abs(fd).max())


def test_quirk_Q3_clearance_branch_at_theta_zero():
    nlp, Z = _small_problem()
    Z[2] = 0.0       # theta_1 == 0 takes the "+" branch (src/constraints.jl:269-273)
    Z[22] = 0.3
    Z[42] = -0.3
    v = nlp.jac_c_coo(Z)
    th = v[300 * (nlp.N - 1) : 300 * (nlp.N - 1) + nlp.N]
    assert th[0] == 0.25 * np.cos(0.0)
    assert th[1] == -0.25 * np.cos(0.3) and th[2] == 0.25 * np.cos(-0.3)


def test_quirk_Q5_write_set_and_coo_consistency():
    nlp, Z = _small_problem(N=11, k_trans=5)
    D = nlp.jac_c_dense(Z)  # NaN where jac_c! never assigns
    N, kt = nlp.N, nlp.k_trans
    assert np.count_nonzero(~np.isnan(D)) == 435 + 525 * (N - 1) + 4 * N - kt + 3
    rows, cols = nlp.jac_structure()
    vals = nlp.jac_c_coo(Z)
    assert len(set(zip(rows.tolist(), cols.tolist()))) == nlp.nnz  # no duplicate positions
    assert not np.isnan(D[rows, cols]).any()
    assert np.array_equal(D[rows, cols], vals)
    rest = ~np.isnan(D)
    rest[rows, cols] = False  # what the COO leaves out of the write-set: explicit zeros of the -I(n) blocks
    assert np.count_nonzero(rest) == 210 * (N - 1) and not D[rest].any()


def test_quirk_Q2_gradient_has_no_dh_term():
    nlp, Z = _small_problem()
    g = nlp.grad_f(Z)
    k = 2
    x, u = Z[20 * k : 20 * k + 15], Z[20 * k + 15 : 20 * k + 20]
    cost = nlp.cost[k]
    assert np.array_equal(g[20 * k : 20 * k + 15], u[4] * (cost[:15] * x + cost[20:35]))
    assert g[20 * k + 19] == u[4] * (cost[19] * u[4] + cost[39])  # = 0 for R55 = 0; no l_k term
