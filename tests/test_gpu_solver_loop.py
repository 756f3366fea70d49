"""The evaluator inside a solver loop: scipy's trust-constr (Ipopt is not installed) drives the four MOI callbacks; the
iterates obtained with the GPU evaluator and with the CPU oracle behind the same callback surface coincide."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


class _OracleProblem:
    """The oracle behind the HybridNLP host-mode surface the callbacks use."""

    def __init__(self, nb):
        from oracle import oracle as O
        from tests.helpers import oracle_model

        self.o = O.OracleNLP(nb.N, int(nb.k_trans[0]), int(nb.init_mode[0]), nb.x0[0], nb.xf[0], nb.obj, oracle_model(nb.model))
        self.B, self.n_nlp, self.z_stride = 1, self.o.n_nlp, self.o.n_nlp

    def num_duals(self, b=0):
        return self.o.m_nlp

    def num_primals(self):
        return self.o.n_nlp

    def cinds(self, b=0):
        return self.o.cinds()

    def eval_f_host(self, x):
        return np.array([self.o.eval_f(x)])

    def grad_f_host(self, x):
        return self.o.grad_f(x)

    def eval_c_host(self, x):
        return self.o.eval_c(x)

    def jac_c_dense_host(self, x, jac, b=0):
        D = self.o.jac_c_dense(x)
        ok = ~np.isnan(D)
        jac[ok] = D[ok]
        return jac


def test_parity_along_a_solver_trajectory():
    """Drive scipy's trust-constr with the GPU callbacks, then re-evaluate every iterate it visited with the oracle:
    objective, gradient, constraints and the dense Jacobian agree at each of them (the iterates themselves cannot be
    compared between two solves: the landing NLP is degenerate -- scipy reports a singular Jacobian, the reference's
    own Ipopt run ended in "Restoration Failed", src/main.ipynb:727 -- so 1e-11 differences change the path)."""
    import quadruped_landing_amd as Q
    from quadruped_landing_amd import moi, nlp as NLP, problem_gen as PG
    from solve_with_scipy import run
    from tests.helpers import rel_err

    N, kt, iters = 9, 4, 10
    nb = PG.notebook_problem(N=N, k_trans=kt)
    x_l, x_u = NLP.variable_bounds_forces(N)
    gpu = Q.HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    res, trace = run(gpu, moi, nb.Z[0], x_l, x_u, iters)
    assert len(trace) >= 5 and not np.array_equal(trace[-1], nb.Z[0])
    orc = _OracleProblem(nb)
    m, n = gpu.num_duals(), gpu.num_primals()
    for x in trace:
        assert abs(moi.eval_objective(gpu, x) - orc.o.eval_f(x)) <= 1e-8 * max(1.0, abs(orc.o.eval_f(x)))
        g = np.zeros(n)
        moi.eval_objective_gradient(gpu, g, x)
        assert rel_err(g, orc.o.grad_f(x), floor=1e-12) <= 1e-8
        c = np.zeros(m)
        moi.eval_constraint(gpu, c, x)
        assert rel_err(c, orc.o.eval_c(x), floor=1.0) <= 1e-8
        vec = np.full(m * n, np.nan)
        moi.eval_constraint_jacobian(gpu, vec, x)
        D, Dref = vec.reshape((m, n), order="F"), orc.o.jac_c_dense(x)
        assert np.array_equal(np.isnan(D), np.isnan(Dref))
        assert rel_err(D, Dref, floor=1e-300) <= 1e-8
