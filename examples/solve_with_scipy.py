"""Drop-in demonstration: a generic NLP solver driving the evaluator through the MOI callback surface.

The reference hands its HybridNLP to Ipopt through MOI (src/moi.jl:46-103).  Ipopt is not installed here, so this
example uses scipy's trust-constr (an interior-point / SQP method that tolerates infeasible starts) as the solver; what matters is the shape of the interaction: the solver owns the host
buffers and calls eval_objective / eval_objective_gradient / eval_constraint / eval_constraint_jacobian, exactly
the callbacks of src/moi.jl:1-24, and every evaluation runs on the GPU behind the C ABI.

The landing NLP itself is degenerate (redundant contact / dynamics rows; the reference's own Ipopt run ends in
"Restoration Failed", src/main.ipynb:727), so do not expect convergence from a generic solver -- the point is the
callback traffic, which tests/test_gpu_solver_loop.py checks against the oracle at every iterate.

    python examples/solve_with_scipy.py [N=9] [k_trans=4] [iters=20]
"""
import os
import sys

import numpy as np
from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_callbacks(prob, moi):
    """prob: anything with the HybridNLP host-mode surface; returns the four scipy callables."""
    m, n = prob.num_duals(), prob.num_primals()
    neq = prob.cinds()[5][1]  # rows 1..neq are equalities, the rest clearance inequalities (c >= 0)

    def f(x):
        return moi.eval_objective(prob, x)

    def g(x):
        out = np.zeros(n)
        moi.eval_objective_gradient(prob, out, x)
        return out

    def c(x):
        out = np.zeros(m)
        moi.eval_constraint(prob, out, x)
        return out

    def J(x):
        vec = np.zeros(m * n)  # the solver's dense buffer; only the write-set is assigned, the rest stays 0
        moi.eval_constraint_jacobian(prob, vec, x)
        return vec.reshape((m, n), order="F")

    lb = np.zeros(m)
    ub = np.concatenate([np.zeros(neq), np.full(m - neq, np.inf)])  # nlp.lb / nlp.ub, src/nlp.jl:66-69
    cons = NonlinearConstraint(c, lb, ub, jac=J)
    return f, g, cons


def run(prob, moi, Z0, x_l, x_u, iters):
    f, g, cons = make_callbacks(prob, moi)
    trace = []
    res = minimize(f, Z0, jac=g, hess=BFGS(), constraints=[cons], bounds=Bounds(x_l, x_u, keep_feasible=False),
                   method="trust-constr", options={"maxiter": iters, "gtol": 1e-10, "xtol": 1e-14},
                   callback=lambda xk, state: trace.append(np.array(xk)) and False)
    return res, trace


def main():
    import quadruped_landing_amd as Q
    from quadruped_landing_amd import moi, nlp as NLP, problem_gen as PG

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    kt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    nb = PG.notebook_problem(N=N, k_trans=kt)
    prob = Q.HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    x_l, x_u = NLP.variable_bounds_forces(N)
    res, trace = run(prob, moi, nb.Z[0], x_l, x_u, iters)
    c = np.zeros(prob.num_duals())
    moi.eval_constraint(prob, c, res.x)
    neq = prob.cinds()[5][1]
    print(f"N={N} k_trans={kt}: {res.nit} trust-constr iterations, f = {res.fun:.6f}, "
          f"max |c_eq| {np.abs(c[:neq]).max():.2e} (initial guess: "
          f"{np.abs(moi_c(prob, moi, nb.Z[0])[:neq]).max():.2e}), min clearance {c[neq:].min():.3f}")


def moi_c(prob, moi, x):
    out = np.zeros(prob.num_duals())
    moi.eval_constraint(prob, out, x)
    return out


if __name__ == "__main__":
    main()
