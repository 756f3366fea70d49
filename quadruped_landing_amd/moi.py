"""MOI.AbstractNLPEvaluator callback surface (mirror of src/moi.jl:1-33).

Same method names and argument order as the reference's MOI methods on HybridNLP; `prob` is a
quadruped_landing_amd.nlp.HybridNLP.  Host arrays in, host arrays filled in place -- the shape
Ipopt's callbacks have (src/moi.jl:1-24) -- with the arithmetic done by the HIP kernels.
The Julia veneer with the identical surface is integration/julia/HybridNLPHIP.jl.
"""
from __future__ import annotations

import numpy as np


def eval_objective(prob, x):
    """MOI.eval_objective, src/moi.jl:1-3.  Scalar for B == 1, else (B,) array."""
    f = prob.eval_f_host(x)
    return float(f[0]) if prob.B == 1 else f


def eval_objective_gradient(prob, grad_f, x):
    """MOI.eval_objective_gradient, src/moi.jl:5-8 (in place)."""
    g = prob.grad_f_host(x)
    np.asarray(grad_f).reshape(-1)[:] = g.reshape(prob.B, prob.z_stride)[:, : prob.n_nlp].reshape(-1) \
        if np.asarray(grad_f).size == prob.B * prob.n_nlp else g
    return None


def eval_constraint(prob, g, x):
    """MOI.eval_constraint, src/moi.jl:10-13 (in place)."""
    c = prob.eval_c_host(x)
    out = np.asarray(g).reshape(-1)
    if prob.B == 1:
        out[:] = c[: out.size]
    else:
        out[:] = c
    return None


def eval_constraint_jacobian(prob, vec, x, b: int = 0):
    """MOI.eval_constraint_jacobian, src/moi.jl:15-24: `vec` is a flat length m_nlp*n_nlp buffer,
    reshaped column-major (m_nlp, n_nlp); only the jac_c! write-set is assigned."""
    m_nlp, n_nlp = prob.num_duals(b), prob.num_primals()
    jac = np.asarray(vec).reshape((m_nlp, n_nlp), order="F")
    if not np.shares_memory(jac, vec):
        raise ValueError("vec must be a contiguous float64 buffer")
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if x.size == n_nlp:                     # problem b's own decision vector
        xb = x
    elif x.size == prob.B * prob.z_stride:  # the whole batch in the handle's layout: take problem b's slice
        xb = x.reshape(prob.B, prob.z_stride)[b, :n_nlp]
    elif x.size == prob.B * n_nlp:
        xb = x.reshape(prob.B, n_nlp)[b]
    else:
        raise ValueError(f"x has {x.size} entries; expected n_nlp = {n_nlp} or the whole batch")
    prob.jac_c_dense_host(xb, jac, b)
    return None


def features_available(prob):
    """src/moi.jl:26-28"""
    return ["Grad", "Jac"]


def initialize(prob, features):
    """src/moi.jl:30"""
    return None


def jacobian_structure(prob, b: int = 0):
    """src/moi.jl:31-33: all (row, col) pairs of the dense m_nlp x n_nlp matrix, row index fastest,
    1-based like the reference."""
    m_nlp, n_nlp = prob.num_duals(b), prob.num_primals()
    cols, rows = np.divmod(np.arange(m_nlp * n_nlp), m_nlp)
    return list(zip((rows + 1).tolist(), (cols + 1).tolist()))


def sparse_jacobian_structure(prob, b: int = 0):
    """The write-set of jac_c! as 1-based (row, col) pairs in the order of the block-COO values --
    what `use_sparse_jacobian=true` (src/nlp.jl:35,75-76) was meant to hand to Ipopt."""
    rows, cols = prob.jacobian_structure(b)
    return list(zip((rows + 1).tolist(), (cols + 1).tolist()))
