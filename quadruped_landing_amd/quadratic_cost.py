"""Host-side cost setup: QuadraticCost / LQRCost (mirror of src/quadratic_cost.jl:16-42).

A cost is stored as the 41-double record the kernels read:
``[Q diag (15) | R diag (5) | q (15) | r (5) | c]``.  Only setup lives here; the
evaluation (stagecost/termcost, src/quadratic_cost.jl:44-52) runs on the GPU.
"""
from __future__ import annotations

import numpy as np

COST_STRIDE = 41


def _seq_quad(D, v):
    """0.5*v'D*v with the reference's left-to-right summation, vectorised over leading dims."""
    acc = (0.5 * (D[..., 0] * v[..., 0])) * v[..., 0]
    for i in range(1, v.shape[-1]):
        acc = acc + (0.5 * (D[..., i] * v[..., i])) * v[..., i]
    return acc


def LQRCost(Q, R, xf, uf=None):
    """LQRCost(Q, R, xf, uf) -> 41-double record(s).  Q, R: diagonals (15,), (5,) or full matrices.
    xf: (...,15), uf: (...,5) broadcast over leading dims."""
    Q = np.asarray(Q, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64)
    Qd = np.diag(Q) if Q.ndim == 2 else Q
    Rd = np.diag(R) if R.ndim == 2 else R
    xf = np.asarray(xf, dtype=np.float64)
    uf = np.zeros(xf.shape[:-1] + (5,)) if uf is None else np.asarray(uf, dtype=np.float64)
    lead = np.broadcast_shapes(xf.shape[:-1], uf.shape[:-1])
    out = np.empty(lead + (COST_STRIDE,))
    out[..., 0:15] = Qd
    out[..., 15:20] = Rd
    out[..., 20:35] = (-Qd) * xf          # q = -Q * xf
    out[..., 35:40] = (-Rd) * uf          # r = -R * uf
    Qb = np.broadcast_to(Qd, lead + (15,))
    Rb = np.broadcast_to(Rd, lead + (5,))
    out[..., 40] = _seq_quad(Qb, np.broadcast_to(xf, lead + (15,))) + _seq_quad(Rb, np.broadcast_to(uf, lead + (5,)))
    return out


def lqr_objective(Q, R, Qf, Xref, Uref):
    """The notebook's objective (src/main.ipynb:158-161): obj[k] = LQRCost(Q,R,Xref[k],Uref[k]) for
    k < N and obj[N] = LQRCost(Qf, R*0, Xref[N], Uref[1]).  Xref: (...,N,15), Uref: (...,N-1,5)."""
    Xref = np.asarray(Xref, dtype=np.float64)
    Uref = np.asarray(Uref, dtype=np.float64)
    N = Xref.shape[-2]
    tab = np.empty(Xref.shape[:-2] + (N, COST_STRIDE))
    tab[..., : N - 1, :] = LQRCost(Q, R, Xref[..., : N - 1, :], Uref)
    R = np.asarray(R, dtype=np.float64)
    tab[..., N - 1, :] = LQRCost(Qf, R * 0, Xref[..., N - 1, :], Uref[..., 0, :])
    return tab
