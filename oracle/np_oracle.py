"""Second, independent restatement of the reference path in numpy.

TEST INFRASTRUCTURE ONLY.  Written separately from qln_oracle.c (vectorised over
knots, dtype-generic) so the two restatements can be checked against each other,
and so the step Jacobian can be cross-checked by complex-step differentiation:
the RK4 step (src/planar_quadruped.jl:189-221) is a polynomial in [x;u], so
Im f(z + i*eps*e_j)/eps is its exact derivative up to rounding.
"""
from __future__ import annotations

import numpy as np

G, MB, MF, LB = -9.81, 10.0, 0.1, 0.5  # src/planar_quadruped.jl:11-20
IB = MB * LB**2 / 12


def dynamics(mode, s, u):
    """src/planar_quadruped.jl:36-185; s: (...,14), u: (...,5) -> (...,14)."""
    s = np.asarray(s)
    u = np.asarray(u)
    out = np.zeros(s.shape, dtype=np.result_type(s.dtype, u.dtype))
    F1x, F1y, F2x, F2y = u[..., 0], u[..., 1], u[..., 2], u[..., 3]
    out[..., 0:3] = s[..., 7:10]
    if mode == 2:
        out[..., 3:5] = s[..., 10:12]
        out[..., 10] = -F1x / MF
        out[..., 11] = -F1y / MF + G
    if mode == 1:
        out[..., 5:7] = s[..., 12:14]
        out[..., 12] = -F2x / MF
        out[..., 13] = -F2y / MF + G
    out[..., 7] = (F1x + F2x) / MB
    out[..., 8] = (F1y + F2y) / MB + G
    tau = -F1x * (s[..., 4] - s[..., 1]) + F1y * (s[..., 3] - s[..., 0]) - F2x * (s[..., 6] - s[..., 1]) + F2y * (s[..., 5] - s[..., 0])
    out[..., 9] = tau / IB
    return out


def rk4(mode, x, u):
    """src/planar_quadruped.jl:189-221; x: (...,15), u: (...,5) -> (...,15)."""
    x = np.asarray(x)
    u = np.asarray(u)
    h = u[..., 4:5]
    s = x[..., :14]
    f1 = dynamics(mode, s, u)
    f2 = dynamics(mode, s + 0.5 * h * f1, u)
    f3 = dynamics(mode, s + 0.5 * h * f2, u)
    f4 = dynamics(mode, s + h * f3, u)
    sn = s + (h / 6.0) * (f1 + 2 * f2 + 2 * f3 + f4)
    return np.concatenate([sn, x[..., 14:15] + u[..., 4:5]], axis=-1)


def jump_map(x):
    """src/planar_quadruped.jl:250-260."""
    xn = np.array(x, copy=True)
    xn[..., [4, 6, 10, 11, 12, 13]] = 0.0
    return xn


JUMP_DIAG = np.array([1, 1, 1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0], dtype=np.float64)  # :262-263


def step_jacobian_complex(mode, x, u, eps=1e-30):
    """15x20 Jacobian of the RK4 step by complex-step differentiation."""
    z = np.concatenate([np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)])
    J = np.zeros((15, 20))
    for j in range(20):
        zc = z.astype(np.complex128)
        zc[j] += 1j * eps
        J[:, j] = rk4(mode, zc[:15], zc[15:]).imag / eps
    return J


def knot_modes(N, k_trans, init_mode):
    """src/constraints.jl:23-37: (mode, jump) for 1-based dynamics knots 1..N-1."""
    K = np.arange(1, N)
    mode = np.where(K <= k_trans - 1, init_mode, 3)
    jump = K == k_trans - 1
    return mode, jump


def eval_c(N, k_trans, init_mode, x0, xf, Z):
    """src/constraints.jl:145-158, all seven groups concatenated in cinds order."""
    Z = np.asarray(Z, dtype=np.float64)
    X = np.stack([Z[20 * k : 20 * k + 15] for k in range(N)])
    U = np.stack([Z[20 * k + 15 : 20 * k + 20] for k in range(N - 1)])
    mode, jump = knot_modes(N, k_trans, init_mode)
    d = np.zeros((N - 1, 15))
    for m in (1, 2, 3):
        sel = mode == m
        if sel.any():
            d[sel] = rk4(m, X[:-1][sel], U[sel])
    d[jump] = jump_map(d[jump])
    d = d - X[1:]
    a, b = (4, 6) if init_mode == 1 else (6, 4)
    return np.concatenate([
        X[0] - x0,
        X[-1][:14] - np.asarray(xf)[:14],
        d.reshape(-1),
        X[:, a],
        X[k_trans - 1 :, b],
        [U[-1][1] + U[-1][3] + MB * G],
        X[:, 1] - LB / 2 * np.abs(np.sin(X[:, 2])),
    ])


def stagecost(cost, x, u):
    Q, R, q, r, c = cost[:15], cost[15:20], cost[20:35], cost[35:40], cost[40]
    return 0.5 * np.sum(Q * x * x) + q @ x + 0.5 * np.sum(R * u * u) + r @ u + c


def eval_f(N, cost, Z):
    """src/costs.jl:6-16 (summation order is numpy's, so compare with a tolerance)."""
    J = 0.0
    for k in range(N - 1):
        x, u = Z[20 * k : 20 * k + 15], Z[20 * k + 15 : 20 * k + 20]
        J += u[4] * stagecost(cost[k], x, u)
    xN = Z[20 * (N - 1) :]
    c = cost[N - 1]
    return J + 0.5 * np.sum(c[:15] * xN * xN) + c[20:35] @ xN + c[40]
