"""reference_trajectory (mirror of src/ref_traj.jl:6-39), vectorised over a batch."""
from __future__ import annotations

import numpy as np

from .planar_quadruped import PlanarQuadruped


def reference_trajectory(model: PlanarQuadruped, N, k_trans, xterm, init_mode, dt):
    """Returns (Xref, Uref).  Scalars k_trans/init_mode and xterm (15,) give (N,15), (N-1,5);
    arrays of shape (B,) / (B,15) give (B,N,15), (B,N-1,5)."""
    g, mb = model.g, model.mb
    k_trans = np.asarray(k_trans)
    batched = k_trans.ndim > 0
    kt = np.atleast_1d(k_trans).astype(np.int64)
    im = np.broadcast_to(np.atleast_1d(np.asarray(init_mode)), kt.shape)
    xt = np.broadcast_to(np.asarray(xterm, dtype=np.float64), kt.shape + (15,))
    B = kt.shape[0]
    Xref = np.repeat(xt[:, None, :], N, axis=1).copy()
    # Xref[end, :] = range(0, dt*(N-1), length=N); the clock slot carries zero weight in every cost
    Xref[:, :, 14] = np.linspace(0.0, dt * (N - 1), N)
    Uref = np.zeros((B, N - 1, 5))
    K = np.arange(1, N)[None, :]                 # 1-based knot index
    before = K <= (kt[:, None] - 1)              # 1:k_trans-1
    full, half = -mb * g, -mb * g / 2
    lead_col = np.where(im == 1, 1, 3)           # F1y if init_mode == 1 else F2y
    other_col = np.where(im == 1, 3, 1)
    rows = np.arange(B)[:, None]
    Uref[rows, np.arange(N - 1)[None, :], lead_col[:, None]] = np.where(before, full, half)
    Uref[rows, np.arange(N - 1)[None, :], other_col[:, None]] = np.where(before, 0.0, half)
    Uref[:, :, 4] = np.where(before, 0.001, 0.02)
    if not batched:
        return Xref[0], Uref[0]
    return Xref, Uref
