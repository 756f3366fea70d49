"""Synthetic batched landing problems (SURVEY.md 8d / BASELINE.md 5).

The step before the hot path: per-problem random drop states around the notebook's
initial condition (src/main.ipynb:114-124), the notebook's terminal state (:129-132),
cost (:152-161), reference trajectory (src/ref_traj.jl) and initial-guess rule
(src/main.ipynb:181-198) plus noise, so that no structural zero makes the Jacobian
artificially cheap.  Deterministic: numpy.random.default_rng(seed) (PCG64).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .nlp import packZ
from .planar_quadruped import PlanarQuadruped
from .quadratic_cost import lqr_objective
from .ref_traj import reference_trajectory

Q_DIAG = np.array([10.0] * 14 + [0.0])              # src/main.ipynb:152
R_DIAG = np.array([1e-3, 1e-2, 1e-3, 1e-2, 0.0])    # src/main.ipynb:153


@dataclass
class LandingBatch:
    model: PlanarQuadruped
    N: int
    k_trans: np.ndarray    # (B,) int32
    init_mode: np.ndarray  # (B,) int32
    x0: np.ndarray         # (B,15)
    xf: np.ndarray         # (B,15)
    obj: np.ndarray | None  # (N,41) or (B,N,41); None when the cost is built on the device
    Z: np.ndarray          # (B, 20N-5)

    @property
    def B(self):
        return self.x0.shape[0]


def terminal_state(model: PlanarQuadruped):
    """xterm, src/main.ipynb:129-132"""
    xterm = np.zeros(15)
    xterm[0] = -model.lb / 2
    xterm[1] = np.sqrt(model.l1**2 + model.l2**2)
    xterm[5] = -model.lb
    return xterm


def notebook_initial_state(model: PlanarQuadruped, theta0_deg=-30.0, drop_h=2.0):
    """xinit, src/main.ipynb:92-124"""
    x = np.zeros(15)
    x[0] = -model.lb / 2.5
    x[1] = np.sqrt(model.l1**2 + model.l2**2) + 0.1
    x[2] = theta0_deg * np.pi / 180
    x[5] = -model.lb
    x[6] = 0.2
    x[8] = -np.sqrt(2 * 9.81 * drop_h)
    x[9] = -np.pi / 2
    x[13] = -1.0
    return x


def initial_guess(N, k_trans, x0, xf, Uref):
    """Notebook initial-guess rule (src/main.ipynb:181-198) + packZ, vectorised over the batch.
    x0, xf: (B,15); k_trans: (B,); Uref: (B,N-1,5) -> Z (B, 20N-5)."""
    B = x0.shape[0]
    K = np.arange(1, N + 1)[None, :, None]               # 1-based knot
    kt = np.asarray(k_trans)[:, None, None]
    ramp = x0[:, None, :] + (xf - x0)[:, None, :] / (kt - 1) * (K - 1)
    X = np.where(K <= kt, ramp, 0.0)
    after = np.broadcast_to(K > kt, X.shape).copy()
    after[:, :, 14] = False
    X = np.where(after, np.broadcast_to(xf[:, None, :], X.shape), X)
    dtk = np.where(np.arange(1, N)[None, :] < np.asarray(k_trans)[:, None], 0.001, 0.02)
    # Xguess[k+1][end] = Xguess[k][end] + dt_k, sequentially from Xguess[1][end] (src/main.ipynb:191-198)
    X[:, :, 14] = np.cumsum(np.concatenate([X[:, :1, 14], dtk], axis=1), axis=1)
    return packZ(N, X, Uref)


def make_batch(B: int, N: int = 40, k_trans=14, init_mode=1, *, seed: int = 0, ragged: bool = False,
               noise: float = 0.05, dt: float = 0.009, model: PlanarQuadruped | None = None,
               build_obj: bool = True) -> LandingBatch:
    """BASELINE.json configs 2/3/5 (uniform k_trans/init_mode) or config 4 (ragged=True:
    per-problem k_trans ~ U{2..N-1}, init_mode ~ U{1,2}, h ~ U(0.001, 0.02))."""
    model = model or PlanarQuadruped()
    rng = np.random.default_rng(seed)
    if ragged:
        kt = rng.integers(2, N, size=B).astype(np.int32)         # U{2..N-1}
        im = rng.integers(1, 3, size=B).astype(np.int32)
    else:
        kt = np.full(B, k_trans, dtype=np.int32)
        im = np.full(B, init_mode, dtype=np.int32)
    x0 = np.tile(notebook_initial_state(model), (B, 1))
    x0[:, 2] = np.deg2rad(rng.uniform(*THETA0_DEG, size=B))
    x0[:, 6] = rng.uniform(*Y2_0, size=B)
    x0[:, 8] = -np.sqrt(2 * 9.81 * rng.uniform(*DROP_HEIGHT, size=B))
    x0[:, 9] = rng.uniform(*OMEGA0, size=B)
    swap = im == 2                                              # mirror the feet for init_mode 2
    x0[swap, 3:5], x0[swap, 5:7] = x0[swap, 5:7].copy(), x0[swap, 3:5].copy()
    x0[swap, 10:12], x0[swap, 12:14] = x0[swap, 12:14].copy(), x0[swap, 10:12].copy()
    xf = np.tile(terminal_state(model), (B, 1))
    Xref, Uref = reference_trajectory(model, N, kt, xf, im, dt)
    if not build_obj:
        obj = None  # the caller builds it on the device: HybridNLP.set_lqr_cost(Q_DIAG, R_DIAG, Q_DIAG, dt, per_problem=ragged)
    elif ragged:
        obj = lqr_objective(Q_DIAG, R_DIAG, Q_DIAG, Xref, Uref)
    else:
        obj = lqr_objective(Q_DIAG, R_DIAG, Q_DIAG, Xref[0], Uref[0])
    Z = initial_guess(N, kt, x0, xf, Uref)
    Z += rng.normal(0.0, noise, size=Z.shape)
    hcols = 19 + 20 * np.arange(N - 1)
    if ragged:
        Z[:, hcols] = rng.uniform(0.001, 0.02, size=(B, N - 1))
    else:
        Z[:, hcols] = np.clip(Z[:, hcols], 0.001, 0.02)
    return LandingBatch(model, N, kt, im, x0, xf, obj, Z)


# ranges of the random drop states (the notebook's values are their centres)
THETA0_DEG = (-40.0, -10.0)
Y2_0 = (0.1, 0.3)
DROP_HEIGHT = (0.5, 2.5)
OMEGA0 = (-np.pi / 2, 0.0)


def drop_state_sampler(seed: int, model: PlanarQuadruped | None = None, stream_offset: int = 0):
    """The qln_drop_state_sampler that makes qln_sample_drop_states draw exactly what make_batch(seed=seed) draws on
    the host for x0 (uniform k_trans / init_mode batches: the x0 draws are the first 4B of the stream; the ragged
    workload consumes its k_trans / init_mode integers first -- B 64-bit outputs when numpy rejects none of them:
    stream_offset = ragged_descriptors(seed, B, N)[2])."""
    from . import _lib

    model = model or PlanarQuadruped()
    st = np.random.PCG64(seed).state["state"]
    s = _lib.QlnDropStateSampler()
    mask = (1 << 64) - 1
    s.pcg_state[0], s.pcg_state[1] = st["state"] >> 64, st["state"] & mask
    s.pcg_inc[0], s.pcg_inc[1] = st["inc"] >> 64, st["inc"] & mask
    s.stream_offset = int(stream_offset)
    for i, v in enumerate(notebook_initial_state(model)):
        s.x0_template[i] = float(v)
    for name, rng in (("theta_deg", THETA0_DEG), ("y2", Y2_0), ("drop_height", DROP_HEIGHT), ("omega", OMEGA0)):
        getattr(s, name)[0], getattr(s, name)[1] = float(rng[0]), float(rng[1])
    s.two_g = 2 * 9.81
    return s


def ragged_descriptors(seed: int, B: int, N: int, device: int | None = None):
    """config 4's per-problem descriptors -- k_trans ~ U{2..N-1}, init_mode ~ U{1,2}, exactly make_batch(ragged=True)'s first
    two draws -- and where the stream stands behind them.  Returns (k_trans, init_mode, stream_offset) with stream_offset
    the number of 64-bit outputs consumed, i.e. the offset at which qln_sample_drop_states continues numpy's stream -- or
    None if numpy rejected a draw for this seed (Lemire's method redraws with probability < range / 2^32 per draw, which
    shifts every later position): the caller then generates the workload on the host.
    device = None: drawn with numpy on the host, and checked by comparing the generator's state with PCG64(seed) advanced
    by exactly B outputs.  device = ordinal: drawn on that GPU (qln_sample_bounded_integers), which counts the rejections."""
    # numpy consumes nothing for a range of one value (N = 3: k_trans = 2 always); 32-bit draws otherwise, two per 64-bit
    # output -- a buffered odd half is NOT used by the 64-bit uniform draws that follow
    draws_kt = 0 if N - 2 == 1 else B
    offset64 = (draws_kt + B + 1) // 2
    if device is None:
        rng = np.random.default_rng(seed)
        kt = rng.integers(2, N, size=B).astype(np.int32)
        im = rng.integers(1, 3, size=B).astype(np.int32)
        clean = rng.bit_generator.state["state"] == np.random.PCG64(seed).advance(offset64).state["state"]
        return kt, im, (offset64 if clean else None)
    import ctypes as C
    from . import _lib

    st = np.random.PCG64(seed).state["state"]
    mask = (1 << 64) - 1
    state = (C.c_uint64 * 2)(st["state"] >> 64, st["state"] & mask)
    inc = (C.c_uint64 * 2)(st["inc"] >> 64, st["inc"] & mask)
    kt, im = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32)
    rej = C.c_int64()
    total = 0
    i32 = C.POINTER(C.c_int32)
    for out, off, lo, hi in ((kt, 0, 2, N), (im, draws_kt, 1, 3)):
        _lib.check(_lib.lib().qln_sample_bounded_integers(device, state, inc, off, lo, hi, B, out.ctypes.data_as(i32), C.byref(rej)))
        total += rej.value
    return kt, im, (offset64 if total == 0 else None)


# Left-right mirror of the planar model about the midpoint of the landed feet (x -> -lb - x, the feet trade names): the
# dynamics, the clearance row, the costs of the notebook and reference_trajectory() are symmetric under it, the contact
# schedule init_mode 1 <-> 2.  (Of solve()'s variable bounds only quirk Q6's "x1 >= 0" is not.)
_MIRROR_X_FROM = np.array([0, 1, 2, 5, 6, 3, 4, 7, 8, 9, 12, 13, 10, 11, 14])
_MIRROR_X_SIGN = np.array([-1.0, 1, -1, -1, 1, -1, 1, -1, 1, -1, -1, 1, -1, 1, 1])
_MIRROR_U_FROM = np.array([2, 3, 0, 1, 4])
_MIRROR_U_SIGN = np.array([-1.0, 1, -1, 1, 1])


def mirror_states(X, lb: float):
    """(..., 15) states -> their mirror images"""
    Y = np.asarray(X, dtype=np.float64)[..., _MIRROR_X_FROM] * _MIRROR_X_SIGN
    Y[..., [0, 3, 5]] -= lb
    return Y


def mirror_controls(U):
    return np.asarray(U, dtype=np.float64)[..., _MIRROR_U_FROM] * _MIRROR_U_SIGN


def mirror_Z(Z, N: int, lb: float):
    """(B, 20N-5) decision vectors -> those of the mirrored problems"""
    Z = np.asarray(Z, dtype=np.float64)
    out = np.empty_like(Z)
    pad = np.concatenate([Z, np.zeros(Z.shape[:-1] + (5,))], axis=-1).reshape(Z.shape[:-1] + (N, 20))
    M = np.concatenate([mirror_states(pad[..., :15], lb), mirror_controls(pad[..., 15:])], axis=-1)
    out[...] = M.reshape(Z.shape[:-1] + (20 * N,))[..., : 20 * N - 5]
    return out


def mirror_batch(batch: LandingBatch, dt: float = 0.009) -> LandingBatch:
    """The mirror images of a batch of landing problems (uniform cost table): init_mode 1 <-> 2, x0 / xf / Z mirrored,
    the cost rebuilt from reference_trajectory() of the other init_mode."""
    lb = batch.model.lb
    im = (3 - batch.init_mode).astype(np.int32)
    x0, xf = mirror_states(batch.x0, lb), mirror_states(batch.xf, lb)
    Xref, Uref = reference_trajectory(batch.model, batch.N, batch.k_trans, xf, im, dt)
    obj = None
    if batch.obj is not None:
        obj = lqr_objective(Q_DIAG, R_DIAG, Q_DIAG, Xref, Uref) if batch.obj.ndim == 3 else lqr_objective(Q_DIAG, R_DIAG, Q_DIAG, Xref[0], Uref[0])
    return LandingBatch(batch.model, batch.N, batch.k_trans.copy(), im, x0, xf, obj, mirror_Z(batch.Z, batch.N, lb))


def notebook_problem(N: int = 61, k_trans: int = 21, init_mode: int = 1, dt: float = 0.009,
                     model: PlanarQuadruped | None = None) -> LandingBatch:
    """The literal notebook problem (src/main.ipynb cells 2-8) as a batch of one, Z = Z0."""
    model = model or PlanarQuadruped()
    x0 = notebook_initial_state(model)[None, :]
    xf = terminal_state(model)[None, :]
    kt = np.array([k_trans], dtype=np.int32)
    im = np.array([init_mode], dtype=np.int32)
    Xref, Uref = reference_trajectory(model, N, kt, xf, im, dt)
    obj = lqr_objective(Q_DIAG, R_DIAG, Q_DIAG, Xref[0], Uref[0])
    Z = initial_guess(N, kt, x0, xf, Uref)
    return LandingBatch(model, N, kt, im, x0, xf, obj, Z)
