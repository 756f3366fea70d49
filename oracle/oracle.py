"""ctypes binding of the C parity oracle (oracle/qln_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by quadruped_landing_amd/.  The oracle is a
CPU restatement of the reference's Julia path (see qln_oracle.h for what pins it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqln_oracle.so")

NX, NU, NZ, COST_STRIDE = 15, 5, 20, 41


class Model(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("g", "mb", "mf", "lb", "l1", "l2")]


class Problem(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("k_trans", C.c_int32),
        ("init_mode", C.c_int32),
        ("model", Model),
        ("x0", C.c_double * NX),
        ("xf", C.c_double * NX),
        ("cost", C.POINTER(C.c_double)),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("N", C.c_int32),
        ("model", Model),
        ("k_trans", C.POINTER(C.c_int32)),
        ("init_mode", C.POINTER(C.c_int32)),
        ("x0", C.POINTER(C.c_double)),
        ("xf", C.POINTER(C.c_double)),
        ("cost", C.POINTER(C.c_double)),
        ("cost_batch", C.c_int32),
        ("z_stride", C.c_int64),
        ("c_off", C.POINTER(C.c_int64)),
        ("j_off", C.POINTER(C.c_int64)),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "qln_oracle.c")
    hdr = os.path.join(_HERE, "qln_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr)
    )
    if force or stale:
        if not os.path.exists(src):  # prebuilt-only checkout
            raise FileNotFoundError(src)
        subprocess.check_call(["make", "-C", _HERE, "-B", "libqln_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.orc_default_model.argtypes = [C.POINTER(Model)]
        L.orc_num_primals.restype = C.c_int32
        L.orc_num_primals.argtypes = [C.c_int32]
        L.orc_num_duals.restype = C.c_int32
        L.orc_num_duals.argtypes = [C.c_int32, C.c_int32]
        L.orc_cinds.argtypes = [C.c_int32, C.c_int32, ip]
        L.orc_constraint_bounds.argtypes = [C.c_int32, C.c_int32, dp, dp]
        L.orc_contact_dynamics.argtypes = [C.POINTER(Model), C.c_int, dp, dp, dp]
        L.orc_contact_dynamics_rk4.argtypes = [C.POINTER(Model), C.c_int, dp, dp, dp]
        L.orc_contact_jacobian.argtypes = [C.POINTER(Model), C.c_int, dp, dp, dp]
        L.orc_jump_map.argtypes = [dp, dp]
        L.orc_jump_jacobian_diag.argtypes = [dp]
        L.orc_reference_trajectory.argtypes = [C.POINTER(Model), C.c_int32, C.c_int32, dp, C.c_int32, C.c_double, dp, dp]
        L.orc_lqr_cost.argtypes = [dp, dp, dp, dp, dp]
        L.orc_stagecost.restype = C.c_double
        L.orc_stagecost.argtypes = [dp, dp, dp]
        L.orc_termcost.restype = C.c_double
        L.orc_termcost.argtypes = [dp, dp]
        L.orc_eval_f.restype = C.c_double
        L.orc_eval_f.argtypes = [C.POINTER(Problem), dp]
        L.orc_grad_f.argtypes = [C.POINTER(Problem), dp, dp]
        L.orc_eval_c.argtypes = [C.POINTER(Problem), dp, dp]
        L.orc_jac_c_dense.argtypes = [C.POINTER(Problem), dp, dp]
        L.orc_jac_nnz.restype = C.c_int32
        L.orc_jac_nnz.argtypes = [C.c_int32, C.c_int32]
        L.orc_jac_nnz_dynamic.restype = C.c_int32
        L.orc_jac_nnz_dynamic.argtypes = [C.c_int32]
        L.orc_jac_structure.argtypes = [C.POINTER(Problem), ip, ip]
        L.orc_jac_c_coo.argtypes = [C.POINTER(Problem), dp, dp]
        L.orc_batch_eval_c_jac.argtypes = [C.POINTER(Batch), dp, dp, dp, C.c_int]
        L.orc_batch_eval_f_grad.argtypes = [C.POINTER(Batch), dp, dp, dp, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_model() -> Model:
    m = Model()
    lib().orc_default_model(C.byref(m))
    return m


# ----------------------------------------------------------------- single problem


class OracleNLP:
    """One landing problem evaluated by the oracle; mirrors the fields of HybridNLP."""

    def __init__(self, N, k_trans, init_mode, x0, xf, cost, model=None):
        self.N, self.k_trans, self.init_mode = int(N), int(k_trans), int(init_mode)
        self.model = model or default_model()
        self.cost = _f64(cost).reshape(self.N, COST_STRIDE)
        self._p = Problem()
        self._p.N, self._p.k_trans, self._p.init_mode = self.N, self.k_trans, self.init_mode
        self._p.model = self.model
        self._p.x0[:] = list(_f64(x0))
        self._p.xf[:] = list(_f64(xf))
        self._p.cost = _dp(self.cost)
        self.n_nlp = lib().orc_num_primals(self.N)
        self.m_nlp = lib().orc_num_duals(self.N, self.k_trans)
        self.nnz = lib().orc_jac_nnz(self.N, self.k_trans)
        self.nnz_dynamic = lib().orc_jac_nnz_dynamic(self.N)

    def cinds(self):
        out = np.zeros(14, dtype=np.int32)
        lib().orc_cinds(self.N, self.k_trans, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(7)]

    def bounds(self):
        lb, ub = np.empty(self.m_nlp), np.empty(self.m_nlp)
        lib().orc_constraint_bounds(self.N, self.k_trans, _dp(lb), _dp(ub))
        return lb, ub

    def eval_f(self, Z):
        Z = _f64(Z)
        return lib().orc_eval_f(C.byref(self._p), _dp(Z))

    def grad_f(self, Z):
        Z = _f64(Z)
        g = np.zeros(self.n_nlp)
        lib().orc_grad_f(C.byref(self._p), _dp(g), _dp(Z))
        return g

    def eval_c(self, Z):
        Z = _f64(Z)
        c = np.full(self.m_nlp, np.nan)
        lib().orc_eval_c(C.byref(self._p), _dp(c), _dp(Z))
        return c

    def jac_c_dense(self, Z, fill=np.nan):
        """Column-major m_nlp x n_nlp, returned as an (m_nlp, n_nlp) array; entries the
        reference never assigns keep `fill`."""
        Z = _f64(Z)
        jac = np.full((self.n_nlp, self.m_nlp), fill, dtype=np.float64)  # C-order of the transpose = column-major
        lib().orc_jac_c_dense(C.byref(self._p), _dp(jac), _dp(Z))
        return jac.T

    def jac_structure(self):
        rows = np.zeros(self.nnz, dtype=np.int32)
        cols = np.zeros(self.nnz, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        lib().orc_jac_structure(C.byref(self._p), rows.ctypes.data_as(ip), cols.ctypes.data_as(ip))
        return rows, cols

    def jac_c_coo(self, Z):
        Z = _f64(Z)
        v = np.full(self.nnz, np.nan)
        lib().orc_jac_c_coo(C.byref(self._p), _dp(v), _dp(Z))
        return v


def contact_dynamics(mode, s, u, model=None):
    m = model or default_model()
    s, u = _f64(s), _f64(u)
    out = np.zeros(14)
    lib().orc_contact_dynamics(C.byref(m), int(mode), _dp(s), _dp(u), _dp(out))
    return out


def contact_dynamics_rk4(mode, x, u, model=None):
    m = model or default_model()
    x, u = _f64(x), _f64(u)
    out = np.zeros(NX)
    lib().orc_contact_dynamics_rk4(C.byref(m), int(mode), _dp(x), _dp(u), _dp(out))
    return out


def contact_jacobian(mode, x, u, model=None):
    m = model or default_model()
    x, u = _f64(x), _f64(u)
    J = np.zeros((NZ, NX))
    lib().orc_contact_jacobian(C.byref(m), int(mode), _dp(x), _dp(u), _dp(J))
    return J.T  # (15, 20)


def reference_trajectory(N, k_trans, xterm, init_mode, dt, model=None):
    m = model or default_model()
    xterm = _f64(xterm)
    Xref, Uref = np.zeros((N, NX)), np.zeros((N - 1, NU))
    lib().orc_reference_trajectory(C.byref(m), N, k_trans, _dp(xterm), init_mode, float(dt), _dp(Xref), _dp(Uref))
    return Xref, Uref


def lqr_cost(Qd, Rd, xf, uf):
    Qd, Rd, xf, uf = _f64(Qd), _f64(Rd), _f64(xf), _f64(uf)
    out = np.zeros(COST_STRIDE)
    lib().orc_lqr_cost(_dp(Qd), _dp(Rd), _dp(xf), _dp(uf), _dp(out))
    return out


def lqr_cost_table(Qd, Rd, Qfd, Xref, Uref):
    """src/main.ipynb:158-161: obj[k] = LQRCost(Q,R,Xref[k],Uref[k]); obj[N] = LQRCost(Qf, R*0, Xref[N], Uref[1])."""
    N = Xref.shape[0]
    tab = np.zeros((N, COST_STRIDE))
    for k in range(N - 1):
        tab[k] = lqr_cost(Qd, Rd, Xref[k], Uref[k])
    tab[N - 1] = lqr_cost(Qfd, np.asarray(Rd) * 0, Xref[N - 1], Uref[0])
    return tab


# ----------------------------------------------------------------- notebook problem (src/main.ipynb)


def notebook_problem(N=61, k_trans=21, init_mode=1, dt=0.009, theta0_deg=-30.0, drop_h=2.0):
    """The literal problem of src/main.ipynb:92-132,152-161 (cells 2-6)."""
    m = default_model()
    lb, l1, l2 = m.lb, m.l1, m.l2
    v_init_y = np.sqrt(2 * 9.81 * drop_h)
    xinit = np.zeros(NX)
    xinit[0] = -lb / 2.5
    xinit[1] = np.sqrt(l1**2 + l2**2) + 0.1
    xinit[2] = theta0_deg * np.pi / 180
    xinit[5] = -lb
    xinit[6] = 0.2
    xinit[8] = -v_init_y
    xinit[9] = -np.pi / 2
    xinit[13] = -1.0
    xterm = np.zeros(NX)
    xterm[0] = -lb / 2
    xterm[1] = np.sqrt(l1**2 + l2**2)
    xterm[5] = -lb
    Xref, Uref = reference_trajectory(N, k_trans, xterm, init_mode, dt, m)
    Qd = np.array([10.0] * 14 + [0.0])
    Rd = np.array([1e-3, 1e-2, 1e-3, 1e-2, 0.0])
    cost = lqr_cost_table(Qd, Rd, Qd, Xref, Uref)
    nlp = OracleNLP(N, k_trans, init_mode, xinit, xterm, cost, m)
    return nlp, xinit, xterm, Xref, Uref


def notebook_initial_guess(N, k_trans, xinit, xterm, Uref):
    """src/main.ipynb:181-198 (cell 7) + packZ (src/nlp.jl:94-102)."""
    X = np.zeros((N, NX))
    for k in range(1, N + 1):
        if k <= k_trans:
            X[k - 1] = xinit + (xterm - xinit) / (k_trans - 1) * (k - 1)
        else:
            X[k - 1, :14] = xterm[:14]
    for k in range(1, N):
        X[k, 14] = X[k - 1, 14] + (0.001 if k < k_trans else 0.02)
    Z = np.zeros(NZ * N - NU)
    for k in range(N - 1):
        Z[NZ * k : NZ * k + NX] = X[k]
        Z[NZ * k + NX : NZ * k + NZ] = Uref[k]
    Z[NZ * (N - 1) :] = X[N - 1]
    return Z


# ----------------------------------------------------------------- batched


def batch_eval(N, model, k_trans, init_mode, x0, xf, cost, Z, z_stride, c_off, j_off, c_total, j_total,
               want_c=True, want_j=True, want_f=False, want_grad=False, nthreads=1):
    """Evaluate a batch with the oracle using the same strides/offsets as the product ABI."""
    k_trans = np.ascontiguousarray(k_trans, dtype=np.int32)
    init_mode = np.ascontiguousarray(init_mode, dtype=np.int32)
    B = k_trans.shape[0]
    x0, xf, cost, Z = _f64(x0), _f64(xf), _f64(cost), _f64(Z)
    c_off = np.ascontiguousarray(c_off, dtype=np.int64)
    j_off = np.ascontiguousarray(j_off, dtype=np.int64)
    b = Batch()
    b.B, b.N, b.model = B, N, model
    ip, lp = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    b.k_trans, b.init_mode = k_trans.ctypes.data_as(ip), init_mode.ctypes.data_as(ip)
    b.x0, b.xf, b.cost = _dp(x0), _dp(xf), _dp(cost)
    b.cost_batch = 1 if cost.size == N * COST_STRIDE else B
    b.z_stride = int(z_stride)
    b.c_off, b.j_off = c_off.ctypes.data_as(lp), j_off.ctypes.data_as(lp)
    out = {}
    if want_c or want_j:
        c = np.full(int(c_total), np.nan) if want_c else None
        v = np.full(int(j_total), np.nan) if want_j else None
        lib().orc_batch_eval_c_jac(C.byref(b), _dp(Z), _dp(c), _dp(v), int(nthreads))
        out["c"], out["vals"] = c, v
    if want_f or want_grad:
        f = np.zeros(B) if want_f else None
        g = np.zeros(Z.shape[0]) if want_grad else None
        lib().orc_batch_eval_f_grad(C.byref(b), _dp(Z), _dp(f), _dp(g), int(nthreads))
        out["f"], out["grad"] = f, g
    return out
