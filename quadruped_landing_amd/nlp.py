"""HybridNLP: the landing NLP (mirror of src/nlp.jl:13-114) generalised to a batch of B
independent problems, evaluated by the gfx950 kernels through the C ABI.

The index maps (xinds/uinds/cinds), bounds and sizes are host logic and are pure
numpy; every evaluation (eval_f, grad_f, eval_c, jac_c) is a kernel launch behind
include/qln_evaluator.h -- there is no CPU implementation of the arithmetic here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .planar_quadruped import PlanarQuadruped

n, m = 15, 5


# ----------------------------------------------------------------------------- index maps


def num_primals(N: int) -> int:
    """src/nlp.jl:86"""
    return n * N + m * (N - 1)


def num_duals(N: int, k_trans: int) -> int:
    """src/nlp.jl:87 (= cinds[end][end])"""
    return cinds(N, k_trans)[-1][-1]


def xinds(N: int):
    """src/nlp.jl:38, 1-based like the reference."""
    return [np.arange(1, n + 1) + (k - 1) * (n + m) for k in range(1, N + 1)]


def uinds(N: int):
    """src/nlp.jl:39, 1-based."""
    return [np.arange(n + 1, n + m + 1) + (k - 1) * (n + m) for k in range(1, N)]


def cinds(N: int, k_trans: int):
    """src/nlp.jl:48-63: seven 1-based inclusive (start, end) ranges."""
    out, e = [], 0
    for ln in (n, n - 1, (N - 1) * n, N, N - k_trans + 1, 1, N):
        out.append((e + 1, e + ln))
        e += ln
    return out


def constraint_bounds(N: int, k_trans: int):
    """src/nlp.jl:66-69"""
    ci = cinds(N, k_trans)
    m_nlp = ci[-1][-1]
    lb, ub = np.zeros(m_nlp), np.zeros(m_nlp)
    ub[ci[6][0] - 1 : ci[6][1]] = np.inf
    return lb, ub


def packZ(N: int, X, U):
    """src/nlp.jl:94-102.  X: (...,N,15), U: (...,N-1,5) -> Z (...,20N-5)."""
    X = np.asarray(X, dtype=np.float64)
    U = np.asarray(U, dtype=np.float64)
    Z = np.zeros(X.shape[:-2] + (num_primals(N),))
    body = Z[..., : 20 * (N - 1)].reshape(X.shape[:-2] + (N - 1, 20))
    body[..., :15] = X[..., : N - 1, :]
    body[..., 15:] = U
    Z[..., 20 * (N - 1) :] = X[..., N - 1, :]
    return Z


def unpackZ(N: int, Z):
    """src/nlp.jl:110-114"""
    Z = np.asarray(Z)
    body = Z[..., : 20 * (N - 1)].reshape(Z.shape[:-1] + (N - 1, 20))
    X = np.concatenate([body[..., :15], Z[..., None, 20 * (N - 1) :]], axis=-2)
    return X, body[..., 15:]


# ----------------------------------------------------------------------------- evaluator


def _torch():
    import torch

    return torch


# layout of the step-block section of vals (include/qln_evaluator.h, QLN_JAC_FORMAT_*)
JAC_FORMATS = {"dense_blocks": _lib.QLN_JAC_FORMAT_DENSE_BLOCKS, "structural": _lib.QLN_JAC_FORMAT_STRUCTURAL}


class _PlacedBuffer:
    """Owner of a buffer obtained from qln_vals_alloc_placed, exposed to torch through __cuda_array_interface__
    (the tensor made from it keeps this object alive; the buffer goes back to the driver when both are gone)."""

    def __init__(self, nlp, ptr: int, numel: int):
        self._nlp, self._ptr = nlp, ptr
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def __del__(self):
        h = getattr(self._nlp, "_h", None)
        if h and self._ptr:
            # qln_vals_free_placed waits for the whole device (the tensor may have been used on any stream) and
            # reports a failing unmap / release; a destructor cannot raise, so a failure becomes a warning
            try:
                rc = _lib.lib().qln_vals_free_placed(h, self._ptr)
                if rc != _lib.QLN_OK:
                    import warnings

                    warnings.warn(f"qln_vals_free_placed failed ({rc}): {_lib.lib().qln_last_error().decode()}")
            except Exception:
                pass
            self._ptr = 0


class HybridNLP:
    """Batch of B landing problems with a common horizon N on one GPU.

    Constructor arguments follow HybridNLP(model, obj, init_mode, k_trans, N, x0, xf)
    (src/nlp.jl:34-37); `obj` is the 41-double cost table of quadratic_cost.lqr_objective,
    either (N,41) shared by all problems or (B,N,41).  Scalars broadcast over the batch.
    """

    def __init__(self, model: PlanarQuadruped, obj, init_mode, k_trans, N: int, x0, xf, *,
                 device: int = 0, z_stride: int = 0, align: int = 16, stream=None, jac_format: str = "dense_blocks"):
        self.model = model
        if jac_format not in JAC_FORMATS:
            raise ValueError(f"jac_format must be one of {sorted(JAC_FORMATS)}")
        self.jac_format = jac_format
        self.N = int(N)
        x0 = np.asarray(x0, dtype=np.float64)
        B = x0.shape[0] if x0.ndim == 2 else 1
        self.B = B
        self.x0 = np.ascontiguousarray(np.broadcast_to(x0, (B, n)))
        self.xf = np.ascontiguousarray(np.broadcast_to(np.asarray(xf, dtype=np.float64), (B, n)))
        self.k_trans = np.ascontiguousarray(np.broadcast_to(np.asarray(k_trans, dtype=np.int32), (B,)))
        self.init_mode = np.ascontiguousarray(np.broadcast_to(np.asarray(init_mode, dtype=np.int32), (B,)))
        if obj is None:  # no cost yet: build it on the device with set_lqr_cost()
            self.cost_batch = 1
        else:
            obj = np.ascontiguousarray(obj, dtype=np.float64)
            if obj.shape == (self.N, _lib.COST_STRIDE):
                self.cost_batch = 1
            elif obj.shape == (B, self.N, _lib.COST_STRIDE):
                self.cost_batch = B
            else:
                raise ValueError(f"obj must have shape (N,41) or (B,N,41), got {obj.shape}")
        self.obj = obj
        self.device = int(device)

        L = _lib.lib()
        d = _lib.QlnBatchDesc()
        d.B, d.N = B, self.N
        d.model = _lib.QlnModel(model.g, model.mb, model.mf, model.lb, model.l1, model.l2)
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        d.k_trans = self.k_trans.ctypes.data_as(ip)
        d.init_mode = self.init_mode.ctypes.data_as(ip)
        d.x0, d.xf = self.x0.ctypes.data_as(dp), self.xf.ctypes.data_as(dp)
        d.cost = self.obj.ctypes.data_as(dp) if self.obj is not None else None
        d.cost_batch, d.z_stride, d.align = self.cost_batch, int(z_stride), int(align)
        d.jac_format = JAC_FORMATS[jac_format]
        h = C.c_void_p()
        _lib.check(L.qln_create(C.byref(d), self.device, C.byref(h)))
        self._h = h
        dims = _lib.QlnDims()
        _lib.check(L.qln_get_dims(self._h, C.byref(dims)))
        self.dims = dims
        self.n_nlp, self.z_stride = dims.n_nlp, dims.z_stride
        self.nnz_dynamic = dims.nnz_dynamic
        self.c_off = np.zeros(B, dtype=np.int64)
        self.j_off = np.zeros(B, dtype=np.int64)
        lp = C.POINTER(C.c_int64)
        _lib.check(L.qln_get_offsets(self._h, self.c_off.ctypes.data_as(lp), self.j_off.ctypes.data_as(lp)))
        if stream is not None:
            self.set_stream(stream)

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().qln_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """`stream`: a torch.cuda.Stream or a raw hipStream_t address."""
        ptr = getattr(stream, "cuda_stream", stream)
        _lib.check(_lib.lib().qln_set_stream(self._h, C.c_void_p(int(ptr))))

    def synchronize(self):
        _lib.check(_lib.lib().qln_synchronize(self._h))

    # -- sizes / index maps (src/nlp.jl:85-87) ------------------------------------------------
    def num_primals(self) -> int:
        return self.n_nlp

    def num_duals(self, b: int = 0) -> int:
        mm, _ = self.problem_dims(b)
        return mm

    def size(self):
        return (n, m, self.N)

    def problem_dims(self, b: int = 0):
        mm, nz = C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().qln_problem_dims(self._h, b, C.byref(mm), C.byref(nz)))
        return mm.value, nz.value

    def problem_nnz_dynamic(self, b: int = 0) -> int:
        """State-dependent values at the head of problem b's vals segment (step blocks + N clearance entries)."""
        nd = C.c_int32()
        _lib.check(_lib.lib().qln_problem_nnz_dynamic(self._h, b, C.byref(nd)))
        return nd.value

    def cinds(self, b: int = 0):
        out = (C.c_int32 * 14)()
        _lib.check(_lib.lib().qln_constraint_index_ranges(self._h, b, out))
        return [(out[2 * i], out[2 * i + 1]) for i in range(7)]

    def bounds(self, b: int = 0):
        """(lb, ub) of problem b, src/nlp.jl:66-69."""
        mm, _ = self.problem_dims(b)
        lb, ub = np.empty(mm), np.empty(mm)
        _lib.check(_lib.lib().qln_constraint_bounds(self._h, b, lb.ctypes.data, ub.ctypes.data))
        return lb, ub

    def jacobian_structure(self, b: int = 0):
        """0-based (rows, cols) of problem b's block-COO values."""
        _, nz = self.problem_dims(b)
        rows, cols = np.zeros(nz, dtype=np.int32), np.zeros(nz, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        _lib.check(_lib.lib().qln_jacobian_structure(self._h, b, rows.ctypes.data_as(ip), cols.ctypes.data_as(ip)))
        return rows, cols

    # -- device buffers -------------------------------------------------------------------------
    def _dev(self):
        return _torch().device("cuda", self.device)

    def new_Z(self):
        return _torch().zeros(self.dims.z_total, dtype=_torch().float64, device=self._dev())

    def new_c(self):
        return _torch().zeros(self.dims.c_total, dtype=_torch().float64, device=self._dev())

    def new_vals(self):
        return _torch().zeros(self.dims.j_total, dtype=_torch().float64, device=self._dev())

    def placed_info(self, vals):
        """(chunk_bytes, chunks_scanned, window_first_chunk) of a buffer made by new_vals_regions."""
        a, b, c_ = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().qln_vals_placed_info(self._h, vals.data_ptr(), C.byref(a), C.byref(b), C.byref(c_)))
        return a.value, b.value, c_.value

    @staticmethod
    def placed_address_space():
        """(bytes of virtual address space retired by placed allocations in this process, the cap at which the library refuses)"""
        a, b = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().qln_vals_placed_address_space(C.byref(a), C.byref(b)))
        return a.value, b.value

    def new_vals_regions(self, Z, c=None, transient_gib: float = 64.0):
        """The Jacobian buffer placed across two 32-GiB regions of device memory (qln_vals_alloc_placed: HIP
        virtual-memory API, the fused launch timed on windows of a j_total + `transient_gib` range, the fastest window kept
        and everything else released).  Returns (vals, ms) -- ms = launch time on the window kept.  Raises QlnError if the
        device has not got the memory free or the virtual-memory API fails."""
        t = _torch()
        self._check(Z, self.dims.z_total, "Z")
        if c is not None:  # overwritten by the timed launches; None = the library uses a scratch buffer of its own
            self._check(c, self.dims.c_total, "c")
        ptr, ms = C.c_void_p(), C.c_float()
        _lib.check(_lib.lib().qln_vals_alloc_placed_budget(self._h, Z.data_ptr(), None if c is None else c.data_ptr(),
                                                           int(transient_gib * 2**30), C.byref(ptr), C.byref(ms)))
        vals = t.as_tensor(_PlacedBuffer(self, ptr.value, int(self.dims.j_total)), device=self._dev())
        return vals, float(ms.value)

    def new_vals_placed(self, Z, c, trials: int = 4, launches: int = 3, spread: bool = True, regions: bool = True):
        """Setup-time placement choice for the (large, long-lived) Jacobian buffer.

        Where a multi-GB buffer lies physically changes the store bandwidth the hot kernel sustains by ~20 % on
        MI355X: device memory behaves as 32-GiB regions, and the kernel's eight write fronts (the first four in the
        first half of the buffer) run fastest when the two halves lie in different regions (DESIGN.md section 5,
        profiles/r01_placement_windows.txt).  With `regions` the buffer is built that way (new_vals_regions).
        If that is not possible (too little free memory), or `regions` is off, `trials` plain allocations are
        timed instead -- held simultaneously and, with `spread`, ~32 GiB apart so that they sample different
        regions -- and the fastest is kept.  Returns (vals, [ms]) -- one time per candidate tried.
        """
        t = _torch()
        if regions:
            try:
                vals, ms = self.new_vals_regions(Z, c)
                return vals, [ms]
            except _lib.QlnError:
                pass
        trials = max(1, int(trials))
        nbytes = 8 * int(self.dims.j_total)
        spacer_bytes = 0
        if spread and trials > 1:
            free, _ = t.cuda.mem_get_info(self._dev())
            # one candidate + one spacer per trial, about 32 GiB apart, within 85 % of what is free now
            spacer_bytes = max(0, min(32 * 2**30 - nbytes, int(0.85 * free) // trials - nbytes))
        cands, times, spacers = [], [], []
        for i in range(trials):
            v = self.new_vals()
            self.init_jacobian_constants(v)
            ms = self.time_c_and_jac(Z, c, v, warmup=1, iters=max(1, int(launches)))
            cands.append(v)
            times.append(float(np.median(ms)))
            if spacer_bytes >= 2**20 and i + 1 < trials:
                spacers.append(t.empty(spacer_bytes, dtype=t.uint8, device=self._dev()))
        best = int(np.argmin(times))
        vals = cands[best]
        cands.clear()
        spacers.clear()
        del v
        t.cuda.empty_cache()
        return vals, times

    def new_f(self):
        return _torch().zeros(self.B, dtype=_torch().float64, device=self._dev())

    def upload_Z(self, Z_host):
        """Z_host: (B, n_nlp) -> device buffer in the handle's z_stride layout."""
        t = _torch()
        Z_host = np.asarray(Z_host, dtype=np.float64).reshape(self.B, self.n_nlp)
        buf = np.zeros((self.B, self.z_stride))
        buf[:, : self.n_nlp] = Z_host
        return t.from_numpy(buf.reshape(-1)).to(self._dev())

    def _check(self, t, total, name):
        T = _torch()
        if not (isinstance(t, T.Tensor) and t.is_cuda and t.dtype == T.float64 and t.is_contiguous()):
            raise TypeError(f"{name} must be a contiguous float64 CUDA tensor")
        if t.device.index != self.device:
            raise ValueError(f"{name} is on cuda:{t.device.index}, handle is on cuda:{self.device}")
        if t.numel() < total:
            raise ValueError(f"{name} has {t.numel()} elements, need {total}")
        return C.c_void_p(t.data_ptr())

    # -- evaluation (device tensors, stream-ordered) ----------------------------------------------
    def eval_f(self, Z, out=None):
        """src/costs.jl:6-16 for every problem -> (B,) tensor."""
        out = self.new_f() if out is None else out
        _lib.check(_lib.lib().qln_eval_objective(self._h, self._check(Z, self.dims.z_total, "Z"), self._check(out, self.B, "f")))
        return out

    def grad_f(self, Z, out=None):
        """src/costs.jl:23-34 -> tensor laid out like Z."""
        out = self.new_Z() if out is None else out
        _lib.check(_lib.lib().qln_eval_objective_gradient(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(out, self.dims.z_total, "grad")))
        return out

    def eval_c(self, Z, out=None):
        """src/constraints.jl:145-158 -> c buffer (problem b at c_off[b])."""
        out = self.new_c() if out is None else out
        _lib.check(_lib.lib().qln_eval_constraint(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(out, self.dims.c_total, "c")))
        return out

    def jac_c(self, Z, out=None, write_constants: bool = True):
        """src/constraints.jl:212-291 -> block-COO values (problem b at j_off[b])."""
        out = self.new_vals() if out is None else out
        flags = _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0
        _lib.check(_lib.lib().qln_eval_constraint_jacobian(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(out, self.dims.j_total, "vals"), flags))
        return out

    def eval_c_and_jac(self, Z, c=None, vals=None, write_constants: bool = True):
        """The fused hot path: eval_c! + jac_c! in one launch."""
        c = self.new_c() if c is None else c
        vals = self.new_vals() if vals is None else vals
        flags = _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0
        _lib.check(_lib.lib().qln_eval_constraint_and_jacobian(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(c, self.dims.c_total, "c"),
            self._check(vals, self.dims.j_total, "vals"), flags))
        return c, vals

    def eval_all(self, Z, f=None, grad=None, c=None, vals=None, write_constants: bool = True):
        """f, grad_f, c and the Jacobian values from one read of Z, in one launch (qln_eval_all)."""
        f = self.new_f() if f is None else f
        grad = self.new_Z() if grad is None else grad
        c = self.new_c() if c is None else c
        vals = self.new_vals() if vals is None else vals
        flags = _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0
        _lib.check(_lib.lib().qln_eval_all(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(f, self.B, "f"),
            self._check(grad, self.dims.z_total, "grad"), self._check(c, self.dims.c_total, "c"),
            self._check(vals, self.dims.j_total, "vals"), flags))
        return f, grad, c, vals

    def eval_f_and_c(self, Z, f=None, c=None):
        """f and c from one read of Z in one launch (qln_eval_objective_and_constraint): what a line search asks for."""
        f = self.new_f() if f is None else f
        c = self.new_c() if c is None else c
        _lib.check(_lib.lib().qln_eval_objective_and_constraint(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(f, self.B, "f"), self._check(c, self.dims.c_total, "c")))
        return f, c

    def init_jacobian_constants(self, vals):
        _lib.check(_lib.lib().qln_jacobian_init_constants(self._h, self._check(vals, self.dims.j_total, "vals")))
        return vals

    def jac_vec(self, Z, v, out=None):
        """y = jac_c(Z) @ v for every problem (v in Z's layout, y in c's); the Jacobian is re-derived in registers."""
        self._check(Z, self.dims.z_total, "Z")
        self._check(v, self.dims.z_total, "v")
        out = self.new_c() if out is None else out
        self._check(out, self.dims.c_total, "out")
        _lib.check(_lib.lib().qln_eval_constraint_jvp(self._h, Z.data_ptr(), v.data_ptr(), out.data_ptr()))
        return out

    def jac_t_vec(self, Z, lam, out=None):
        """g = jac_c(Z)' @ lam for every problem (lam in c's layout, g in Z's)."""
        self._check(Z, self.dims.z_total, "Z")
        self._check(lam, self.dims.c_total, "lam")
        out = self.new_Z() if out is None else out
        self._check(out, self.dims.z_total, "out")
        _lib.check(_lib.lib().qln_eval_constraint_vjp(self._h, Z.data_ptr(), lam.data_ptr(), out.data_ptr()))
        return out

    def gauss_newton_step(self, Z, c, out=None, max_iters: int = 200, rel_tol: float = 1e-10, radius=None,
                          col_scale=None, info=None):
        """dZ = D x, x the minimum-norm least-squares step on the constraint violation of every problem (CGLS in LDS,
        one wave per problem); `c` = eval_c(Z).  Optional device tensors: `radius` (B,) trust radius on |x|,
        `col_scale` (n_nlp,) diagonal of D (0 = variable held fixed), `info` (B, 8): {iterations, |(AD)'rho|^2,
        |(AD)'(A dZ + rho)|^2, |A dZ + rho|^2, |rho|^2, cut at the radius, |x|, 0}."""
        self._check(Z, self.dims.z_total, "Z")
        self._check(c, self.dims.c_total, "c")
        out = self.new_Z() if out is None else out
        self._check(out, self.dims.z_total, "out")
        ptr = lambda t, total, name: None if t is None else self._check(t, total, name)
        _lib.check(_lib.lib().qln_gauss_newton_step(self._h, Z.data_ptr(), c.data_ptr(), out.data_ptr(), int(max_iters),
                                                    float(rel_tol), ptr(radius, self.B, "radius"),
                                                    ptr(col_scale, self.n_nlp, "col_scale"),
                                                    ptr(info, _lib.GN_INFO_STRIDE * self.B, "info")))
        return out

    def solve(self, Z, info=None, **options):
        """Batched solve of the reference NLP (solve(), src/moi.jl:46-103) on the GPU, in place on `Z` (device tensor,
        layout of Z; the controls of the guess are used, the states are rolled out from x0).  `options`: fields of
        qln_solve_options (max_outer, max_inner, tol_violation, rho0, ..., q6_bounds, exact_h_gradient).
        Returns (Z, info) with info a (B, 16) tensor: outer iterations, iLQR iterations, f, violation, rho, status, ..."""
        t = _torch()
        self._check(Z, self.dims.z_total, "Z")
        if info is None:
            info = t.zeros(self.B * _lib.SOLVE_INFO_STRIDE, dtype=t.float64, device=self._dev())
        self._check(info, self.B * _lib.SOLVE_INFO_STRIDE, "info")
        opt = _lib.QlnSolveOptions()
        _lib.check(_lib.lib().qln_solve_default_options(C.byref(opt)))
        for k, v in options.items():
            if not hasattr(opt, k):
                raise TypeError(f"unknown solve option {k!r}")
            setattr(opt, k, v)
        _lib.check(_lib.lib().qln_solve(self._h, Z.data_ptr(), C.byref(opt), info.data_ptr()))
        return Z, info.view(self.B, _lib.SOLVE_INFO_STRIDE)

    def kinematic_constraint(self, Z, with_jacobian: bool = True):
        """OPT-IN, nothing in the reference to compare with: the leg-length rows the reference has only as commented-out code
        (src/constraints.jl:115-138).  Returns (d (B, 2N), jac (B, 2N, 4) or None, (lower, upper))."""
        t = _torch()
        self._check(Z, self.dims.z_total, "Z")
        d = t.empty(self.B * 2 * self.N, dtype=t.float64, device=self._dev())
        jac = t.empty(self.B * 8 * self.N, dtype=t.float64, device=self._dev()) if with_jacobian else None
        _lib.check(_lib.lib().qln_eval_kinematic_constraint(self._h, Z.data_ptr(), d.data_ptr(),
                                                            jac.data_ptr() if jac is not None else None))
        lo, up = C.c_double(), C.c_double()
        _lib.check(_lib.lib().qln_kinematic_bounds(self._h, C.byref(lo), C.byref(up)))
        return d.view(self.B, 2 * self.N), (jac.view(self.B, 2 * self.N, 4) if jac is not None else None), (lo.value, up.value)

    def friction_cone(self, Z, mu: float, with_jacobian: bool = True):
        """OPT-IN, nothing in the reference to compare with (it has no friction constraint): the friction pyramid
        |F_x| <= mu F_y of the feet that stand on the ground at each dynamics knot, two linear rows per foot, bounds
        0 <= d < inf.  Returns (d (B, N-1, 4), jac (B, N-1, 4, 2) or None); see qln_eval_friction_cone."""
        t = _torch()
        self._check(Z, self.dims.z_total, "Z")
        n = self.B * (self.N - 1)
        d = t.empty(4 * n, dtype=t.float64, device=self._dev())
        jac = t.empty(8 * n, dtype=t.float64, device=self._dev()) if with_jacobian else None
        _lib.check(_lib.lib().qln_eval_friction_cone(self._h, Z.data_ptr(), float(mu), d.data_ptr(),
                                                     jac.data_ptr() if jac is not None else None))
        return d.view(self.B, self.N - 1, 4), (jac.view(self.B, self.N - 1, 4, 2) if jac is not None else None)

    def constraint_violation(self, c, out=None):
        """Per-problem constraint violation as Ipopt reports it (src/main.ipynb:712) -> (B,) tensor."""
        out = self.new_f() if out is None else out
        _lib.check(_lib.lib().qln_constraint_violation(self._h, self._check(c, self.dims.c_total, "c"),
                                                       self._check(out, self.B, "viol")))
        return out

    def initial_guess(self, out=None):
        """Z0 of the notebook's initial-guess rule (src/main.ipynb:181-198) for every problem, on the device."""
        out = self.new_Z() if out is None else out
        _lib.check(_lib.lib().qln_initial_guess(self._h, self._check(out, self.dims.z_total, "Z")))
        return out

    def sample_drop_states(self, sampler):
        """Redraw x0 of every problem on the device (qln_sample_drop_states; `sampler` from
        problem_gen.drop_state_sampler): bit-identical to problem_gen.make_batch's host draws.  Updates self.x0."""
        _lib.check(_lib.lib().qln_sample_drop_states(self._h, C.byref(sampler)))
        self.x0, _ = self.boundary_states()
        return self.x0

    def perturb_point(self, Z, sampler, sigma: float = 0.05, h_min: float = 0.001, h_max: float = 0.02, redraw_h: bool = False):
        """Z += N(0, sigma^2), h clipped (or redrawn uniformly): the evaluation point of SURVEY.md 8d, on the device."""
        _lib.check(_lib.lib().qln_perturb_point(self._h, C.byref(sampler), self._check(Z, self.dims.z_total, "Z"), float(sigma),
                                                float(h_min), float(h_max), int(redraw_h)))
        return Z

    def boundary_states(self):
        """(x0, xf) as the handle holds them now, (B, 15) each."""
        x0, xf = np.zeros((self.B, n)), np.zeros((self.B, n))
        _lib.check(_lib.lib().qln_get_boundary_states(self._h, x0.ctypes.data, xf.ctypes.data))
        return x0, xf

    def set_lqr_cost(self, Q, R, Qf, dt: float, per_problem: bool = False):
        """The notebook's objective (src/main.ipynb:158-161) built on the device from the handle's own k_trans /
        init_mode / xf; Q, Qf: 15 diagonal entries, R: 5."""
        Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(15)
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(5)
        Qf = np.ascontiguousarray(Qf, dtype=np.float64).reshape(15)
        _lib.check(_lib.lib().qln_set_lqr_cost(self._h, Q.ctypes.data, R.ctypes.data, Qf.ctypes.data, float(dt),
                                               1 if per_problem else 0))
        self.cost_batch = self.B if per_problem else 1

    def get_cost(self):
        cb = C.c_int32()
        _lib.check(_lib.lib().qln_get_cost(self._h, None, C.byref(cb)))
        out = np.zeros((cb.value, self.N, _lib.COST_STRIDE))
        _lib.check(_lib.lib().qln_get_cost(self._h, out.ctypes.data, C.byref(cb)))
        return out[0] if cb.value == 1 else out

    def time_c_and_jac(self, Z, c, vals, warmup: int, iters: int, write_constants: bool = False):
        """HIP-event duration (ms) of each of `iters` launches of the fused hot path."""
        ms = (C.c_float * iters)()
        flags = _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0
        _lib.check(_lib.lib().qln_time_constraint_and_jacobian(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(c, self.dims.c_total, "c"),
            self._check(vals, self.dims.j_total, "vals"), flags, warmup, iters, ms))
        return np.array(ms[:], dtype=np.float64)

    def time_c_and_jac_total(self, Z, c, vals, warmup: int, iters: int, write_constants: bool = False) -> float:
        """Elapsed ms of `iters` launches of the fused hot path issued back to back, one pair of HIP events around all of them."""
        ms = C.c_float()
        flags = _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0
        _lib.check(_lib.lib().qln_time_constraint_and_jacobian_total(
            self._h, self._check(Z, self.dims.z_total, "Z"), self._check(c, self.dims.c_total, "c"),
            self._check(vals, self.dims.j_total, "vals"), flags, warmup, iters, C.byref(ms)))
        return float(ms.value)

    # -- host-pointer (MOI) mode ------------------------------------------------------------------
    def _host_Z(self, Z):
        Z = np.asarray(Z, dtype=np.float64)
        if Z.size == self.B * self.n_nlp and self.z_stride != self.n_nlp:
            buf = np.zeros((self.B, self.z_stride))
            buf[:, : self.n_nlp] = Z.reshape(self.B, self.n_nlp)
            Z = buf
        Z = np.ascontiguousarray(Z.reshape(-1))
        if Z.size != self.dims.z_total:
            raise ValueError(f"Z has {Z.size} entries, expected {self.dims.z_total}")
        return Z

    def eval_f_host(self, Z):
        Z = self._host_Z(Z)
        f = np.zeros(self.B)
        _lib.check(_lib.lib().qln_eval_objective_host(self._h, Z.ctypes.data, f.ctypes.data))
        return f

    def grad_f_host(self, Z):
        Z = self._host_Z(Z)
        g = np.zeros(self.dims.z_total)
        _lib.check(_lib.lib().qln_eval_objective_gradient_host(self._h, Z.ctypes.data, g.ctypes.data))
        return g

    def eval_c_host(self, Z):
        Z = self._host_Z(Z)
        c = np.zeros(self.dims.c_total)
        _lib.check(_lib.lib().qln_eval_constraint_host(self._h, Z.ctypes.data, c.ctypes.data))
        return c

    def jac_c_host(self, Z):
        Z = self._host_Z(Z)
        v = np.zeros(self.dims.j_total)
        _lib.check(_lib.lib().qln_eval_constraint_jacobian_host(self._h, Z.ctypes.data, v.ctypes.data))
        return v

    def jac_c_dense_host(self, Z_b, jac, b: int = 0):
        """Reference-compatible dense Jacobian of problem b: `jac` is a Fortran-ordered
        (m_nlp, n_nlp) float64 array; only the jac_c! write-set is assigned."""
        Z_b = np.ascontiguousarray(Z_b, dtype=np.float64)
        mm, _ = self.problem_dims(b)
        if Z_b.size != self.n_nlp:
            raise ValueError("Z_b must hold one problem's n_nlp entries")
        if not (jac.dtype == np.float64 and jac.flags.f_contiguous and jac.shape == (mm, self.n_nlp)):
            raise ValueError("jac must be a Fortran-ordered float64 (m_nlp, n_nlp) array")
        _lib.check(_lib.lib().qln_eval_constraint_jacobian_dense_host(self._h, b, Z_b.ctypes.data, jac.ctypes.data))
        return jac

    # -- views ------------------------------------------------------------------------------------
    def split_c(self, c_host, b: int = 0):
        mm, _ = self.problem_dims(b)
        return np.asarray(c_host)[self.c_off[b] : self.c_off[b] + mm]

    def split_vals(self, vals_host, b: int = 0):
        _, nz = self.problem_dims(b)
        return np.asarray(vals_host)[self.j_off[b] : self.j_off[b] + nz]


# ----------------------------------------------------------------------------- caller-side pieces of solve()


def variable_bounds(N: int):
    """Variable bounds exactly as `solve()` sets them (src/moi.jl:51-67), 0-based arrays (x_l, x_u).

    Reproduces the reference's quirk Q6 on purpose: its "lower bound of F" lines index
    `22+20(k-1)` and `24+20(k-1)` (1-based), which are yb_{k+1} and x1_{k+1}, not F1y/F2y (17, 19);
    the Ipopt header of the shipped run confirms it ("variables with only lower bounds: 120",
    src/main.ipynb:222).  Pass quirk_Q6=False to bound the forces instead.
    """
    return _variable_bounds(N, True)


def _variable_bounds(N: int, quirk_Q6: bool):
    n_nlp = num_primals(N)
    x_l, x_u = np.full(n_nlp, -np.inf), np.full(n_nlp, np.inf)
    for k in range(1, N + 1):
        x_l[3 + 20 * (k - 1) - 1] = -np.pi / 2  # theta >= -pi/2
        x_u[3 + 20 * (k - 1) - 1] = np.pi / 2
        if k < N:
            x_l[20 + 20 * (k - 1) - 1] = 0.001  # dt
            x_u[20 + 20 * (k - 1) - 1] = 0.02
            if quirk_Q6:
                x_l[22 + 20 * (k - 1) - 1] = 0.0
                x_l[24 + 20 * (k - 1) - 1] = 0.0
            else:
                x_l[17 + 20 * (k - 1) - 1] = 0.0
                x_l[19 + 20 * (k - 1) - 1] = 0.0
    return x_l, x_u


def variable_bounds_forces(N: int):
    """The bounds the comment in src/moi.jl:63 describes (F1y, F2y >= 0) -- not what the reference does."""
    return _variable_bounds(N, False)


def ipopt_initial_point(Z0, x_l, x_u, *, bound_push: float = 1e-2, bound_frac: float = 1e-2,
                        bound_relax_factor: float = 1e-8, constr_viol_tol: float = 1e-6):
    """The point at which Ipopt 3.13 evaluates iteration 0 when `solve()` (src/moi.jl:46-103) hands it `Z0` and the
    variable bounds: Ipopt first relaxes every finite bound by min(constr_viol_tol, bound_relax_factor*max(1,|b|))
    and then pushes the starting point inside the relaxed bounds by bound_push / bound_frac (Ipopt options of the
    same names at their defaults; `solve()` sets constr_viol_tol = c_tol = 1e-6).  With `variable_bounds(N)` -- quirk
    Q6 included -- this reproduces the iteration-0 objective the notebook printed, 1.8380701e+00
    (src/main.ipynb:232): known answer KA6."""
    x = np.array(Z0, dtype=np.float64, copy=True)
    x_l, x_u = np.asarray(x_l, dtype=np.float64), np.asarray(x_u, dtype=np.float64)
    has_l, has_u = np.isfinite(x_l), np.isfinite(x_u)
    with np.errstate(invalid="ignore"):
        lo = np.where(has_l, x_l - np.minimum(constr_viol_tol, bound_relax_factor * np.maximum(1.0, np.abs(x_l))), x_l)
        up = np.where(has_u, x_u + np.minimum(constr_viol_tol, bound_relax_factor * np.maximum(1.0, np.abs(x_u))), x_u)
        p_l = bound_push * np.maximum(1.0, np.abs(lo))
        p_u = bound_push * np.maximum(1.0, np.abs(up))
        both = has_l & has_u
        frac = bound_frac * (up - lo)
        p_l = np.where(both, np.minimum(p_l, frac), p_l)
        p_u = np.where(both, np.minimum(p_u, frac), p_u)
        x = np.where(has_l, np.maximum(x, lo + p_l), x)
        x = np.where(has_u, np.minimum(x, up - p_u), x)
    return x
