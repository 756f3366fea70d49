"""ctypes binding and host mirror of include/qln_multi.h (libqln_multi.so): the multi-GPU layer behind the C ABI.

  MultiNLP   one process, n devices: the batch is cut into contiguous ranges (qln_shard_range), one evaluator handle /
             stream / buffer set per device, no data-path collective, one RCCL gather of the per-problem results at
             the end (BASELINE.json north_star).
  Comm       one process per GPU (a launcher started the ranks): RCCL communicator from a 128-byte id that the host
             passes from rank 0 to the others; gather with per-rank counts, max-reduction, barrier.

Nothing here computes: every method is one call into the library.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libqln_multi.so")

QLN_ERR_COMM = -5
GATHER_F, GATHER_VIOL, GATHER_C = 1, 2, 4
COMM_ID_BYTES = 128

_vp = C.c_void_p
_i64p, _i32p, _ip = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
_dpp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "qln_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, _i64p, _i64p]),
    "qln_comm_get_unique_id": (C.c_int, [_vp]),
    "qln_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "qln_comm_destroy": (C.c_int, [_vp]),
    "qln_comm_rank": (C.c_int, [_vp, _ip, _ip]),
    "qln_comm_exchange_counts": (C.c_int, [_vp, C.c_int64, _i64p]),
    "qln_comm_gather": (C.c_int, [_vp, _vp, C.c_int64, _vp, _i64p, C.c_int, _vp]),
    "qln_comm_max": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "qln_comm_barrier": (C.c_int, [_vp]),
    "qln_multi_create": (C.c_int, [C.POINTER(_lib.QlnBatchDesc), C.c_int, _ip, C.POINTER(_vp)]),
    "qln_multi_create_on_one_device": (C.c_int, [C.POINTER(_lib.QlnBatchDesc), C.c_int, C.c_int, C.POINTER(_vp)]),
    "qln_multi_destroy": (C.c_int, [_vp]),
    "qln_multi_num_devices": (C.c_int, [_vp, _ip]),
    "qln_multi_shard": (C.c_int, [_vp, C.c_int, _ip, _i64p, _i64p, C.POINTER(_vp), _dpp, _dpp, _dpp, _dpp, _dpp]),
    "qln_multi_get_offsets": (C.c_int, [_vp, _i64p, _i64p]),
    "qln_multi_set_Z": (C.c_int, [_vp, _vp]),
    "qln_multi_initial_guess": (C.c_int, [_vp]),
    "qln_multi_set_lqr_cost": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_int]),
    "qln_multi_alloc_vals": (C.c_int, [_vp, C.c_int]),
    "qln_multi_eval_constraint_and_jacobian": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "qln_multi_eval_objective": (C.c_int, [_vp]),
    "qln_multi_constraint_violation": (C.c_int, [_vp]),
    "qln_multi_solve": (C.c_int, [_vp, C.POINTER(_lib.QlnSolveOptions)]),
    "qln_multi_solve_info": (C.c_int, [_vp, _vp]),
    "qln_multi_synchronize": (C.c_int, [_vp]),
    "qln_multi_gather": (C.c_int, [_vp, C.c_uint32, C.c_int]),
    "qln_multi_gathered_to_host": (C.c_int, [_vp, _vp, _vp, _vp]),
    "qln_multi_time_constraint_and_jacobian": (C.c_int, [_vp, C.c_int32, C.c_int32, _fp, C.POINTER(C.c_double)]),
    "qln_multi_plan": (C.c_int, [C.POINTER(_lib.QlnBatchDesc), C.c_int, _vp, _i64p, _i64p]),
}


class QlnShardPlan(C.Structure):
    _fields_ = [
        ("b_begin", C.c_int64),
        ("b_end", C.c_int64),
        ("z_begin", C.c_int64),
        ("cost_begin", C.c_int64),
        ("cost_batch", C.c_int32),
        ("reserved", C.c_int32),
        ("c_displ", C.c_int64),
        ("z_total", C.c_int64),
        ("c_total", C.c_int64),
        ("j_total", C.c_int64),
    ]


def plan(desc, n_devices: int):
    """qln_multi_plan: (list of per-shard dicts, c_off [B], c_total) -- the host bookkeeping of the multi-GPU layer; needs
    no GPU.  `desc` is a _lib.QlnBatchDesc describing the whole batch."""
    plans = (QlnShardPlan * n_devices)()
    c_off = np.zeros(desc.B, dtype=np.int64)
    tot = C.c_int64()
    _lib.check(lib().qln_multi_plan(C.byref(desc), n_devices, C.cast(plans, _vp), c_off.ctypes.data_as(_i64p), C.byref(tot)))
    return [{f: getattr(p, f) for f, _ in QlnShardPlan._fields_ if f != "reserved"} for p in plans], c_off, tot.value

_mlib = None


def lib() -> C.CDLL:
    """Load libqln_multi.so (after libqln_hip.so, which it is linked against)."""
    global _mlib
    if _mlib is None:
        _lib.lib()  # torch's HIP/RCCL runtime first, then the evaluator (see _lib.lib)
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make -C quadruped_landing_amd/csrc all`")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _mlib = L
    return _mlib


def shard_range(n_problems: int, rank: int, world: int):
    a, b = C.c_int64(), C.c_int64()
    _lib.check(lib().qln_shard_range(n_problems, rank, world, C.byref(a), C.byref(b)))
    return a.value, b.value


class Comm:
    """One rank of a job with one process per GPU (qln_comm_*)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of Comm.unique_id()")
        self.rank, self.world, self.device = rank, world, device
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, COMM_ID_BYTES)
        _lib.check(lib().qln_comm_init_rank(buf, rank, world, device, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _lib.check(lib().qln_comm_get_unique_id(buf))
        return buf.raw

    def exchange_counts(self, count: int):
        counts = (C.c_int64 * self.world)()
        _lib.check(lib().qln_comm_exchange_counts(self._h, int(count), counts))
        return list(counts)

    def gather(self, t, root: int = 0, stream=None):
        """Gather 1-D float64 device tensors of possibly different lengths to `root` over RCCL, ordered on `stream`
        (default: torch's current stream).  Returns (gathered tensor on the root / None elsewhere, per-rank counts)."""
        import torch

        t = t.contiguous()
        counts = self.exchange_counts(t.numel())
        s = torch.cuda.current_stream(t.device) if stream is None else stream
        out = torch.empty(int(sum(counts)), dtype=torch.float64, device=t.device) if self.rank == root else None
        arr = (C.c_int64 * self.world)(*counts)
        _lib.check(lib().qln_comm_gather(self._h, t.data_ptr(), t.numel(), out.data_ptr() if out is not None else None,
                                         arr, root, int(getattr(s, "cuda_stream", s))))
        return out, counts

    def max(self, value: float) -> float:
        v = C.c_double(float(value))
        _lib.check(lib().qln_comm_max(self._h, C.byref(v)))
        return v.value

    def barrier(self):
        _lib.check(lib().qln_comm_barrier(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().qln_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiNLP:
    """The batch of `HybridNLP` sharded over the devices of one process (qln_multi_*).  Constructor arguments as for
    HybridNLP, plus `devices` (list of HIP ordinals)."""

    def __init__(self, model, obj, init_mode, k_trans, N: int, x0, xf, *, devices, z_stride: int = 0, align: int = 16,
                 jac_format: str = "dense_blocks", one_device: bool = False):
        """`one_device=True`: the rehearsal of the n > 1 paths on one GPU (qln_multi_create_on_one_device) -- `devices` then
        lists the SAME ordinal once per shard; no RCCL clique, the gather's exchange is device copies."""
        from .nlp import JAC_FORMATS, n

        x0 = np.asarray(x0, dtype=np.float64)
        B = x0.shape[0]
        self.B, self.N = B, int(N)
        self.x0 = np.ascontiguousarray(np.broadcast_to(x0, (B, n)))
        self.xf = np.ascontiguousarray(np.broadcast_to(np.asarray(xf, dtype=np.float64), (B, n)))
        self.k_trans = np.ascontiguousarray(np.broadcast_to(np.asarray(k_trans, dtype=np.int32), (B,)))
        self.init_mode = np.ascontiguousarray(np.broadcast_to(np.asarray(init_mode, dtype=np.int32), (B,)))
        self.obj = None if obj is None else np.ascontiguousarray(obj, dtype=np.float64)
        cost_batch = 1 if (self.obj is None or self.obj.ndim == 2) else B
        d = _lib.QlnBatchDesc()
        d.B, d.N = B, self.N
        d.model = _lib.QlnModel(model.g, model.mb, model.mf, model.lb, model.l1, model.l2)
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        d.k_trans, d.init_mode = self.k_trans.ctypes.data_as(ip), self.init_mode.ctypes.data_as(ip)
        d.x0, d.xf = self.x0.ctypes.data_as(dp), self.xf.ctypes.data_as(dp)
        d.cost = self.obj.ctypes.data_as(dp) if self.obj is not None else None
        d.cost_batch, d.z_stride, d.align, d.jac_format = cost_batch, int(z_stride), int(align), JAC_FORMATS[jac_format]
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        if one_device:
            if len(set(devices)) != 1:
                raise ValueError("one_device=True: `devices` lists one ordinal, once per shard")
            _lib.check(lib().qln_multi_create_on_one_device(C.byref(d), len(devices), int(devices[0]), C.byref(h)))
        else:
            _lib.check(lib().qln_multi_create(C.byref(d), len(devices), devs, C.byref(h)))
        self._h = h
        self.n_devices = len(devices)
        self.n_nlp = 20 * self.N - 5
        self.z_stride = int(z_stride) or self.n_nlp
        self.c_off = np.zeros(B, dtype=np.int64)
        tot = C.c_int64()
        _lib.check(lib().qln_multi_get_offsets(self._h, self.c_off.ctypes.data_as(_i64p), C.byref(tot)))
        self.c_total = tot.value

    def shard(self, r: int):
        """dict(device, begin, end, handle, Z, c, vals, f, viol) -- raw addresses, for use with the single-GPU ABI."""
        dev, lo, hi = C.c_int(), C.c_int64(), C.c_int64()
        h = C.c_void_p()
        p = [C.c_void_p() for _ in range(5)]
        _lib.check(lib().qln_multi_shard(self._h, r, C.byref(dev), C.byref(lo), C.byref(hi), C.byref(h), *[C.byref(x) for x in p]))
        return dict(device=dev.value, begin=lo.value, end=hi.value, handle=h, Z=p[0].value, c=p[1].value, vals=p[2].value,
                    f=p[3].value, viol=p[4].value)

    def shard_tensor(self, r: int, name: str):
        """A torch view (no copy, not owning) of shard r's device buffer `name` in {Z, c, vals, f, viol}; valid while
        this object lives."""
        import torch

        s = self.shard(r)
        dims = _lib.QlnDims()
        _lib.check(_lib.lib().qln_get_dims(s["handle"], C.byref(dims)))
        numel = {"Z": dims.z_total, "c": dims.c_total, "vals": dims.j_total, "f": dims.B, "viol": dims.B}[name]
        if not s[name]:
            raise ValueError(f"shard {r} has no {name} buffer yet")

        class _View:
            pass

        v = _View()
        v.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f8", "data": (int(s[name]), False), "version": 2}
        v._owner = self
        return torch.as_tensor(v, device=torch.device("cuda", s["device"]))

    def set_Z(self, Z_host):
        Z = np.zeros((self.B, self.z_stride))
        Z[:, : self.n_nlp] = np.asarray(Z_host, dtype=np.float64).reshape(self.B, -1)[:, : self.n_nlp]
        _lib.check(lib().qln_multi_set_Z(self._h, Z.ctypes.data))  # returns when the copies are complete

    def sample_drop_states(self, samplers):
        """qln_sample_drop_states on every shard's handle (`samplers`: one qln_drop_state_sampler per shard): the synthetic
        drop states generated where they are used, as the one-process-per-GPU driver does with its rank's sampler."""
        for r, smp in enumerate(samplers):
            _lib.check(_lib.lib().qln_sample_drop_states(self.shard(r)["handle"], C.byref(smp)))

    def perturb_point(self, samplers, sigma: float = 0.05, h_min: float = 0.001, h_max: float = 0.02, redraw_h: bool = False):
        """qln_perturb_point on every shard's Z (after initial_guess)."""
        for r, smp in enumerate(samplers):
            sh = self.shard(r)
            _lib.check(_lib.lib().qln_perturb_point(sh["handle"], C.byref(smp), sh["Z"], float(sigma), float(h_min), float(h_max),
                                                    int(redraw_h)))

    def initial_guess(self):
        _lib.check(lib().qln_multi_initial_guess(self._h))

    def set_lqr_cost(self, Q, R, Qf, dt: float, per_problem: bool = False):
        Q, R, Qf = (np.ascontiguousarray(a, dtype=np.float64) for a in (Q, R, Qf))
        _lib.check(lib().qln_multi_set_lqr_cost(self._h, Q.ctypes.data, R.ctypes.data, Qf.ctypes.data, float(dt), int(per_problem)))

    def alloc_vals(self, placed: bool = False):
        _lib.check(lib().qln_multi_alloc_vals(self._h, int(placed)))

    def eval_c_and_jac(self, with_jacobian: bool = True, write_constants: bool = False):
        _lib.check(lib().qln_multi_eval_constraint_and_jacobian(self._h, int(with_jacobian),
                                                                _lib.QLN_JAC_WRITE_CONSTANTS if write_constants else 0))

    def eval_f(self):
        _lib.check(lib().qln_multi_eval_objective(self._h))

    def constraint_violation(self):
        _lib.check(lib().qln_multi_constraint_violation(self._h))

    def solve(self, **options):
        """qln_solve on every shard, in place on the shards' Z; returns the (B, 16) info array (global problem order)."""
        opt = _lib.QlnSolveOptions()
        _lib.check(_lib.lib().qln_solve_default_options(C.byref(opt)))
        for k, v in options.items():
            if not hasattr(opt, k):
                raise TypeError(f"unknown solve option {k!r}")
            setattr(opt, k, v)
        _lib.check(lib().qln_multi_solve(self._h, C.byref(opt)))
        info = np.zeros((self.B, _lib.SOLVE_INFO_STRIDE))
        _lib.check(lib().qln_multi_solve_info(self._h, info.ctypes.data))
        return info

    def synchronize(self):
        _lib.check(lib().qln_multi_synchronize(self._h))

    def gather(self, what: int = GATHER_F | GATHER_VIOL, root_shard: int = 0):
        _lib.check(lib().qln_multi_gather(self._h, what, root_shard))

    def gathered(self, f: bool = True, viol: bool = True, c: bool = False):
        out = [np.zeros(self.B) if f else None, np.zeros(self.B) if viol else None, np.zeros(self.c_total) if c else None]
        _lib.check(lib().qln_multi_gathered_to_host(self._h, *[a.ctypes.data if a is not None else None for a in out]))
        return out

    def time_c_and_jac(self, warmup: int, iters: int):
        """(ms per device from HIP events around its `iters` launches, host wall ms from the common release of the per-device
        issue threads until every device is idle)"""
        ms = (C.c_float * self.n_devices)()
        wall = C.c_double()
        _lib.check(lib().qln_multi_time_constraint_and_jacobian(self._h, warmup, iters, ms, C.byref(wall)))
        return np.array(ms[:], dtype=np.float64), wall.value

    def close(self):
        if getattr(self, "_h", None):
            lib().qln_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
