"""ctypes binding of the C ABI in include/qln_evaluator.h (libqln_hip.so).

There is no CPU fallback: if the HIP library is missing this module raises, and
qln_create fails with QLN_ERR_NO_DEVICE when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QLN_LIB_PATH lets the A/B measurement scripts under bench/ load another in-tree build of the same ABI
LIB_PATH = os.environ.get("QLN_LIB_PATH") or os.path.join(_HERE, "csrc", "libqln_hip.so")

QLN_OK = 0
QLN_ERR_INVALID_ARGUMENT = -1
QLN_ERR_HIP = -2
QLN_ERR_NO_DEVICE = -3
QLN_ERR_UNSUPPORTED = -4
QLN_JAC_WRITE_CONSTANTS = 1
QLN_JAC_FORMAT_DENSE_BLOCKS = 0
QLN_JAC_FORMAT_STRUCTURAL = 1

NX, NU, NZ, COST_STRIDE = 15, 5, 20, 41
GN_INFO_STRIDE = 8


class QlnModel(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("g", "mb", "mf", "lb", "l1", "l2")]


class QlnBatchDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("N", C.c_int32),
        ("model", QlnModel),
        ("k_trans", C.POINTER(C.c_int32)),
        ("init_mode", C.POINTER(C.c_int32)),
        ("x0", C.POINTER(C.c_double)),
        ("xf", C.POINTER(C.c_double)),
        ("cost", C.POINTER(C.c_double)),
        ("cost_batch", C.c_int32),
        ("z_stride", C.c_int64),
        ("align", C.c_int32),
        ("jac_format", C.c_int32),
    ]


class QlnDims(C.Structure):
    _fields_ = [
        ("B", C.c_int32),
        ("N", C.c_int32),
        ("n_nlp", C.c_int32),
        ("m_nlp_max", C.c_int32),
        ("nnz_max", C.c_int32),
        ("nnz_dynamic", C.c_int32),
        ("z_stride", C.c_int64),
        ("z_total", C.c_int64),
        ("c_total", C.c_int64),
        ("j_total", C.c_int64),
    ]


class QlnSolveOptions(C.Structure):
    _fields_ = [
        ("max_outer", C.c_int32),
        ("max_inner", C.c_int32),
        ("tol_violation", C.c_double),
        ("inner_tol", C.c_double),
        ("rho0", C.c_double),
        ("rho_factor", C.c_double),
        ("rho_max", C.c_double),
        ("h_min", C.c_double),
        ("h_max", C.c_double),
        ("theta_min", C.c_double),
        ("theta_max", C.c_double),
        ("q6_bounds", C.c_int32),
        ("exact_h_gradient", C.c_int32),
        ("h_prox", C.c_double),
        ("rescue_outer", C.c_int32),
    ]


class QlnDropStateSampler(C.Structure):
    _fields_ = [
        ("pcg_state", C.c_uint64 * 2),
        ("pcg_inc", C.c_uint64 * 2),
        ("stream_offset", C.c_int64),
        ("x0_template", C.c_double * 15),
        ("theta_deg", C.c_double * 2),
        ("y2", C.c_double * 2),
        ("drop_height", C.c_double * 2),
        ("omega", C.c_double * 2),
        ("two_g", C.c_double),
    ]


SOLVE_INFO_STRIDE = 16


class QlnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qln error {code}: {msg}")
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol include/qln_evaluator.h declares
_vp, _dp = C.c_void_p, C.c_void_p  # device or host double* passed as raw addresses
_i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
SIGNATURES = {
    "qln_last_error": (C.c_char_p, []),
    "qln_set_last_error": (C.c_int, [C.c_int, C.c_char_p]),
    "qln_version": (C.c_char_p, []),
    "qln_create": (C.c_int, [C.POINTER(QlnBatchDesc), C.c_int, C.POINTER(_vp)]),
    "qln_destroy": (C.c_int, [_vp]),
    "qln_layout": (C.c_int, [C.POINTER(QlnBatchDesc), C.POINTER(QlnDims), _i64p, _i64p]),
    "qln_set_stream": (C.c_int, [_vp, _vp]),
    "qln_synchronize": (C.c_int, [_vp]),
    "qln_get_dims": (C.c_int, [_vp, C.POINTER(QlnDims)]),
    "qln_get_offsets": (C.c_int, [_vp, _i64p, _i64p]),
    "qln_problem_dims": (C.c_int, [_vp, C.c_int32, _i32p, _i32p]),
    "qln_problem_nnz_dynamic": (C.c_int, [_vp, C.c_int32, _i32p]),
    "qln_constraint_index_ranges": (C.c_int, [_vp, C.c_int32, _i32p]),
    "qln_constraint_bounds": (C.c_int, [_vp, C.c_int32, _dp, _dp]),
    "qln_jacobian_structure": (C.c_int, [_vp, C.c_int32, _i32p, _i32p]),
    "qln_eval_objective": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_objective_gradient": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_constraint": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_constraint_jacobian": (C.c_int, [_vp, _dp, _dp, C.c_uint32]),
    "qln_eval_constraint_and_jacobian": (C.c_int, [_vp, _dp, _dp, _dp, C.c_uint32]),
    "qln_eval_all": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, C.c_uint32]),
    "qln_eval_objective_and_constraint": (C.c_int, [_vp, _dp, _dp, _dp]),
    "qln_jacobian_init_constants": (C.c_int, [_vp, _dp]),
    "qln_eval_constraint_jvp": (C.c_int, [_vp, _dp, _dp, _dp]),
    "qln_eval_constraint_vjp": (C.c_int, [_vp, _dp, _dp, _dp]),
    "qln_gauss_newton_step": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int32, C.c_double, _dp, _dp, _dp]),
    "qln_eval_kinematic_constraint": (C.c_int, [_vp, _dp, _dp, _dp]),
    "qln_kinematic_bounds": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "qln_eval_friction_cone": (C.c_int, [_vp, _dp, C.c_double, _dp, _dp]),
    "qln_constraint_violation": (C.c_int, [_vp, _dp, _dp]),
    "qln_solve_default_options": (C.c_int, [C.POINTER(QlnSolveOptions)]),
    "qln_variable_bounds": (C.c_int, [C.c_int32, C.POINTER(QlnSolveOptions), _dp, _dp]),
    "qln_solve": (C.c_int, [_vp, _dp, C.POINTER(QlnSolveOptions), _dp]),
    "qln_solve_host": (C.c_int, [_vp, _dp, C.POINTER(QlnSolveOptions), _dp]),
    "qln_initial_guess": (C.c_int, [_vp, _dp]),
    "qln_sample_drop_states": (C.c_int, [_vp, C.POINTER(QlnDropStateSampler)]),
    "qln_sample_bounded_integers": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int64, C.c_int32, C.c_int32, C.c_int64,
                                            _i32p, _i64p]),
    "qln_perturb_point": (C.c_int, [_vp, C.POINTER(QlnDropStateSampler), _dp, C.c_double, C.c_double, C.c_double, C.c_int]),
    "qln_get_boundary_states": (C.c_int, [_vp, _dp, _dp]),
    "qln_set_lqr_cost": (C.c_int, [_vp, _dp, _dp, _dp, C.c_double, C.c_int]),
    "qln_get_cost": (C.c_int, [_vp, _dp, _i32p]),
    "qln_eval_objective_host": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_objective_gradient_host": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_constraint_host": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_constraint_jacobian_host": (C.c_int, [_vp, _dp, _dp]),
    "qln_eval_constraint_jacobian_dense_host": (C.c_int, [_vp, C.c_int32, _dp, _dp]),
    "qln_vals_alloc_placed": (C.c_int, [_vp, _dp, _dp, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "qln_vals_free_placed": (C.c_int, [_vp, _dp]),
    "qln_vals_alloc_placed_budget": (C.c_int, [_vp, _dp, _dp, C.c_int64, C.POINTER(_vp), C.POINTER(C.c_float)]),
    "qln_vals_placed_address_space": (C.c_int, [_i64p, _i64p]),
    "qln_vals_placed_info": (C.c_int, [_vp, _dp, _i64p, _i64p, _i64p]),
    "qln_time_constraint_and_jacobian": (C.c_int, [_vp, _dp, _dp, _dp, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    "qln_time_constraint_and_jacobian_total": (C.c_int, [_vp, _dp, _dp, _dp, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
}


def lib() -> C.CDLL:
    """Load libqln_hip.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and not os.environ.get("QLN_LIB_PATH"):
            # a source checkout without the built artefact: compile it in-tree once (hipcc cross-compiles gfx950
            # without a GPU); anything else is an error -- there is no CPU fallback
            import shutil
            import subprocess

            csrc = os.path.join(_HERE, "csrc")
            if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
                subprocess.call(["make", "-C", csrc, "libqln_hip.so"], stdout=subprocess.DEVNULL)
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C quadruped_landing_amd/csrc` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # The process must hold ONE HIP runtime: PyTorch-ROCm ships its own libamdhip64 and the handle's buffers are
        # torch tensors, so torch's copy is loaded first and libqln_hip.so binds to it (loading this library before
        # torch would bring in /opt/rocm's copy, and the second runtime to initialise then sees no device).
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != QLN_OK:
        raise QlnError(rc, lib().qln_last_error().decode())
