"""The C-ABI library loads without a GPU, exports every symbol include/qln_evaluator.h declares, and
fails loudly (no CPU fallback) when asked to create a handle without a device."""
import ctypes as C
import os
import re

import pytest

from quadruped_landing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "qln_evaluator.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qln_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert b"gfx950" in L.qln_version()


def test_struct_layouts_match_the_header():
    # qln_model: 6 doubles; qln_batch_desc and qln_dims as declared (natural alignment)
    assert C.sizeof(_lib.QlnModel) == 48
    assert C.sizeof(_lib.QlnBatchDesc) == 8 + 48 + 5 * 8 + 8 + 8 + 8
    assert C.sizeof(_lib.QlnDims) == 6 * 4 + 4 * 8


def test_argument_validation_needs_no_gpu():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.qln_create(None, 0, C.byref(h)) == _lib.QLN_ERR_INVALID_ARGUMENT
    d = _lib.QlnBatchDesc()
    d.B, d.N = 1, 1
    assert L.qln_create(C.byref(d), 0, C.byref(h)) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert b"N must be >= 2" in L.qln_last_error()
    assert L.qln_get_dims(None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert L.qln_destroy(None) == _lib.QLN_OK


def test_no_cpu_fallback_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import quadruped_landing_amd as Q
    from quadruped_landing_amd import problem_gen as PG

    b = PG.make_batch(2, 5, 3, 1)
    with pytest.raises(_lib.QlnError) as ei:
        Q.HybridNLP(b.model, b.obj, b.init_mode, b.k_trans, b.N, b.x0, b.xf)
    assert ei.value.code in (_lib.QLN_ERR_NO_DEVICE, _lib.QLN_ERR_HIP)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "quadruped_landing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), f


def test_c_host_program_compiles_and_links_against_the_abi(tmp_path):
    """tests/host_c/moi_host.c (the Julia binding's call sequence from plain C) builds with gcc -Werror against
    include/qln_evaluator.h and links libqln_hip.so; it is RUN by the GPU tests (test_gpu_baseline_configs.py)."""
    from tests.helpers import build_c_host

    exe = build_c_host(tmp_path)
    assert os.path.exists(exe)


def _declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qln_[A-Za-z_]+)\s*\(", src)))


def test_multi_gpu_header_binding_and_library_agree():
    """include/qln_multi.h <-> quadruped_landing_amd/multi.py <-> libqln_multi.so (loads without a GPU)."""
    from quadruped_landing_amd import multi

    assert _declared_in("qln_multi.h") == sorted(multi.SIGNATURES)
    L = multi.lib()
    for name in multi.SIGNATURES:
        assert hasattr(L, name), name


def test_the_c_and_python_partitioning_rules_are_the_same():
    from quadruped_landing_amd import distributed as D, multi

    for n in (1, 7, 11, 64, 65536, 524288, 524289):
        for w in (1, 2, 3, 8):
            if n >= w:
                assert [multi.shard_range(n, r, w) for r in range(w)] == [D.shard_range(n, r, w) for r in range(w)]


def test_multi_create_without_a_gpu_fails_loudly():
    import numpy as np
    from quadruped_landing_amd import multi, problem_gen as PG

    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    b = PG.make_batch(4, 8, 3, 1, seed=0)
    with pytest.raises(_lib.QlnError) as e:
        multi.MultiNLP(b.model, b.obj, b.init_mode, b.k_trans, b.N, b.x0, b.xf, devices=[0])
    assert e.value.code == _lib.QLN_ERR_NO_DEVICE


def test_placed_address_space_accounting_needs_no_gpu():
    """qln_vals_placed_address_space: what placed allocations have retired of the process's address space, and the cap."""
    L = _lib.lib()
    a, b = C.c_int64(-1), C.c_int64(-1)
    assert L.qln_vals_placed_address_space(C.byref(a), C.byref(b)) == _lib.QLN_OK
    assert a.value >= 0 and b.value == 64 << 40
    assert L.qln_vals_placed_address_space(None, None) == _lib.QLN_OK
