"""The host bookkeeping of the multi-GPU layer (include/qln_multi.h: qln_multi_plan), checked WITHOUT a GPU for more than
one shard -- the pool gives one GPU per box, so the n > 1 branches of qln_multi_create / set_Z / gather have never run on
hardware; what they compute on the host is pinned here instead: shard ranges, the cuts of the batch's host arrays (Z rows,
per-problem cost tables), every shard's displacement in the gathered constraint vector and every problem's offset in
it, for uniform and ragged batches, against (a) the single-handle layout (qln_layout of the whole batch / of each shard's
slice built independently in numpy) and (b) the reference's size formula m_nlp = 18N - k_trans + 16 (src/nlp.jl:48-63).
qln_multi_create works from exactly this plan (and refuses to go on if a shard handle's layout differs from it)."""
import ctypes as C

import numpy as np
import pytest

from quadruped_landing_amd import _lib, distributed as D, multi


def _desc(B, N, kt, im, *, cost_batch=1, z_stride=0, align=0, fmt=0):
    d = _lib.QlnBatchDesc()
    d.B, d.N = B, N
    keep = [np.ascontiguousarray(kt, dtype=np.int32), np.ascontiguousarray(im, dtype=np.int32)]
    ip = C.POINTER(C.c_int32)
    d.k_trans, d.init_mode = keep[0].ctypes.data_as(ip), keep[1].ctypes.data_as(ip)
    d.cost_batch, d.z_stride, d.align, d.jac_format = cost_batch, z_stride, align, fmt
    d._keep = keep
    return d


def _layout(d):
    dims = _lib.QlnDims()
    c_off, j_off = np.zeros(d.B, dtype=np.int64), np.zeros(d.B, dtype=np.int64)
    i64 = C.POINTER(C.c_int64)
    _lib.check(_lib.lib().qln_layout(C.byref(d), C.byref(dims), c_off.ctypes.data_as(i64), j_off.ctypes.data_as(i64)))
    return dims, c_off, j_off


@pytest.mark.parametrize("B,N,ragged", [(11, 9, True), (12, 40, False), (7, 80, True), (3, 5, True)])
@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("opts", [dict(), dict(z_stride=1700, align=3, cost_batch="B"), dict(align=1, fmt=1)])
def test_shard_plan_against_the_single_handle_layout(B, N, ragged, n, opts):
    rng = np.random.default_rng(B * 100 + N)
    kt = rng.integers(1, N + 2, size=B) if ragged else np.full(B, 14 if N >= 14 else 2)
    im = rng.integers(1, 3, size=B)
    o = dict(opts)
    if o.get("z_stride"):
        o["z_stride"] = max(o["z_stride"], 20 * N - 5)
    if o.get("cost_batch") == "B":
        o["cost_batch"] = B
    d = _desc(B, N, kt, im, **o)
    plans, c_off, c_total = multi.plan(d, n)
    align = o.get("align", 0) or 16
    z_stride = o.get("z_stride", 0) or 20 * N - 5
    displ = 0
    for r, p in enumerate(plans):
        lo, hi = D.shard_range(B, r, n)
        assert (p["b_begin"], p["b_end"]) == (lo, hi) == multi.shard_range(B, r, n)
        assert p["z_begin"] == lo * z_stride                                   # the cut of the host Z
        per_problem = o.get("cost_batch", 1) == B and B > 1
        assert p["cost_begin"] == (lo * N * 41 if per_problem else 0)          # the cut of a per-problem cost table
        assert p["cost_batch"] == (hi - lo if per_problem else 1)
        assert p["c_displ"] == displ
        # the shard's own handle will lay its slice out like this (an independent descriptor of the slice)
        sd = _desc(hi - lo, N, kt[lo:hi], im[lo:hi], cost_batch=(hi - lo if per_problem else 1), z_stride=o.get("z_stride", 0),
                   align=o.get("align", 0), fmt=o.get("fmt", 0))
        dims, loc_c, loc_j = _layout(sd)
        assert (p["z_total"], p["c_total"], p["j_total"]) == (dims.z_total, dims.c_total, dims.j_total)
        assert np.array_equal(c_off[lo:hi], displ + loc_c)
        # ... which is the exclusive scan of the reference's m_nlp, rounded up to `align`
        m_nlp = 18 * N - kt[lo:hi] + 16
        want, e = [], 0
        for m in m_nlp:
            e = -(-e // align) * align
            want.append(e)
            e += int(m)
        assert np.array_equal(loc_c, want) and dims.c_total == e
        displ += dims.c_total
    assert c_total == displ
    if n == 1:  # one shard = the single-handle layout of the whole batch
        dims, whole_c, _ = _layout(d)
        assert np.array_equal(c_off, whole_c) and c_total == dims.c_total
    # the gathered vector has every problem's rows exactly once, in order, no overlap
    m_all = 18 * N - np.asarray(kt) + 16
    assert np.all(c_off[1:] >= c_off[:-1] + m_all[:-1]) and c_off[-1] + m_all[-1] <= c_total


def test_plan_and_layout_argument_checks_need_no_gpu():
    L, M = _lib.lib(), multi.lib()
    d = _desc(4, 10, [3, 4, 5, 6], [1, 2, 1, 2])
    plans = (multi.QlnShardPlan * 8)()
    assert M.qln_multi_plan(C.byref(d), 5, C.cast(plans, C.c_void_p), None, None) == _lib.QLN_ERR_INVALID_ARGUMENT  # B < n
    assert M.qln_multi_plan(C.byref(d), 0, C.cast(plans, C.c_void_p), None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert M.qln_multi_plan(None, 2, C.cast(plans, C.c_void_p), None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    bad = _desc(4, 10, [3, 4, 12, 6], [1, 2, 1, 2])  # k_trans > N + 1 in the second shard
    assert M.qln_multi_plan(C.byref(bad), 2, C.cast(plans, C.c_void_p), None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert b"k_trans out of range" in L.qln_last_error()
    dims = _lib.QlnDims()
    assert L.qln_layout(C.byref(d), None, None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert L.qln_layout(C.byref(d), C.byref(dims), None, None) == _lib.QLN_OK and dims.c_total > 0


def test_shard_plan_property_random_batches():
    """hypothesis: any batch size, horizon, shard count, alignment and stride -- the plan's offsets tile the gathered constraint
    vector without overlap, every shard's local offsets are the single-handle layout of its slice, and the whole is a
    permutation-free concatenation of the shards (problem order preserved)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(B=st.integers(1, 70), N=st.integers(2, 90), n=st.integers(1, 8), align=st.sampled_from([0, 1, 2, 3, 16, 32]),
           pad=st.integers(0, 7), fmt=st.sampled_from([0, 1]), seed=st.integers(0, 10**6))
    def check(B, N, n, align, pad, fmt, seed):
        n = min(n, B)
        rng = np.random.default_rng(seed)
        kt, im = rng.integers(1, N + 2, size=B), rng.integers(1, 3, size=B)
        zs = (20 * N - 5 + pad) if pad else 0
        d = _desc(B, N, kt, im, z_stride=zs, align=align, fmt=fmt)
        plans, c_off, c_total = multi.plan(d, n)
        m_all = 18 * N - kt + 16
        assert plans[0]["b_begin"] == 0 and plans[-1]["b_end"] == B
        assert all(plans[r]["b_end"] == plans[r + 1]["b_begin"] for r in range(n - 1))
        end = 0
        for p in plans:
            lo, hi = p["b_begin"], p["b_end"]
            sd = _desc(hi - lo, N, kt[lo:hi], im[lo:hi], z_stride=zs, align=align, fmt=fmt)
            dims, loc_c, _ = _layout(sd)
            assert p["c_displ"] == end and np.array_equal(c_off[lo:hi], end + loc_c)
            assert (p["z_total"], p["c_total"], p["j_total"]) == (dims.z_total, dims.c_total, dims.j_total)
            assert p["z_begin"] == lo * (zs or 20 * N - 5)
            end += dims.c_total
        assert c_total == end
        assert np.all(c_off[1:] >= c_off[:-1] + m_all[:-1]) and c_off[-1] + m_all[-1] <= c_total

    check()
