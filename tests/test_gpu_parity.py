"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle.

Tolerances (BASELINE.json north_star): bit-exact constraint indexing / sparsity; <= 1e-8 relative on
FP64 dynamics and Jacobian entries.  The value path (constraints, objective) follows the reference's
operation order without FMA contraction, so it is additionally required to be within a few ulp.
"""
import numpy as np
import pytest

from tests.helpers import oracle_batch, rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-8  # north_star: "within 1e-8 relative on FP64 dynamics/Jacobian entries"


def _fuzz_examples(default):
    """hypothesis examples per property test: the suite's default, or QLN_FUZZ_EXAMPLES for a one-off campaign on the GPU box"""
    import os

    return int(os.environ.get("QLN_FUZZ_EXAMPLES", default))


def _fuzz_fixed():
    """the suite's own run draws the same examples every time (a red suite must be reproducible); a campaign draws fresh ones"""
    import os

    return "QLN_FUZZ_EXAMPLES" not in os.environ


def _gpu_eval(batch, **kw):
    import torch
    from quadruped_landing_amd import HybridNLP

    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, **kw)
    Z = nlp.upload_Z(batch.Z)
    c = torch.full((nlp.dims.c_total,), float("nan"), dtype=torch.float64, device="cuda")
    v = torch.full((nlp.dims.j_total,), float("nan"), dtype=torch.float64, device="cuda")
    nlp.eval_c_and_jac(Z, c, v, write_constants=True)
    f = nlp.eval_f(Z)
    g = nlp.grad_f(Z)
    torch.cuda.synchronize()
    return nlp, c.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()


def _compare(batch, nlp, c, v, f, g):
    ref = oracle_batch(batch, nlp, want_f=True, want_grad=True)
    # every slot the oracle assigns is assigned by the GPU and vice versa (padding stays NaN)
    assert np.array_equal(np.isnan(c), np.isnan(ref["c"]))
    assert np.array_equal(np.isnan(v), np.isnan(ref["vals"]))
    # exact-zero pattern of the Jacobian values = structural sparsity
    okv = ~np.isnan(v)
    assert np.array_equal(v[okv] == 0, ref["vals"][okv] == 0)
    scale_c = 1.0  # residuals are differences of O(1) states: measure against max(|ref|, 1)
    ec = rel_err(c, ref["c"], floor=scale_c)
    ev = rel_err(v, ref["vals"], floor=1e-300)
    ef = rel_err(f, ref["f"])
    eg = rel_err(g.reshape(batch.B, -1)[:, : nlp.n_nlp], ref["grad"].reshape(batch.B, -1)[:, : nlp.n_nlp], floor=1e-300)
    print(f"B={batch.B} N={batch.N}: rel err c={ec:.3e} J={ev:.3e} f={ef:.3e} grad={eg:.3e}")
    assert ec <= RTOL and ev <= RTOL and ef <= RTOL and eg <= RTOL
    return ec, ev, ef, eg


@pytest.mark.parametrize("B,N,kt,im", [(1, 40, 14, 1), (64, 40, 14, 1), (33, 40, 14, 2), (5, 61, 21, 1), (7, 2, 2, 1),
                                       (3, 3, 2, 2), (4, 66, 30, 1), (2, 130, 100, 2)])
def test_uniform_batches(B, N, kt, im):
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(B, N, kt, im, seed=B + N)
    out = _gpu_eval(batch)
    _compare(batch, *out)


@pytest.mark.parametrize("B,N", [(257, 80), (40, 17), (16, 200)])
def test_ragged_batches(B, N):
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(B, N, seed=7, ragged=True)
    out = _gpu_eval(batch)
    _compare(batch, *out)


def test_k_trans_extremes():
    from quadruped_landing_amd import problem_gen as PG

    N = 12
    batch = PG.make_batch(6, N, seed=3, ragged=True)
    batch.k_trans[:] = [1, 2, N - 1, N, N + 1, 5]
    batch.init_mode[:] = [1, 2, 1, 2, 1, 2]
    out = _gpu_eval(batch)
    _compare(batch, *out)


def test_value_path_rounds_like_the_reference():
    """No FMA contraction + reference operation order => constraints and objective agree with the
    oracle to the last bit except where libm and the device sin() differ (clearance rows)."""
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(128, 40, 14, 1, seed=11)
    nlp, c, v, f, g = _gpu_eval(batch)
    ref = oracle_batch(batch, nlp, want_f=True, want_grad=True)
    mm, _ = nlp.problem_dims(0)
    ci = nlp.cinds(0)
    neq = ci[5][1]
    for b in range(batch.B):
        a = c[nlp.c_off[b] : nlp.c_off[b] + neq]
        r = ref["c"][nlp.c_off[b] : nlp.c_off[b] + neq]
        assert np.array_equal(a, r), f"equality rows of problem {b} differ"
    assert np.array_equal(f, ref["f"])
    assert np.array_equal(g, ref["grad"])


def test_strided_layout_and_alignment_options():
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(9, 40, 14, 1, seed=5)
    for kw in (dict(z_stride=800, align=16), dict(z_stride=0, align=1), dict(z_stride=797, align=2)):
        out = _gpu_eval(batch, **kw)
        _compare(batch, *out)


def test_notebook_known_answers_on_gpu(golden_dir):
    """KA1/KA2 of the notebook run (src/main.ipynb:710,712) through the HIP path."""
    import os
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    nb = PG.notebook_problem()
    nb.Z = np.loadtxt(os.path.join(golden_dir, "data_6.csv"))[None, :]
    nlp, c, v, f, g = _gpu_eval(nb)
    assert f[0] == 1.1608112892558562e02
    neq = nlp.cinds(0)[5][1]
    assert np.max(np.abs(c[:neq])) == 1.4928675395736724e-06
    _compare(nb, nlp, c, v, f, g)


def test_dense_jacobian_write_set_matches_reference():
    """MOI dense mode: exactly the jac_c! write-set is assigned (quirk Q5), values match."""
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, moi, problem_gen as PG
    from tests.helpers import oracle_model

    batch = PG.make_batch(1, 40, 14, 1, seed=2)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    m_nlp, n_nlp = nlp.num_duals(), nlp.num_primals()
    vec = np.full(m_nlp * n_nlp, np.nan)
    moi.eval_constraint_jacobian(nlp, vec, batch.Z[0])
    D = vec.reshape((m_nlp, n_nlp), order="F")
    onlp = O.OracleNLP(batch.N, int(batch.k_trans[0]), int(batch.init_mode[0]), batch.x0[0], batch.xf[0], batch.obj,
                       oracle_model(batch.model))
    Dref = onlp.jac_c_dense(batch.Z[0])
    assert np.array_equal(np.isnan(D), np.isnan(Dref))
    assert np.count_nonzero(~np.isnan(D)) == 435 + 525 * (batch.N - 1) + 4 * batch.N - int(batch.k_trans[0]) + 3
    assert rel_err(D, Dref, floor=1e-300) <= RTOL
    # MOI-mode callbacks
    g = np.zeros(m_nlp)
    moi.eval_constraint(nlp, g, batch.Z[0])
    assert rel_err(g, onlp.eval_c(batch.Z[0]), floor=1.0) <= RTOL
    assert abs(moi.eval_objective(nlp, batch.Z[0]) - onlp.eval_f(batch.Z[0])) <= RTOL * abs(onlp.eval_f(batch.Z[0]))
    gr = np.zeros(n_nlp)
    moi.eval_objective_gradient(nlp, gr, batch.Z[0])
    assert rel_err(gr, onlp.grad_f(batch.Z[0]), floor=1e-300) <= RTOL


def test_jacobian_structure_matches_oracle():
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, problem_gen as PG
    from tests.helpers import oracle_model

    batch = PG.make_batch(6, 23, seed=9, ragged=True)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    for b in range(batch.B):
        onlp = O.OracleNLP(batch.N, int(batch.k_trans[b]), int(batch.init_mode[b]), batch.x0[b], batch.xf[b],
                           batch.obj[b], oracle_model(batch.model))
        r, c = nlp.jacobian_structure(b)
        ro, co = onlp.jac_structure()
        assert np.array_equal(r, ro) and np.array_equal(c, co)
        assert nlp.cinds(b) == onlp.cinds()
        lb, ub = nlp.bounds(b)
        lo, uo = onlp.bounds()
        assert np.array_equal(lb, lo) and np.array_equal(ub, uo)


def test_constants_can_be_written_once():
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(17, 40, seed=4, ragged=True)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    full = nlp.jac_c(Z, write_constants=True)
    v = torch.full((nlp.dims.j_total,), float("nan"), dtype=torch.float64, device="cuda")
    nlp.init_jacobian_constants(v)
    nlp.jac_c(Z, v, write_constants=False)
    torch.cuda.synchronize()
    a, b = full.cpu().numpy(), v.cpu().numpy()
    for p in range(batch.B):
        assert np.array_equal(nlp.split_vals(a, p), nlp.split_vals(b, p))


def test_golden_vectors_on_gpu(golden_dir):
    """tests/golden/notebook_N61.npz (oracle output committed after it passed KA1/KA2) through the HIP path."""
    import os
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    g = np.load(os.path.join(golden_dir, "notebook_N61.npz"))
    nb = PG.notebook_problem()
    assert np.array_equal(nb.obj, g["cost"]) and np.array_equal(nb.x0[0], g["x0"])
    for tag, Z in (("sol", np.loadtxt(os.path.join(golden_dir, "data_6.csv"))), ("guess", g["Z0"])):
        nb.Z = Z[None, :]
        nlp, c, v, f, gr = _gpu_eval(nb)
        mm, nz = nlp.problem_dims(0)
        assert np.array_equal(c[:mm][: nlp.cinds(0)[5][1]], g[f"c_{tag}"][: nlp.cinds(0)[5][1]])  # equalities: bit-exact
        assert rel_err(c[:mm], g[f"c_{tag}"], floor=1.0) <= RTOL
        assert f[0] == float(g[f"f_{tag}"]) and np.array_equal(gr[: nlp.n_nlp], g[f"grad_{tag}"])
        assert np.array_equal(v[:nz] == 0, g[f"jac_{tag}"] == 0)
        assert rel_err(v[:nz], g[f"jac_{tag}"], floor=1e-300) <= RTOL
        r, cidx = nlp.jacobian_structure(0)
        assert np.array_equal(r, g["rows"]) and np.array_equal(cidx, g["cols"])


def test_long_horizon_and_odd_batch_sizes():
    from quadruped_landing_amd import problem_gen as PG

    for B, N in ((3, 1000), (13, 65), (5, 129), (1, 66), (15, 64)):  # N-1 = 64, 128: a full last chunk
        batch = PG.make_batch(B, N, seed=B * N, ragged=True)
        out = _gpu_eval(batch)
        _compare(batch, *out)


def test_nan_and_inf_propagate_without_touching_other_problems():
    """The reference has no guards: Inf/NaN in Z simply propagate.  A poisoned problem must not
    disturb its neighbours."""
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(6, 20, 8, 1, seed=8)
    clean = _gpu_eval(batch)
    batch.Z[2, 45] = np.nan    # a state entry of knot 3
    batch.Z[4, 119] = np.inf   # h of knot 6
    nlp, c, v, f, g = _gpu_eval(batch)
    for b in (0, 1, 3, 5):
        assert np.array_equal(nlp.split_c(c, b), clean[0].split_c(clean[1], b))
        assert np.array_equal(nlp.split_vals(v, b), clean[0].split_vals(clean[2], b))
        assert f[b] == clean[3][b]
    assert np.isnan(nlp.split_c(c, 2)).any() and np.isnan(f[2])
    assert not np.isfinite(nlp.split_c(c, 4)).all()


def test_error_codes_and_messages():
    import ctypes as C
    import torch
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    batch = PG.make_batch(4, 10, 4, 1)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    L = _lib.lib()
    Z = nlp.upload_Z(batch.Z)
    vals = nlp.new_vals()
    # null and misaligned pointers are refused, not dereferenced
    assert L.qln_eval_constraint(nlp._h, None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    assert L.qln_eval_constraint_jacobian(nlp._h, C.c_void_p(Z.data_ptr()), C.c_void_p(vals.data_ptr() + 8), 0) == \
        _lib.QLN_ERR_INVALID_ARGUMENT
    assert b"16-byte" in L.qln_last_error()
    assert L.qln_problem_dims(nlp._h, 99, None, None) == _lib.QLN_ERR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        nlp.eval_c(Z[:10])
    with pytest.raises(TypeError):
        nlp.eval_c(Z.cpu())
    bad = PG.make_batch(2, 10, 4, 1)
    bad.k_trans[1] = 12  # > N + 1
    with pytest.raises(_lib.QlnError):
        HybridNLP(bad.model, bad.obj, bad.init_mode, bad.k_trans, bad.N, bad.x0, bad.xf)


def test_launches_follow_the_callers_stream():
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(512, 40, 14, 1, seed=6)
    s = torch.cuda.Stream()
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, stream=s)
    Z = nlp.upload_Z(batch.Z)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        Z2 = Z * 1.0                      # produced on s ...
        c, v = nlp.eval_c_and_jac(Z2)     # ... consumed by the kernel on s, no host sync in between
        norm = c.abs().max()              # and its output consumed on s again
    s.synchronize()
    ref = oracle_batch(batch, nlp)
    ok = ~np.isnan(ref["c"])
    assert float(norm) == np.abs(ref["c"][ok]).max() or abs(float(norm) - np.abs(ref["c"][ok]).max()) <= 1e-12


def test_initial_guess_on_device_is_bitwise_the_notebook_rule():
    """Z0 = packZ(nlp, Xguess, Uref) built on the GPU equals the host generator (which equals
    the oracle's restatement of src/main.ipynb:181-198) bit for bit."""
    import torch
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    nb = PG.notebook_problem()
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    Z0 = nlp.initial_guess().cpu().numpy()
    _, xinit, xterm, _, Uref = O.notebook_problem()
    assert np.array_equal(Z0, O.notebook_initial_guess(61, 21, xinit, xterm, Uref))

    batch = PG.make_batch(300, 80, seed=12, ragged=True, noise=0.0)
    batch.x0[:, 14] = np.random.default_rng(0).uniform(0, 1, size=batch.B)  # non-zero initial clocks
    from quadruped_landing_amd.ref_traj import reference_trajectory
    _, Uref = reference_trajectory(batch.model, batch.N, batch.k_trans, batch.xf, batch.init_mode, 0.009)
    want = PG.initial_guess(batch.N, batch.k_trans, batch.x0, batch.xf, Uref)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, z_stride=1600)
    got = nlp.initial_guess().cpu().numpy().reshape(batch.B, 1600)[:, : nlp.n_nlp]
    assert np.array_equal(got, want)
    bad = PG.make_batch(2, 10, 4, 1)
    bad.k_trans[0] = 1
    n2 = HybridNLP(bad.model, bad.obj, bad.init_mode, bad.k_trans, bad.N, bad.x0, bad.xf)
    with pytest.raises(_lib.QlnError) as ei:
        n2.initial_guess()
    assert ei.value.code == _lib.QLN_ERR_UNSUPPORTED


def test_lqr_cost_built_on_device_is_bitwise_the_host_builder():
    """LQRCost records (src/quadratic_cost.jl:33-42) over reference_trajectory (src/ref_traj.jl)
    built by qln_set_lqr_cost equal the host builder bit for bit, shared and per-problem; f and grad follow."""
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    for ragged in (False, True):
        batch = PG.make_batch(96, 33, 11, 2, seed=21, ragged=ragged)
        nlp = HybridNLP(batch.model, None, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
        Z = nlp.upload_Z(batch.Z)
        with pytest.raises(_lib.QlnError):
            nlp.eval_f(Z)  # no cost table yet
        Qw = PG.Q_DIAG.copy()
        Qw[14] = 0.7  # weight on the clock slot too, so the time ramp of Xref matters
        nlp.set_lqr_cost(Qw, PG.R_DIAG, PG.Q_DIAG * 3.0, 0.009, per_problem=ragged)
        from quadruped_landing_amd.ref_traj import reference_trajectory
        from quadruped_landing_amd.quadratic_cost import lqr_objective
        Xref, Uref = reference_trajectory(batch.model, batch.N, batch.k_trans, batch.xf, batch.init_mode, 0.009)
        want = lqr_objective(Qw, PG.R_DIAG, PG.Q_DIAG * 3.0, Xref, Uref) if ragged else \
            lqr_objective(Qw, PG.R_DIAG, PG.Q_DIAG * 3.0, Xref[0], Uref[0])
        got = nlp.get_cost()
        assert got.shape == want.shape and np.array_equal(got, want)
        ref = HybridNLP(batch.model, want, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
        assert np.array_equal(nlp.eval_f(Z).cpu().numpy(), ref.eval_f(Z).cpu().numpy())
        assert np.array_equal(nlp.grad_f(Z).cpu().numpy(), ref.grad_f(Z).cpu().numpy())


def test_constraint_violation_reproduces_ipopts_figure(golden_dir):
    """KA2 end to end on the GPU: eval_c + the violation reduction give the 'Constraint violation' Ipopt printed for
    the shipped run (src/main.ipynb:712), digit for digit; ragged batches agree with a numpy reduction of the oracle's c."""
    import os
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    nb = PG.notebook_problem()
    nb.Z = np.loadtxt(os.path.join(golden_dir, "data_6.csv"))[None, :]
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    v = nlp.constraint_violation(nlp.eval_c(nlp.upload_Z(nb.Z)))
    assert float(v[0]) == 1.4928675395736724e-06

    batch = PG.make_batch(200, 30, seed=31, ragged=True)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    c = nlp.eval_c(nlp.upload_Z(batch.Z))
    got = nlp.constraint_violation(c).cpu().numpy()
    ref = oracle_batch(batch, nlp, want_j=False)["c"]
    for b in range(batch.B):
        cb = nlp.split_c(ref, b)
        want = max(np.abs(cb[: cb.size - batch.N]).max(), np.maximum(-cb[cb.size - batch.N :], 0).max())
        assert got[b] == want or abs(got[b] - want) <= 4e-16 * want  # clearance rows: device sin vs libm
    c[nlp.c_off[5] + 40] = float("nan")
    assert np.isnan(nlp.constraint_violation(c).cpu().numpy()[5])


def test_random_shapes_and_layouts_property():
    """Randomised shapes/layouts (hypothesis): any B, N, per-problem k_trans / init_mode, Z stride and offset alignment
    gives oracle parity through the C ABI."""
    from hypothesis import given, settings, strategies as st
    from quadruped_landing_amd import problem_gen as PG

    @settings(max_examples=_fuzz_examples(20), deadline=None)
    @given(B=st.integers(1, 40), N=st.integers(2, 150), pad=st.integers(0, 9), align=st.sampled_from([1, 2, 3, 16, 32]),
           seed=st.integers(0, 10**6))
    def check(B, N, pad, align, seed):
        batch = PG.make_batch(B, N, seed=seed, ragged=True) if N > 3 else PG.make_batch(B, N, 2, 1 + seed % 2, seed=seed)
        out = _gpu_eval(batch, z_stride=(20 * N - 5 + pad) if pad else 0, align=align)
        _compare(batch, *out)

    check()


def test_every_entry_point_gives_the_same_bits_property():
    """Randomised (hypothesis): for any B (incl. more problems than one round of the chip's XCD map, B % 8 != 0), N (one to
    three chunks), ragged descriptors, format, Z stride and alignment, every way of asking for a quantity -- the fused launch,
    the constraint-only and Jacobian-only launches, qln_eval_all, qln_eval_objective_and_constraint, the separate objective /
    gradient kernels -- returns the same bits, NaN padding untouched; the fused launch itself is held to the oracle."""
    import torch
    from hypothesis import given, settings, strategies as st
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    ran = [0, 0]

    @settings(max_examples=_fuzz_examples(12), deadline=None, derandomize=_fuzz_fixed())
    @given(B=st.one_of(st.integers(1, 60), st.integers(250, 700)), N=st.integers(2, 140), pad=st.integers(0, 5),
           align=st.sampled_from([1, 2, 16]), fmt=st.sampled_from(["dense_blocks", "structural"]), shared=st.booleans(),
           seed=st.integers(0, 10**6))
    def check(B, N, pad, align, fmt, shared, seed):
        ran[0] += 1
        ran[1] += B * N
        if N > 3:
            batch = PG.make_batch(B, N, seed=seed, ragged=not shared, **({"k_trans": min(14, N), "init_mode": 1 + seed % 2} if shared else {}))
        else:
            batch = PG.make_batch(B, N, 2, 1 + seed % 2, seed=seed)
        nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                        z_stride=(20 * N - 5 + pad) if pad else 0, align=align, jac_format=fmt)
        Z = nlp.upload_Z(batch.Z)
        nan = float("nan")
        mk = lambda n: torch.full((n,), nan, dtype=torch.float64, device="cuda")
        zt, ct, jt = nlp.dims.z_total, nlp.dims.c_total, nlp.dims.j_total
        c0, v0 = nlp.eval_c_and_jac(Z, mk(ct), mk(jt))
        c1 = nlp.eval_c(Z, mk(ct))
        v1 = nlp.jac_c(Z, mk(jt))
        f2, g2, c2, v2 = nlp.eval_all(Z, mk(B), mk(zt), mk(ct), mk(jt))
        f3, c3 = nlp.eval_f_and_c(Z, mk(B), mk(ct))
        f4, g4 = nlp.eval_f(Z), nlp.grad_f(Z, mk(zt))
        torch.cuda.synchronize()
        eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
        assert eq(c0, c1) and eq(c0, c2) and eq(c0, c3), "constraint vector differs between entry points"
        assert eq(v0, v1) and eq(v0, v2), "Jacobian values differ between entry points"
        assert eq(f2, f3) and eq(f2, f4) and eq(g2, g4), "objective / gradient differ between entry points"
        if fmt == "dense_blocks" and B <= 60:
            _compare(batch, nlp, c0.cpu().numpy(), v0.cpu().numpy(), f4.cpu().numpy(), g4.cpu().numpy())
        else:
            ref = oracle_batch(batch, nlp, want_c=True, want_j=False, want_f=True, want_grad=False)
            assert np.array_equal(f4.cpu().numpy(), ref["f"])
            assert np.array_equal(np.isnan(c0.cpu().numpy()), np.isnan(ref["c"])) and rel_err(c0.cpu().numpy(), ref["c"], floor=1.0) <= RTOL

    check()
    print(f"{ran[0]} examples, {ran[1]} knot points")


def test_extreme_magnitudes_value_path_property():
    """Randomised (hypothesis): decision vectors whose entries span the whole FP64 range -- denormals, 1e+-300, exact zeros of
    either sign, negative step lengths, intermediate overflow to Inf and Inf - Inf = NaN.  The value path follows the
    reference's operation order with IEEE division and no contraction, so the equality rows of c, the objective and the
    gradient must equal the oracle's BIT FOR BIT, NaN for NaN and Inf for Inf; the clearance rows differ by at most 1 ulp
    (device sin vs libm, huge arguments included).  (Jacobian entries are not compared here: closed form and dual numbers
    overflow and cancel at different places; the launch must merely survive.)"""
    from hypothesis import given, settings, strategies as st
    from quadruped_landing_amd import problem_gen as PG

    ran = [0]

    @settings(max_examples=_fuzz_examples(10), deadline=None, derandomize=_fuzz_fixed())
    @given(B=st.integers(1, 24), N=st.integers(2, 90), fmt=st.sampled_from(["dense_blocks", "structural"]),
           frac=st.sampled_from([0.002, 0.02, 0.3, 1.0]), lo=st.sampled_from([-320, -300, -30]), hi=st.sampled_from([30, 150, 300]),
           seed=st.integers(0, 10**6))
    def check(B, N, fmt, frac, lo, hi, seed):
        ran[0] += 1
        batch = PG.make_batch(B, N, seed=seed, ragged=True) if N > 3 else PG.make_batch(B, N, 2, 1 + seed % 2, seed=seed)
        rng = np.random.default_rng(seed + 1)
        Z = batch.Z
        hit = rng.random(Z.shape) < frac
        mag = 10.0 ** rng.uniform(lo, hi, size=Z.shape)          # 1e-320 is denormal
        mag[rng.random(Z.shape) < 0.1] = 0.0
        Z[hit] = (np.where(rng.random(Z.shape) < 0.5, -1.0, 1.0) * mag)[hit]
        with np.errstate(all="ignore"):
            nlp, c, v, f, g = _gpu_eval(batch, jac_format=fmt)
            ref = oracle_batch(batch, nlp, want_c=True, want_j=False, want_f=True, want_grad=True)
        same = lambda a, b: np.array_equal(a, b, equal_nan=True)
        for b in range(B):
            neq = nlp.cinds(b)[5][1]  # the equality rows come first; the clearance rows (device sin) are the last group
            a, r = nlp.split_c(c, b), nlp.split_c(ref["c"], b)
            assert same(a[:neq], r[:neq]), f"equality rows differ (problem {b})"
            ai, ri = a[neq:], r[neq:]
            assert np.array_equal(np.isnan(ai), np.isnan(ri))
            ok = np.isfinite(ri)
            assert same(ai[~ok], ri[~ok]) and np.all(np.abs(ai[ok] - ri[ok]) <= 2.3e-16 * np.maximum(np.abs(ri[ok]), 0.25))
        assert same(f, ref["f"]), "objective differs"
        assert same(g.reshape(B, -1)[:, : nlp.n_nlp], ref["grad"].reshape(B, -1)[:, : nlp.n_nlp]), "gradient differs"

    check()
    print(f"{ran[0]} examples")


def test_dense_host_jacobian_of_any_problem_in_a_batch_and_two_handles():
    """MOI dense mode addresses one problem of a batch; two handles on the same device do not interfere."""
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, moi, problem_gen as PG
    from tests.helpers import oracle_model

    batch = PG.make_batch(5, 18, seed=41, ragged=True)
    other = PG.make_batch(7, 25, seed=42, ragged=True)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    nlp2 = HybridNLP(other.model, other.obj, other.init_mode, other.k_trans, other.N, other.x0, other.xf)
    for b in (0, 2, 4):
        m_nlp, n_nlp = nlp.num_duals(b), nlp.num_primals()
        vec = np.full(m_nlp * n_nlp, np.nan)
        _ = nlp2.eval_c_host(other.Z)  # interleave work on the other handle
        moi.eval_constraint_jacobian(nlp, vec, batch.Z[b], b)
        D = vec.reshape((m_nlp, n_nlp), order="F")
        o = O.OracleNLP(batch.N, int(batch.k_trans[b]), int(batch.init_mode[b]), batch.x0[b], batch.xf[b], batch.obj[b],
                        oracle_model(batch.model))
        Dref = o.jac_c_dense(batch.Z[b])
        assert np.array_equal(np.isnan(D), np.isnan(Dref))
        assert rel_err(D, Dref, floor=1e-300) <= RTOL
    c2 = nlp2.eval_c_host(other.Z)
    ref2 = oracle_batch(other, nlp2, want_j=False)["c"]
    ok = ~np.isnan(ref2)
    assert rel_err(c2[ok], ref2[ok], floor=1.0) <= RTOL


def test_region_placed_buffer_gives_the_same_values_and_is_released():
    """qln_vals_alloc_placed: a buffer built with the HIP virtual-memory API behaves like any other device memory."""
    import ctypes as C
    import torch
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    batch = PG.make_batch(300, 40, 14, 1, seed=5)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    c = nlp.new_c()
    free0, _ = torch.cuda.mem_get_info()
    vals, ms = nlp.new_vals_regions(Z, c)
    assert vals.numel() == nlp.dims.j_total and vals.data_ptr() % (2 << 20) == 0 and ms > 0
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 * nlp.dims.j_total + (600 << 20)  # everything outside the window went back to the driver
    vals.zero_()  # (the padding between problems is never written)
    nlp.eval_c_and_jac(Z, c, vals, write_constants=True)
    c2, v2 = nlp.eval_c_and_jac(Z, write_constants=True)
    torch.cuda.synchronize()
    assert torch.equal(vals, v2) and torch.equal(c, c2)
    ptr = vals.data_ptr()
    del vals
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 > free1  # released when the tensor went away
    # not a buffer of this handle (any more)
    assert _lib.lib().qln_vals_free_placed(nlp._h, ptr) == _lib.QLN_ERR_INVALID_ARGUMENT
    # the address space those calls retired is accounted for, and a smaller transient budget works the same way
    retired, cap = nlp.placed_address_space()
    assert retired >= 8 * nlp.dims.j_total and cap == 64 << 40
    v32, ms32 = nlp.new_vals_regions(Z, c, transient_gib=32.0)
    assert nlp.placed_address_space()[0] > retired and ms32 > 0
    v32.zero_()
    nlp.eval_c_and_jac(Z, c, v32, write_constants=True)
    torch.cuda.synchronize()
    assert torch.equal(v32, v2)
    del v32
    assert _lib.lib().qln_vals_alloc_placed_budget(nlp._h, Z.data_ptr(), None, -1, C.byref(C.c_void_p()), None) == _lib.QLN_ERR_INVALID_ARGUMENT
    # allocate / use / free repeatedly (a freed range's addresses must never serve a later buffer through stale
    # translations: the library keeps its virtual ranges reserved), the last one is left to qln_destroy
    for rep in range(3):
        vals, _ = nlp.new_vals_regions(Z, c)
        vals.zero_()
        nlp.eval_c_and_jac(Z, c, vals, write_constants=True)
        torch.cuda.synchronize()
        assert torch.equal(vals, v2), rep
        if rep < 2:
            del vals


@pytest.mark.parametrize("B", [5, 160])
def test_host_pointer_mode_both_paths(B):
    """MOI-mode entry points on host buffers: small batches work on mapped host memory directly (zero copy), batches
    above 8 MB per callback are staged through device memory -- both give what the device-pointer entry points give."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, 40, 14, 1, seed=B)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    total_bytes = 8 * (nlp.dims.z_total + nlp.dims.c_total + nlp.dims.j_total)
    assert (total_bytes <= 8 << 20) == (B == 5)  # the two sizes sit on either side of the switch
    Zd = nlp.upload_Z(batch.Z)
    c, v = nlp.eval_c_and_jac(Zd, write_constants=True)
    f, g = nlp.eval_f(Zd), nlp.grad_f(Zd)
    torch.cuda.synchronize()
    for rep in range(2):  # the second round reuses the handle's buffers
        assert np.array_equal(nlp.eval_c_host(batch.Z), c.cpu().numpy())
        assert np.array_equal(nlp.eval_f_host(batch.Z), f.cpu().numpy())
        assert np.array_equal(nlp.grad_f_host(batch.Z).reshape(B, -1)[:, : nlp.n_nlp],
                              g.cpu().numpy().reshape(B, -1)[:, : nlp.n_nlp])
        vh, vd = nlp.jac_c_host(batch.Z), v.cpu().numpy()
        for b in range(B):
            assert np.array_equal(nlp.split_vals(vh, b), nlp.split_vals(vd, b))
        m = nlp.problem_dims(B - 1)[0]
        D = np.full((m, nlp.n_nlp), np.nan, order="F")
        nlp.jac_c_dense_host(batch.Z[B - 1], D, B - 1)
        r, cidx = nlp.jacobian_structure(B - 1)
        assert np.array_equal(D[r, cidx], nlp.split_vals(vd, B - 1))


@pytest.mark.parametrize("B,N,ragged,fmt", [(64, 40, False, "dense_blocks"), (33, 61, False, "structural"), (29, 80, True, "dense_blocks"),
                                            (17, 130, True, "structural"), (5, 64, False, "dense_blocks"), (3, 65, False, "dense_blocks"),
                                            (4, 2, False, "dense_blocks")])
def test_eval_all_gives_the_bits_of_the_separate_entry_points(B, N, ragged, fmt):
    """qln_eval_all: f, grad_f, c and the Jacobian values out of one read of Z in one launch are bit for bit what
    qln_eval_objective / _gradient / qln_eval_constraint_and_jacobian give (which are held to the oracle above) --
    incl. horizons of exactly one chunk (N - 1 = 64: the terminal knot has no lane left), more than one chunk, N = 2."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, N, min(14, N), 1, seed=B + N, ragged=ragged)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, jac_format=fmt)
    Z = nlp.upload_Z(batch.Z)
    nan = float("nan")
    mk = lambda n: torch.full((n,), nan, dtype=torch.float64, device="cuda")
    f, g, c, v = nlp.eval_all(Z, mk(B), mk(nlp.dims.z_total), mk(nlp.dims.c_total), mk(nlp.dims.j_total))
    c1, v1 = nlp.eval_c_and_jac(Z, mk(nlp.dims.c_total), mk(nlp.dims.j_total))
    f1, g1 = nlp.eval_f(Z), nlp.grad_f(Z, mk(nlp.dims.z_total))
    torch.cuda.synchronize()
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert eq(c, c1) and eq(v, v1) and eq(f, f1) and eq(g, g1)
    assert not torch.isnan(f).any()
    ref = oracle_batch(batch, nlp, want_c=False, want_j=False, want_f=True, want_grad=True)
    assert np.array_equal(f.cpu().numpy(), ref["f"])  # and the objective is bit-identical to the oracle


@pytest.mark.parametrize("N,kt,im,z_pad", [(64, 20, 1, 0), (63, 33, 2, 0), (2, 2, 1, 0), (17, 6, 1, 0),
                                            # one horizon per number of 16-byte load instructions a slice of Z takes (1 .. 10)
                                            (6, 3, 1, 0), (7, 3, 2, 0), (14, 5, 1, 0), (20, 7, 1, 0), (26, 9, 2, 0), (33, 12, 1, 0),
                                            (39, 14, 1, 0), (45, 20, 1, 0), (46, 20, 2, 0), (52, 30, 1, 0), (58, 10, 1, 0),
                                            # slices 16-byte aligned (even stride) and with a gap between problems
                                            (40, 14, 1, 1), (40, 14, 2, 6)])
def test_shared_cost_table_kernels_at_their_size_limits(N, kt, im, z_pad):
    """A batch of >= 4 096 problems that shares one cost table runs eval_f / grad_f! through the shared-table kernels
    (k_objective_shared: persistent waves, lane = knot's cost record in registers, a problem's slice of Z as 16-byte pieces
    that are only 8-byte aligned when the stride is odd; k_objective_gradient_shared: table in LDS) for N <= 64: objective
    and gradient of EVERY problem bit-identical to the oracle at every instantiation of those kernels, for a batch size
    that leaves the persistent waves' last round partly empty."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    B = 4096 + 3
    batch = PG.make_batch(B, N, kt, im, seed=N)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                    z_stride=(20 * N - 5 + z_pad) if z_pad else 0)
    Z = nlp.upload_Z(batch.Z)
    f, g = nlp.eval_f(Z), nlp.grad_f(Z)
    torch.cuda.synchronize()
    ref = oracle_batch(batch, nlp, want_c=False, want_j=False, want_f=True, want_grad=True, nthreads=8)
    assert np.array_equal(f.cpu().numpy(), ref["f"])
    assert np.array_equal(g.cpu().numpy().reshape(B, -1)[:, : nlp.n_nlp], ref["grad"].reshape(B, -1)[:, : nlp.n_nlp])


@pytest.mark.parametrize("B,N,ragged", [(64, 40, False), (5000, 40, False), (33, 61, False), (29, 80, True), (5, 64, False), (3, 65, False),
                                        (4, 2, False), (7, 130, True), (9, 41, False), (6, 66, True)])
def test_objective_and_constraint_in_one_launch_give_the_bits_of_the_separate_entry_points(B, N, ragged):
    """qln_eval_objective_and_constraint: f and c out of one read of Z in one launch -- what a line search asks for -- are bit for
    bit qln_eval_objective's and qln_eval_constraint's (which are held to the oracle above), at every chunking the launch
    chooses (40-knot chunks up to N = 41 and from N = 66, 64-knot chunks between), shared and per-problem cost tables."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, N, min(14, N), 1, seed=B + N, ragged=ragged)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    nan = float("nan")
    mk = lambda n: torch.full((n,), nan, dtype=torch.float64, device="cuda")
    f, c = nlp.eval_f_and_c(Z, mk(B), mk(nlp.dims.c_total))
    f1, c1 = nlp.eval_f(Z), nlp.eval_c(Z, mk(nlp.dims.c_total))
    torch.cuda.synchronize()
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert eq(f, f1) and eq(c, c1) and not torch.isnan(f).any()
    ref = oracle_batch(batch, nlp, want_c=False, want_j=False, want_f=True)
    assert np.array_equal(f.cpu().numpy(), ref["f"])
    with pytest.raises(TypeError):
        nlp.eval_f_and_c(Z.cpu())
