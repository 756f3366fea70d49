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
