"""The BASELINE.json configurations that no other GPU test runs at their exact sizes, through the HIP path:

  configs[1]  B=1024, N=40, k_trans=14: EVERY problem against the oracle -- both Jacobian formats, the host-pointer
              (MOI) mode, the dense reference-compatible scatter;
  configs[4]  B=524288, N=40 sharded over 8 GPUs: the eight shard_range shards run one after the other on the one
              visible GPU (shard -> handle -> evaluate -> the per-shard results placed at their GLOBAL index, as the
              end-of-job gather does), first / last / sampled problems of every shard against the oracle;

plus two checks that do not pass through the oracle's arithmetic:
  KA6         the notebook's iteration-0 objective 1.8380701e+00 (src/main.ipynb:232) through the HIP path;
  grad_f      the GPU gradient against central differences of the GPU objective, modulo the documented quirk Q2.
"""
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import build_c_host, oracle_batch, oracle_model, rel_err, write_problem_file

pytestmark = pytest.mark.gpu
RTOL = 1e-8  # north_star: "within 1e-8 relative on FP64 dynamics/Jacobian entries"


# ------------------------------------------------------------------------------------------------ configs[1]


def test_config1_B1024_every_problem_dense_blocks():
    """BASELINE.json configs[1] exactly: B=1024, N=40, k_trans=14, init_mode=1, FP64, one GPU; all 1024 problems."""
    from quadruped_landing_amd import problem_gen as PG
    from tests.test_gpu_parity import _compare, _gpu_eval

    batch = PG.make_batch(1024, 40, 14, 1, seed=0)  # bench.py's batch for this workload (seed = rank = 0)
    nlp, c, v, f, g = _gpu_eval(batch)
    ec, ev, ef, eg = _compare(batch, nlp, c, v, f, g)
    # the value path is bit-identical to the oracle on this configuration too (clearance rows aside: device sin)
    ref = oracle_batch(batch, nlp, want_f=True, want_grad=True)
    neq = nlp.cinds(0)[5][1]
    for b in range(batch.B):
        assert np.array_equal(c[nlp.c_off[b] : nlp.c_off[b] + neq], ref["c"][nlp.c_off[b] : nlp.c_off[b] + neq]), b
    assert np.array_equal(f, ref["f"]) and np.array_equal(g, ref["grad"])


def test_config1_B1024_every_problem_structural_format():
    from quadruped_landing_amd import problem_gen as PG
    from tests.test_gpu_structural import _check

    _check(PG.make_batch(1024, 40, 14, 1, seed=0))


def test_config1_B1024_host_pointer_mode():
    """The MOI-mode entry points at configs[1]'s size (staged through device memory: 98 MB per callback) give what the
    device-pointer entry points give, bit for bit; the dense scatter of a few problems matches the oracle's dense
    jac_c! including its write-set."""
    import torch
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, moi, problem_gen as PG

    batch = PG.make_batch(1024, 40, 14, 1, seed=0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Zd = nlp.upload_Z(batch.Z)
    c, v = nlp.eval_c_and_jac(Zd, write_constants=True)
    f, g = nlp.eval_f(Zd), nlp.grad_f(Zd)
    torch.cuda.synchronize()
    assert np.array_equal(nlp.eval_c_host(batch.Z), c.cpu().numpy())
    assert np.array_equal(nlp.eval_f_host(batch.Z), f.cpu().numpy())
    assert np.array_equal(nlp.grad_f_host(batch.Z), g.cpu().numpy())
    assert np.array_equal(nlp.jac_c_host(batch.Z), v.cpu().numpy())
    for b in (0, 511, 1023):
        m, n_nlp = nlp.num_duals(b), nlp.num_primals()
        vec = np.full(m * n_nlp, np.nan)
        moi.eval_constraint_jacobian(nlp, vec, batch.Z, b)  # the whole batch as x: problem b's slice is taken
        D = vec.reshape((m, n_nlp), order="F")
        o = O.OracleNLP(batch.N, 14, 1, batch.x0[b], batch.xf[b], batch.obj, oracle_model(batch.model))
        Dref = o.jac_c_dense(batch.Z[b])
        assert np.array_equal(np.isnan(D), np.isnan(Dref))  # exactly the reference's write-set
        assert int((~np.isnan(D)).sum()) == 435 + 525 * 39 + 4 * 40 - 14 + 3
        assert rel_err(D, Dref, floor=1e-300) <= RTOL


# ------------------------------------------------------------------------------------------------ configs[4]


def test_config4_B524288_eight_shards_on_one_gpu():
    """BASELINE.json configs[4]: B = 524288 over 8 GPUs = shard_range(524288, r, 8), r = 0..7.  Each shard gets its
    own handle (as its rank would build it: the shard's problems generated with seed = rank, like bench.py), is
    evaluated by the fused launch, and its per-problem results land at their global index in the gathered arrays.
    Checked: the shard ranges tile the global batch; every output slot of every shard is written and the padding is
    not; first / last / sampled problems of every shard equal the oracle at the same global index; the gathered
    objective / violation vectors carry each shard's values at its global range."""
    import torch
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, distributed as D, problem_gen as PG

    total, world, N, kt = 524288, 8, 40, 14
    f_global = torch.full((total,), float("nan"), dtype=torch.float64, device="cuda")
    viol_global = torch.full((total,), float("nan"), dtype=torch.float64, device="cuda")
    rng = np.random.default_rng(4)
    expect, end = {}, 0
    worst = 0.0
    for r in range(world):
        lo, hi = D.shard_range(total, r, world)
        assert lo == end and hi - lo == 65536
        end = hi
        batch = PG.make_batch(hi - lo, N, kt, 1, seed=r)
        nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                        stream=torch.cuda.current_stream())
        Z = nlp.upload_Z(batch.Z)
        c = torch.full((nlp.dims.c_total,), float("nan"), dtype=torch.float64, device="cuda")
        v = torch.full((nlp.dims.j_total,), float("nan"), dtype=torch.float64, device="cuda")
        nlp.eval_c_and_jac(Z, c, v, write_constants=True)
        nlp.eval_f(Z, f_global[lo:hi])                      # the gather: shard r's results at [lo, hi)
        nlp.constraint_violation(c, viol_global[lo:hi])
        torch.cuda.synchronize()
        m, nz = 18 * N - kt + 16, 300 * (N - 1) + N + 435 + 15 * (N - 1) + 3 * N - kt + 3
        assert int(torch.isnan(c).sum()) == nlp.dims.c_total - (hi - lo) * m
        assert int(torch.isnan(v).sum()) == nlp.dims.j_total - (hi - lo) * nz
        for g_idx in np.unique(np.concatenate([[lo, hi - 1], rng.integers(lo, hi, size=6)])):
            b = int(g_idx - lo)
            o = O.OracleNLP(N, kt, 1, batch.x0[b], batch.xf[b], batch.obj, oracle_model(batch.model))
            oc, ov, of = o.eval_c(batch.Z[b]), o.jac_c_coo(batch.Z[b]), o.eval_f(batch.Z[b])
            gc = c[nlp.c_off[b] : nlp.c_off[b] + m].cpu().numpy()
            gv = v[nlp.j_off[b] : nlp.j_off[b] + nz].cpu().numpy()
            assert np.array_equal(gv == 0, ov == 0)
            worst = max(worst, rel_err(gc, oc, floor=1.0), rel_err(gv, ov, floor=1e-300))
            neq = m - N
            expect[int(g_idx)] = (of, float(np.max(np.maximum(np.abs(oc[:neq]).max(), np.maximum(-oc[neq:], 0).max()))))
        del nlp, Z, c, v, batch
        torch.cuda.empty_cache()
    assert end == total and worst <= RTOL
    assert not bool(torch.isnan(f_global).any()) and not bool(torch.isnan(viol_global).any())
    fg, vg = f_global.cpu().numpy(), viol_global.cpu().numpy()
    for g_idx, (of, oviol) in expect.items():
        assert fg[g_idx] == of                                   # objective: bit-identical to the oracle
        assert abs(vg[g_idx] - oviol) <= 1e-12 * max(1.0, oviol)  # violation: a max over rows that agree to rounding
    print(f"configs[4]: 8 shards x 65536 problems, {len(expect)} problems checked at their global index, "
          f"worst rel err {worst:.3e}")


# ------------------------------------------------------------------------------------------------ KA6, gradient


def test_KA6_iteration0_objective_through_the_hip_path():
    """src/main.ipynb:232: Ipopt's iteration-0 objective 1.8380701e+00 = eval_f at Z0 pushed inside solve()'s variable
    bounds (quirk Q6 included; nlp.ipopt_initial_point) -- a known answer at a non-solution point, all 8 digits."""
    import torch
    from quadruped_landing_amd import HybridNLP, nlp as NL, problem_gen as PG

    nb = PG.notebook_problem()
    h = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    Z0 = h.initial_guess()  # the notebook's Z0, built on the device
    torch.cuda.synchronize()
    assert np.array_equal(Z0.cpu().numpy(), nb.Z[0])
    assert f"{float(h.eval_f(Z0)[0]):.7e}" == "1.5438468e+00"
    Zp = NL.ipopt_initial_point(nb.Z[0], *NL.variable_bounds(nb.N))
    assert f"{float(h.eval_f(h.upload_Z(Zp[None, :]))[0]):.7e}" == "1.8380701e+00"
    assert f"{h.eval_f_host(Zp)[0]:.7e}" == "1.8380701e+00"  # and through the MOI-mode entry point
    Zf = NL.ipopt_initial_point(nb.Z[0], *NL.variable_bounds_forces(nb.N))
    assert f"{h.eval_f_host(Zf)[0]:.7e}" == "1.8371626e+00"  # the bounds the source comment describes do not give it


@pytest.mark.parametrize("N,kt,ragged", [(40, 14, False), (23, 9, True)])
def test_gpu_gradient_is_the_central_difference_of_the_gpu_objective(N, kt, ragged):
    """grad_f! against eval_f, both from the GPU, no oracle in between: the objective is a polynomial of degree <= 3
    in Z (quadratic stage costs times h_k), so central differences are exact up to rounding.  The reference's
    gradient leaves out d(h_k l_k)/dh_k = l_k (quirk Q2, src/costs.jl:26-31): on the h entries the difference must
    be exactly the stage cost, formed here from the cost records and Z."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    one = PG.make_batch(3, N, kt, 1, seed=21, ragged=ragged)
    b = 1
    n_nlp = 20 * N - 5
    obj = one.obj if one.obj.ndim == 2 else one.obj[b]
    h1 = HybridNLP(one.model, obj, one.init_mode[b], one.k_trans[b], N, one.x0[b], one.xf[b])
    g = h1.grad_f(h1.upload_Z(one.Z[b][None, :])).cpu().numpy()[:n_nlp]
    # one problem per perturbed entry, two batches (+eps, -eps)
    eps = 1e-4
    hB = HybridNLP(one.model, obj, one.init_mode[b], one.k_trans[b], N, np.tile(one.x0[b], (n_nlp, 1)), one.xf[b])
    Zp = np.tile(one.Z[b], (n_nlp, 1))
    Zm = Zp.copy()
    Zp[np.arange(n_nlp), np.arange(n_nlp)] += eps
    Zm[np.arange(n_nlp), np.arange(n_nlp)] -= eps
    fd = (hB.eval_f(hB.upload_Z(Zp)) - hB.eval_f(hB.upload_Z(Zm))).cpu().numpy() / (2 * eps)
    torch.cuda.synchronize()
    # Q2: the stage cost l_k(x_k, u_k) = 0.5 x'Qx + q'x + 0.5 u'Ru + r'u + c (src/quadratic_cost.jl:44-47)
    z = one.Z[b][: 20 * (N - 1)].reshape(N - 1, 20)
    rec = np.asarray(obj)[: N - 1]
    stage = 0.5 * np.sum(rec[:, :20] * z * z, axis=1) + np.sum(rec[:, 20:40] * z, axis=1) + rec[:, 40]
    missing = np.zeros(n_nlp)
    missing[19 : 20 * (N - 1) : 20] = stage
    scale = np.max(np.abs(fd))
    err = np.max(np.abs(fd - (g + missing))) / scale
    print(f"N={N}: max |central difference - (grad_f + Q2 term)| / max|df| = {err:.2e}; "
          f"max Q2 term / max|df| = {np.max(np.abs(missing)) / scale:.2e}")
    assert err <= 1e-8
    assert np.max(np.abs(missing)) > 1e-3 * scale  # the Q2 term is not negligible: the check would see its absence


# ------------------------------------------------------------------------------------------------ C host


def test_c_host_makes_the_julia_bindings_call_sequence(tmp_path, golden_dir):
    """tests/host_c/moi_host.c: HybridNLPHIP.jl's call sequence from a plain C program (by-value qln_batch_desc,
    caller-malloc'd buffers of exactly m_nlp / n_nlp / m_nlp*n_nlp / nnz doubles with guard words, dense matrix
    pre-filled with a sentinel) on the notebook problem at Z = data_6.csv: KA1, KA2 and the size of jac_c!'s
    write-set, 435 + 525(N-1) + 4N - k_trans + 3 (src/constraints.jl:219-274)."""
    from quadruped_landing_amd import problem_gen as PG

    exe = build_c_host(tmp_path)
    nb = PG.notebook_problem()
    Z = np.loadtxt(os.path.join(golden_dir, "data_6.csv"))
    pf = tmp_path / "problem.txt"
    write_problem_file(pf, nb, Z, Z0=nb.Z[0])
    out = subprocess.run([str(exe), str(pf)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0
    kv = dict(line.split("=", 1) for line in out.stdout.splitlines() if "=" in line)
    N, k_trans = 61, 21
    assert (int(kv["n_nlp"]), int(kv["m_nlp"]), int(kv["n_eq"]), int(kv["n_ineq"])) == (1215, 1093, 1032, 61)  # KA3
    assert kv["totals_match"] == "1" and kv["guards_intact"] == "1"
    assert float(kv["f"]) == 1.1608112892558562e02                   # KA1, src/main.ipynb:710
    assert float(kv["max_abs_c_eq"]) == 1.4928675395736724e-06       # KA2, src/main.ipynb:712
    assert float(kv["min_c_ineq"]) >= 0
    assert int(kv["grad_written"]) == 1215
    write_set = 435 + 525 * (N - 1) + 4 * N - k_trans + 3
    assert int(kv["dense_write_set"]) == write_set == 32161
    assert int(kv["nnz"]) == int(kv["sparse_in_range"]) == int(kv["sparse_equals_dense"])
    assert int(kv["nnz"]) == write_set - 210 * (N - 1)  # the COO list leaves out the off-diagonal zeros of the -I blocks
    # ... and solve(Z0, nlp) itself from the C host: the GPU solve reaches the feasibility of the reference's Ipopt run
    assert kv["solve_status"] == "0" and float(kv["solve_violation"]) <= 1.4928675395736724e-06
    assert 90.0 <= float(kv["solve_f"]) <= 125.0
