"""qln_solve: the batched solve of the reference NLP on the GPU (the reference's solve(),
src/moi.jl:46-103).  The reference holds no data for solver iterates -- its own run of this NLP ended in "EXIT:
Restoration Failed!" at objective 1.1608112892558562e+02 / constraint violation 1.4928675395736724e-06
(src/main.ipynb:710-727) -- so the result is judged by the EVALUATOR (which is parity-checked against the oracle
elsewhere): constraint violation as Ipopt defines it, objective, variable bounds of solve() incl. quirk Q6, all computed
from the returned Z by qln_eval_constraint / qln_constraint_violation / qln_eval_objective, and cross-checked on the
host by the oracle."""
import numpy as np
import pytest

from tests.helpers import oracle_model

pytestmark = pytest.mark.gpu


def _judge(nlp, Z):
    """violation (Ipopt's definition, from the evaluator), objective, bound violation of solve()'s variable bounds"""
    import torch
    from quadruped_landing_amd import nlp as NL

    c = nlp.eval_c(Z)
    viol = nlp.constraint_violation(c).cpu().numpy()
    f = nlp.eval_f(Z).cpu().numpy()
    torch.cuda.synchronize()
    Zh = Z.cpu().numpy().reshape(nlp.B, -1)[:, : nlp.n_nlp]
    xl, xu = NL.variable_bounds(nlp.N)
    bviol = np.maximum(np.maximum(xl - Zh, Zh - xu), 0.0).max(axis=1)
    return viol, f, bviol, c.cpu().numpy(), Zh


def test_notebook_problem_from_Z0_reaches_the_reference_runs_feasibility():
    """The notebook problem (N = 61, k_trans = 21) from its initial guess Z0: constraint violation <= 1.4928675e-06 (what
    the reference's own Ipopt run ended with, src/main.ipynb:712) and an objective in the neighbourhood of its
    116.08 (:710) -- the value reached is printed; it is a target, not a parity pin."""
    import torch
    from oracle import oracle as O
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    nb = PG.notebook_problem()
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    Z = nlp.initial_guess()
    Z, info = nlp.solve(Z)
    torch.cuda.synchronize()
    inf = info.cpu().numpy()[0]
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    print(f"notebook problem: outer {inf[0]:.0f}, iLQR iterations {inf[1]:.0f}, status {inf[5]:.0f}, rho {inf[4]:.1e}; "
          f"evaluator: f = {f[0]:.6f} (reference run: 116.081129), violation = {viol[0]:.3e} (reference run: 1.493e-06), "
          f"bound violation {bviol[0]:.2e}, sum h = {inf[8]:.4f}")
    assert inf[5] == 0
    assert viol[0] <= 1.4928675395736724e-06
    assert bviol[0] <= 1e-6
    assert 90.0 <= f[0] <= 125.0           # neighbourhood of the reference run's 116.08
    assert abs(inf[2] - f[0]) <= 1e-9 * abs(f[0])  # the solver's own report is the evaluator's objective
    # rows the roll-out satisfies by construction are zero to the last bit: initial condition, dynamics, contact
    ci = nlp.cinds(0)
    for grp in (0, 2, 3, 4):
        assert np.all(c[ci[grp][0] - 1 : ci[grp][1]] == 0.0), grp
    # the same verdict from the CPU oracle on the same Z
    o = O.OracleNLP(nb.N, 21, 1, nb.x0[0], nb.xf[0], nb.obj, oracle_model(nb.model))
    oc = o.eval_c(Zh[0])
    neq = ci[5][1]
    assert max(np.abs(oc[:neq]).max(), np.maximum(-oc[neq:], 0).max()) <= 1.4928675395736724e-06
    assert abs(o.eval_f(Zh[0]) - f[0]) <= 1e-12 * abs(f[0])


@pytest.mark.parametrize("exact", [0, 1])
def test_batch_of_random_landing_problems(exact):
    """256 problems of BASELINE.json configs[1]'s shape (N = 40, k_trans = 14, random drop states): every one solved to
    the tolerance, judged by the evaluator; the exact-gradient mode reaches a lower objective than the reference-gradient
    mode (it may move the step lengths h to lower the cost; quirk Q2 hides that from the reference's gradient)."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(256, 40, 14, 1, seed=3, noise=0.0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.initial_guess()
    # (the exact gradient couples the step lengths to the cost: its inner solves need to be tighter than the default 8)
    Z, info = nlp.solve(Z, exact_h_gradient=exact, **(dict(max_inner=30, max_outer=40) if exact else {}))
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    print(f"exact_h_gradient={exact}: status counts {np.bincount(inf[:, 5].astype(int), minlength=3)}, iLQR iterations "
          f"median {np.median(inf[:, 1]):.0f} max {inf[:, 1].max():.0f}, violation max {viol.max():.2e}, f median {np.median(f):.3f}")
    assert np.all(np.isfinite(Zh))
    solved = (inf[:, 5] == 0)
    assert solved.mean() >= 0.98
    assert viol[solved].max() <= 1e-6 * 1.0001 and bviol[solved].max() <= 1e-6
    test_batch_of_random_landing_problems.f = getattr(test_batch_of_random_landing_problems, "f", {})
    test_batch_of_random_landing_problems.f[exact] = f
    if exact == 1 and 0 in test_batch_of_random_landing_problems.f:
        assert np.median(f) < np.median(test_batch_of_random_landing_problems.f[0])


def test_solve_is_deterministic_and_independent_of_batch_position():
    """The same problem gives the same bits wherever it sits in a batch (one wave per problem, no cross-problem state)."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(9, 25, 9, 1, seed=8, noise=0.0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z, info = nlp.solve(nlp.initial_guess(), max_outer=6)
    one = HybridNLP(batch.model, batch.obj, 1, 9, 25, batch.x0[4], batch.xf[4])
    Z1, info1 = one.solve(one.initial_guess(), max_outer=6)
    torch.cuda.synchronize()
    assert torch.equal(Z.view(9, -1)[4], Z1.view(1, -1)[0])
    assert torch.equal(info[4][:10], info1[0][:10])  # (entries 10-14 are phase timers)


def test_long_horizon_with_per_problem_transition_knots():
    """N = 80 (more knots than lanes: the per-knot phases loop over two chunks) and a different transition knot per
    problem: every one solved, judged by the evaluator."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG, quadratic_cost as QC
    from quadruped_landing_amd.ref_traj import reference_trajectory

    B, N = 96, 80
    base = PG.make_batch(B, N, 20, 1, seed=13, noise=0.0, build_obj=False)
    kt = np.random.default_rng(5).integers(10, 41, size=B).astype(np.int32)
    nlp = HybridNLP(base.model, None, 1, kt, N, base.x0, base.xf)
    nlp.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=True)  # Xref/Uref depend on k_trans
    Z, info = nlp.solve(nlp.initial_guess())
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    print(f"N=80, k_trans in [10, 40]: status counts {np.bincount(inf[:, 5].astype(int), minlength=3)}, iLQR iterations median "
          f"{np.median(inf[:, 1]):.0f} max {inf[:, 1].max():.0f}, violation max {viol.max():.2e}")
    assert (inf[:, 5] == 0).mean() >= 0.98
    ok = inf[:, 5] == 0
    assert viol[ok].max() <= 1e-6 * 1.0001 and bviol[ok].max() <= 1e-6
    for b in np.nonzero(ok)[0][:8]:  # dynamics / contact rows of a rolled-out trajectory are exactly zero
        ci = nlp.cinds(int(b))
        seg = nlp.split_c(c, int(b))
        assert np.all(seg[ci[2][0] - 1 : ci[4][1]] == 0.0)


def test_solve_rejects_what_it_cannot_do():
    """Error behaviour of the boundary: no cost table, bad options, a horizon that does not fit a CU's LDS."""
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    b = PG.make_batch(2, 12, 5, 1, seed=1, build_obj=False)
    nlp = HybridNLP(b.model, None, b.init_mode, b.k_trans, b.N, b.x0, b.xf)
    Z = nlp.upload_Z(b.Z)
    with pytest.raises(_lib.QlnError) as e:
        nlp.solve(Z)
    assert e.value.code == _lib.QLN_ERR_INVALID_ARGUMENT and "cost" in str(e.value)
    nlp.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009)
    for bad in (dict(max_inner=0), dict(tol_violation=0.0), dict(rho_factor=1.0), dict(h_min=0.0), dict(h_max=0.0005), dict(h_prox=-1.0)):
        with pytest.raises(_lib.QlnError) as e:
            nlp.solve(Z, **bad)
        assert e.value.code == _lib.QLN_ERR_INVALID_ARGUMENT, bad
    with pytest.raises(TypeError):
        nlp.solve(Z, no_such_option=1)
    big = PG.make_batch(1, 700, 50, 1, seed=1)  # 29 N + 1.3k doubles of LDS: N <= ~650 fits the 160 KB of a CU
    nb = HybridNLP(big.model, big.obj, big.init_mode, big.k_trans, big.N, big.x0, big.xf)
    with pytest.raises(_lib.QlnError) as e:
        nb.solve(nb.upload_Z(big.Z))
    assert e.value.code == _lib.QLN_ERR_UNSUPPORTED


def test_transition_knot_extremes():
    """k_trans at the ends of its range (the jump at the first / last dynamics knot): the solver terminates with a finite
    trajectory whose roll-out rows are exact, and solves the ones that are feasible landings."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    N = 30
    base = PG.make_batch(8, N, 10, 1, seed=2, noise=0.0, build_obj=False)
    kt = np.array([2, 3, 4, N - 2, N - 1, N, 10, 15], dtype=np.int32)
    nlp = HybridNLP(base.model, None, 1, kt, N, base.x0, base.xf)
    nlp.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=True)
    Z, info = nlp.solve(nlp.initial_guess())
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    print("k_trans", kt.tolist(), "status", inf[:, 5].astype(int).tolist(), "violation", [f"{v:.1e}" for v in viol])
    assert np.all(np.isfinite(Zh)) and np.all(np.isfinite(f))
    for b in range(8):
        ci = nlp.cinds(b)
        seg = nlp.split_c(c, b)
        assert np.all(seg[: 15] == 0.0) and np.all(seg[ci[2][0] - 1 : ci[4][1]] == 0.0), b  # init, dynamics, contact rows
    assert (inf[:, 5] == 0).sum() >= 6


@pytest.mark.parametrize("N,kt,B", [(121, 41, 16), (600, 200, 2)])
def test_horizons_up_to_what_a_compute_units_lds_holds(N, kt, B):
    """N = 121 (two 64-knot chunks in every lane = knot phase) and N = 600 (143 KB of the 160 KB of LDS; N = 700 is
    refused, see test_solve_rejects_what_it_cannot_do): solved to the tolerance, roll-out rows exact."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, N, kt, 1, seed=5, noise=0.0, dt=0.009 * 40 / N)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z, info = nlp.solve(nlp.initial_guess())
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    print(f"N={N}: status {inf[:, 5].astype(int).tolist()}, iLQR iterations {inf[:, 1].astype(int).tolist()}, violation max {viol.max():.2e}")
    assert np.all(inf[:, 5] == 0) and viol.max() <= 1e-6 * 1.0001 and bviol.max() <= 1e-6
    for b in range(B):
        ci = nlp.cinds(b)
        seg = nlp.split_c(c, b)
        assert np.all(seg[:15] == 0.0) and np.all(seg[ci[2][0] - 1 : ci[4][1]] == 0.0)


@pytest.mark.parametrize("N,kt,feasible", [(2, 2, False), (3, 3, False), (4, 2, None), (64, 20, True), (65, 20, True)])
def test_tiny_and_chunk_boundary_horizons_terminate(N, kt, feasible):
    """Horizons at the edges: N = 2 / 3 (no landing is feasible in one or two steps: the solver must terminate with a
    finite trajectory and say so in the status), N = 4 (three steps of at most 0.02 s: marginal, most drop states can be
    landed, which ones depends on the penalty schedule), and N = 64 / 65 (the lane = knot phases' chunk boundary)."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(8, N, kt, 1, seed=5, noise=0.0, dt=0.009 * 40 / N)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z, info = nlp.solve(nlp.initial_guess())
    torch.cuda.synchronize()
    inf = info.cpu().numpy()
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    assert np.all(np.isfinite(Zh)) and np.all(np.isfinite(f)) and np.all(np.isin(inf[:, 5], (0, 1, 2)))
    if feasible is True:
        assert np.all(inf[:, 5] == 0) and viol.max() <= 1e-6 * 1.0001
    elif feasible is False:
        assert np.all(inf[:, 5] != 0) and viol.min() > 1e-6   # reported as not converged, and indeed not feasible
    else:
        ok = inf[:, 5] == 0
        assert ok.sum() >= 6 and viol[ok].max() <= 1e-6 * 1.0001 and viol[~ok].min(initial=1.0) > 1e-6
    for b in range(8):  # whatever the status, the returned states are the RK4 roll-out of the returned controls
        ci = nlp.cinds(b)
        seg = nlp.split_c(c, b)
        assert np.all(seg[:15] == 0.0) and np.all(seg[ci[2][0] - 1 : ci[2][1]] == 0.0)


def test_both_register_budgets_of_the_solver_kernel_give_the_same_bits():
    """k_al_ilqr is compiled twice (one or two waves per SIMD, chosen by the batch size: B > 1 024 takes the second): the
    same problems solved in a batch of 1 024 and as the first 1 024 of a batch of 1 100 must come out bit-identical --
    what a problem's solution is does not depend on how many others were solved beside it."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    big = PG.make_batch(1100, 40, 14, 1, seed=17, noise=0.0)
    out = []
    for nb in (1024, 1100):
        nlp = HybridNLP(big.model, big.obj, big.init_mode[:nb], big.k_trans[:nb], big.N, big.x0[:nb], big.xf[:nb])
        Z, info = nlp.solve(nlp.initial_guess())
        torch.cuda.synchronize()
        out.append((Z.view(nb, -1)[:1024].clone(), info[:1024, :10].clone()))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


# ---- the solver against what the reference HOLDS: its six solved trajectories (src/data_{1..6}.csv) ----------------------
def _reference_trajectory_problem(golden_dir, i):
    """The notebook problem (N = 61, k_trans = 21, init_mode 1, xf = xterm, the notebook's cost) with x0 taken from the
    file's first 15 entries (data_1..5 are consistent with that problem; their cost weights at solve time are
    unknown, so they pin feasibility only -- data_6 is the notebook's own run: x0 = xinit to 1.7e-9 and its cost IS known)."""
    import os
    from quadruped_landing_amd import problem_gen as PG

    Zf = np.loadtxt(os.path.join(golden_dir, f"data_{i}.csv"))
    nb = PG.notebook_problem()
    if i != 6:
        nb.x0[0] = Zf[:15]
    nb.Z = Zf[None, :].copy()
    return nb, Zf


def _knots(Z):
    return np.concatenate([np.asarray(Z).reshape(-1)[:1215], np.zeros(5)]).reshape(61, 20)


def test_rollout_and_report_of_the_reference_runs_controls(golden_dir):
    """What the reference's data CAN pin of the solver kernel: its roll-out, its objective and its violation measure.
    qln_solve with max_outer = 0 / rescue_outer = 0 rolls the CONTROLS of src/data_6.csv out from xinit (RK4, the
    evaluator's step) and reports.  The file's states satisfy the dynamics rows to 1.49e-6 (KA2), so the roll-out must
    retrace the file to that order -- 1.7e-6 (the CPU oracle gives the same figure) -- its objective must be KA1's
    1.1608112892558562e+02 to 3e-8 relative, and its violation stays under the reference run's own 1.4928675395736724e-06
    (src/main.ipynb:710-712).  The reported f / violation are the evaluator's, bit for bit / to the last ulp of sin()."""
    import torch
    from quadruped_landing_amd import HybridNLP

    nb, Zf = _reference_trajectory_problem(golden_dir, 6)
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    Z, info = nlp.solve(nlp.upload_Z(nb.Z), max_outer=0, rescue_outer=0)
    torch.cuda.synchronize()
    inf = info.cpu().numpy()[0]
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    X, Xf = _knots(Zh[0]), _knots(Zf)
    drift = np.abs(X[:, :15] - Xf[:, :15]).max()
    print(f"roll-out of data_6.csv's controls: max distance from the file's states {drift:.3e}; f = {f[0]:.10f} "
          f"(KA1 116.0811289256), violation {viol[0]:.3e} (reference run 1.493e-06); solver's report f {inf[2]:.10f} viol {inf[3]:.3e}")
    assert inf[0] == 0 and inf[1] == 0 and inf[5] == 1            # nothing iterated
    assert np.array_equal(X[:60, 15:], Xf[:60, 15:])               # the controls are the file's (h inside its bounds)
    assert drift <= 2e-6
    assert abs(f[0] - 1.1608112892558562e02) <= 3e-8 * 1.1608112892558562e02
    assert viol[0] <= 1.4928675395736724e-06
    assert inf[2] == f[0]                                          # reported objective = the evaluator's bits
    assert abs(inf[3] - max(viol[0], bviol[0])) <= 1e-15           # reported violation = the evaluator's (+ bounds)
    ci = nlp.cinds(0)
    for grp in (0, 2, 3, 4):                                       # rows the roll-out satisfies by construction
        assert np.all(c[ci[grp][0] - 1: ci[grp][1]] == 0.0), grp


def test_warm_start_from_the_reference_runs_own_solution(golden_dir):
    """qln_solve started from the controls of src/data_6.csv -- the point the reference's Ipopt run ENDED at, objective
    1.1608112892558562e+02, violation 1.4928675395736724e-06, exit "Restoration Failed" (src/main.ipynb:707-727): a point
    Ipopt gave up at, not a stationary point -- on the notebook problem with the notebook cost.  Required: status 0, a
    violation no larger than the reference run's, an objective no larger than the reference run's, bounds kept.
    What happens (printed): the solver does not stay there.  It walks to the same landing it finds from the notebook's
    initial guess Z0 -- f = 112.1812 from either start (they agree to 3e-6) -- which is 1.14 away from the file in the
    states (m, rad, m/s), 83 N in the forces and 3.2 ms in the step lengths; that distance is asserted as measured so that
    a change of behaviour shows.  So the file pins the solver's roll-out / objective / violation (test above) and an upper
    bound on the objective of a feasible point; the iterates remain unpinned -- the reference holds nothing for them."""
    import torch
    from quadruped_landing_amd import HybridNLP

    nb, Zf = _reference_trajectory_problem(golden_dir, 6)
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    f_file = float(nlp.eval_f(nlp.upload_Z(nb.Z)).cpu()[0])
    assert f_file == 1.1608112892558562e02                      # KA1, through the HIP path, at the starting point
    Z, info = nlp.solve(nlp.upload_Z(nb.Z))
    Z0sol, info0 = nlp.solve(nlp.initial_guess())
    torch.cuda.synchronize()
    inf = info.cpu().numpy()[0]
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    f0 = float(nlp.eval_f(Z0sol).cpu()[0])
    X, Xf, X0 = _knots(Zh[0]), _knots(Zf), _knots(Z0sol.cpu().numpy())
    d_state = np.abs(X[:, :14] - Xf[:, :14]).max()
    d_force = np.abs(X[:60, 15:19] - Xf[:60, 15:19]).max()
    d_h = np.abs(X[:60, 19] - Xf[:60, 19]).max()
    d_z0 = np.abs(X[:, :14] - X0[:, :14]).max()
    print(f"warm start from data_6.csv: outer {inf[0]:.0f}, iLQR iterations {inf[1]:.0f}, status {inf[5]:.0f}; evaluator: "
          f"f = {f[0]:.7f} (file: {f_file:.7f}; from Z0: {f0:.7f}), violation {viol[0]:.3e} (file: 1.493e-06), bound violation "
          f"{bviol[0]:.1e}; distance from the file: states {d_state:.3e}, forces {d_force:.3e} N, step lengths {d_h:.3e} s; "
          f"distance from the solution found from Z0: states {d_z0:.3e}")
    assert inf[5] == 0
    assert viol[0] <= 1.4928675395736724e-06
    assert bviol[0] <= 1e-6
    assert f[0] <= 1.1608112892558562e02 + 1e-6
    assert abs(f[0] - f0) <= 1e-4 and d_z0 <= 5e-3              # the same landing as from Z0
    assert d_state <= 1.5 and d_force <= 100.0 and d_h <= 5e-3  # as measured (1.14, 83 N, 3.2 ms): it leaves the file's
    ci = nlp.cinds(0)
    for grp in (0, 2, 3, 4):
        assert np.all(c[ci[grp][0] - 1: ci[grp][1]] == 0.0), grp


@pytest.mark.parametrize("i", [1, 2, 3, 4, 5])
def test_warm_start_from_the_references_other_trajectories_feasibility_only(golden_dir, i):
    """src/data_{1..5}.csv: also N = 61 landings, consistent with k_trans = 21 / init_mode 1 when x0 is the file's first
    state.  Their Q / R / dt at solve time are unknown, so the OBJECTIVE is not comparable (the notebook's
    cost is used, and said so); what they pin is feasibility, in two steps: (i) the evaluator finds the file feasible to
    the reference's tolerance (<= 5e-6 on every row -- except, for data_1 and data_2, the final-control row
    F1y + F2y + mb g = 0, which those two files violate by 22.7 / 2.27 N: they predate that constraint); (ii) the roll-out
    of the file's controls retraces the file (<= 4e-4: data_3's dynamics residual accumulated over 60 steps); (iii) started
    from the file's controls the solver returns status 0 and a violation <= 1e-6, judged by the evaluator."""
    import torch
    from quadruped_landing_amd import HybridNLP

    nb, Zf = _reference_trajectory_problem(golden_dir, i)
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    c0 = nlp.eval_c(nlp.upload_Z(nb.Z)).cpu().numpy()
    ci = nlp.cinds(0)
    rows = np.ones(c0.size, dtype=bool)
    if i in (1, 2):
        rows[ci[5][0] - 1] = False                                 # the final-control row (see above)
    eq = np.abs(c0[: ci[5][1]])[rows[: ci[5][1]]].max()
    v0 = max(eq, np.maximum(-c0[ci[6][0] - 1:], 0).max())
    Zr, _ = nlp.solve(nlp.upload_Z(nb.Z), max_outer=0, rescue_outer=0)
    drift = np.abs(_knots(Zr.cpu().numpy())[:, :15] - _knots(Zf)[:, :15]).max()
    Z, info = nlp.solve(nlp.upload_Z(nb.Z))
    torch.cuda.synchronize()
    inf = info.cpu().numpy()[0]
    viol, f, bviol, c, Zh = _judge(nlp, Z)
    d_state = np.abs(_knots(Zh[0])[:, :14] - _knots(Zf)[:, :14]).max()
    print(f"data_{i}.csv: file violation {v0:.3e}" + (f" (final-control row {abs(c0[ci[5][0] - 1]):.3g} left out)" if i in (1, 2) else "")
          + f", roll-out of its controls within {drift:.3e} of it; solve: status {inf[5]:.0f}, iLQR iterations {inf[1]:.0f}, "
          f"violation {viol[0]:.3e}, bound violation {bviol[0]:.1e}, f (notebook cost, not the file's) {f[0]:.4f}, max state "
          f"distance from the file {d_state:.3e}")
    assert v0 <= 5e-6
    assert drift <= 4e-4
    assert inf[5] == 0 and viol[0] <= 1e-6 * 1.0001 and bviol[0] <= 1e-6
    assert np.all(np.isfinite(Zh))


def test_solver_invariants_for_random_shapes_property():
    """Randomised (hypothesis): any horizon (2-130), ragged transition knots / contact modes, batch size, guess noise.  Whatever
    the status, (i) the solve terminates with a finite trajectory and a status in {0, 1, 2}; (ii) the returned states are the
    evaluator's RK4 roll-out of the returned controls from x0 -- initial-condition, dynamics and contact-init rows of
    qln_eval_constraint are exactly 0.0; (iii) the report is the evaluator's: info[2] = qln_eval_objective bit for bit,
    info[3] >= qln_constraint_violation (it adds solve()'s variable bounds); (iv) status 0 means what the header says:
    violation <= tol; (v) the same problem solved alone gives the same bits (no dependence on batch position)."""
    import os
    import torch
    from hypothesis import given, settings, strategies as st
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    ran = [0, 0]

    @settings(max_examples=int(os.environ.get("QLN_FUZZ_EXAMPLES", 8)), deadline=None, derandomize="QLN_FUZZ_EXAMPLES" not in os.environ)
    @given(B=st.integers(1, 12), N=st.integers(2, 130), ragged=st.booleans(), noise=st.sampled_from([0.0, 0.01, 0.05]),
           seed=st.integers(0, 10**6))
    def check(B, N, ragged, noise, seed):
        ran[0] += 1
        if N > 3 and ragged:
            batch = PG.make_batch(B, N, seed=seed, ragged=True, noise=noise, dt=0.009 * 40 / N)
        else:
            batch = PG.make_batch(B, N, max(2, min(N, N // 3 + 1)), 1 + seed % 2, seed=seed, noise=noise, dt=0.009 * 40 / N)
        nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
        Z0 = nlp.upload_Z(batch.Z)
        Z, info = nlp.solve(Z0.clone())
        torch.cuda.synchronize()
        inf = info.cpu().numpy()
        viol, f, bviol, c, Zh = _judge(nlp, Z)
        assert np.all(np.isfinite(Zh)) and np.all(np.isfinite(f)) and np.all(np.isin(inf[:, 5], (0, 1, 2)))
        assert np.array_equal(inf[:, 2], f), "reported objective is not the evaluator's"
        assert np.all(inf[:, 3] >= viol) and np.all(inf[:, 3] <= np.maximum(viol, bviol) * (1 + 1e-12) + 1e-300)
        ok = inf[:, 5] == 0
        ran[1] += int(ok.sum())
        assert np.all(inf[ok, 3] <= 1e-6 * 1.0001)
        for b in range(B):
            ci = nlp.cinds(b)
            seg = nlp.split_c(c, b)
            assert np.all(seg[:15] == 0.0) and np.all(seg[ci[2][0] - 1 : ci[2][1]] == 0.0)
        # batch-position independence: the last problem alone
        b = B - 1
        one = HybridNLP(batch.model, batch.obj if batch.obj.ndim == 2 else batch.obj[b : b + 1], batch.init_mode[b : b + 1],
                        batch.k_trans[b : b + 1], batch.N, batch.x0[b : b + 1], batch.xf[b : b + 1])
        Z1, info1 = one.solve(one.upload_Z(batch.Z[b : b + 1]))
        torch.cuda.synchronize()
        assert torch.equal(Z1.view(-1)[: nlp.n_nlp], Z.view(B, -1)[b, : nlp.n_nlp]) and torch.equal(info1[0, :10], info[b, :10])

    check()
    print(f"{ran[0]} examples, {ran[1]} problems solved to tolerance")
