"""GPU parity of qln_gauss_newton_step (the solver iteration on the GPU).

The reference has no such step (it hands its callbacks to Ipopt), so, as SURVEY.md 8f-2 says, there is no reference
oracle for the iterates; what pins the kernel is
  * the same CGLS recurrence in numpy on the ORACLE's Jacobian (forward-mode duals) and constraint values,
    iterate for iterate over the first three iterations, to 1e-8 of the step's size (later CGLS iterates amplify the
    ~1e-11 relative difference between the closed-form and the forward-mode Jacobian entries by many orders of
    magnitude -- a property of the Krylov recurrence that numpy shows as well when its own matrix is perturbed);
  * the converged step against numpy.linalg.lstsq's minimum-norm least-squares solution of the same system;
  * the trust-radius cut and the column scaling against the same Steihaug-CGLS in numpy;
  * used in the outer trust-region loop of examples/feasibility_trust_region.py, every accepted step lowers the merit
    ||rho||^2 of its problem and the reference's constraint violation (KA2's definition, KA5's 3.13e-01 at the
    notebook's initial guess) falls by more than 10x on the notebook problem -- a property check, not a solver claim.
"""
import numpy as np
import pytest

from tests.helpers import oracle_batch

pytestmark = pytest.mark.gpu


def _system(batch, nlp, ref, b):
    """(A, rho) of problem b from the oracle's c and Jacobian: equality rows as they are, clearance rows only if violated."""
    import scipy.sparse as sp

    m, nnz = nlp.problem_dims(b)
    rows, cols = nlp.jacobian_structure(b)
    vals = ref["vals"][nlp.j_off[b] : nlp.j_off[b] + nnz]
    c = ref["c"][nlp.c_off[b] : nlp.c_off[b] + m]
    A = sp.coo_matrix((vals, (rows, cols)), shape=(m, nlp.n_nlp)).tocsr()
    ineq = np.arange(m) >= m - batch.N
    active = ~ineq | (c < 0)
    rho = np.where(active, c, 0.0)
    A = sp.diags(active.astype(float)) @ A
    return A, rho


def _cgls(A, rho, iters):
    x = np.zeros(A.shape[1])
    r = -rho.copy()
    s = A.T @ r
    p = s.copy()
    gamma = s @ s
    for _ in range(iters):
        q = A @ p
        alpha = gamma / (q @ q)
        x += alpha * p
        r -= alpha * q
        s = A.T @ r
        gnew = s @ s
        p = s + (gnew / gamma) * p
        gamma = gnew
    return x, r, gamma


def _setup(batch):
    import torch
    from quadruped_landing_amd import HybridNLP

    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    c = nlp.eval_c(Z)
    return torch, nlp, Z, c


@pytest.mark.parametrize("B,N,ragged", [(3, 40, False), (5, 17, True), (2, 80, True), (2, 3, False)])
def test_iterates_follow_numpy_cgls_on_the_oracle_jacobian(B, N, ragged):
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(B, N, max(2, N // 3), 1, seed=N, ragged=ragged)
    torch, nlp, Z, c = _setup(batch)
    ref = oracle_batch(batch, nlp)
    iters = 3
    info = torch.zeros(8 * B, dtype=torch.float64, device="cuda")
    dZ = nlp.gauss_newton_step(Z, c, max_iters=iters, rel_tol=0.0, info=info)
    torch.cuda.synchronize()
    dZ, info = dZ.cpu().numpy().reshape(B, -1), info.cpu().numpy().reshape(B, 8)
    for b in range(B):
        A, rho = _system(batch, nlp, ref, b)
        x, r, gamma = _cgls(A, rho, iters)
        err = np.abs(dZ[b, : nlp.n_nlp] - x).max() / np.abs(x).max()
        print(f"N={N} problem {b}: step max {np.abs(x).max():.3e}, rel err after {iters} CGLS iterations {err:.3e}")
        assert err <= 1e-8
        assert info[b, 0] == iters
        g0 = (A.T @ rho) @ (A.T @ rho)
        assert abs(info[b, 1] - g0) <= 1e-9 * g0
        assert abs(info[b, 3] - r @ r) <= 1e-6 * (r @ r) + 1e-300
        assert abs(info[b, 2] - gamma) <= 1e-4 * gamma + 1e-300
        assert abs(info[b, 4] - rho @ rho) <= 1e-12 * (rho @ rho) and info[b, 5] == 0
        assert abs(info[b, 6] - np.linalg.norm(x)) <= 1e-8 * np.linalg.norm(x)


def test_converged_step_is_the_minimum_norm_least_squares_solution():
    from quadruped_landing_amd import problem_gen as PG

    # init_mode 2 with the generator's mode-1 initial states pins a foot 0.5 m away from where the terminal state wants
    # it: an inconsistent linear system, so the least-squares residual stays > 0 -- the case lstsq and CGLS must agree on
    batch = PG.make_batch(3, 12, 5, 2, seed=8)
    torch, nlp, Z, c = _setup(batch)
    ref = oracle_batch(batch, nlp)
    info = torch.zeros(8 * batch.B, dtype=torch.float64, device="cuda")
    dZ = nlp.gauss_newton_step(Z, c, max_iters=5000, rel_tol=1e-13, info=info)
    torch.cuda.synchronize()
    dZ, info = dZ.cpu().numpy().reshape(batch.B, -1), info.cpu().numpy().reshape(batch.B, 8)
    for b in range(batch.B):
        A, rho = _system(batch, nlp, ref, b)
        x_ls, *_ = np.linalg.lstsq(A.toarray(), -rho, rcond=1e-13)
        res_ls = np.linalg.norm(A @ x_ls + rho)
        res = np.linalg.norm(A @ dZ[b, : nlp.n_nlp] + rho)
        err = np.abs(dZ[b, : nlp.n_nlp] - x_ls).max() / np.abs(x_ls).max()
        print(f"problem {b}: {int(info[b, 0])} iterations, residual {res:.3e} (lstsq {res_ls:.3e}), step rel err {err:.3e}")
        assert res <= res_ls * (1 + 1e-6) + 1e-9
        assert err <= 1e-5


def _steihaug_cgls(A, rho, radius, iters):
    x = np.zeros(A.shape[1])
    r = -rho.copy()
    s = A.T @ r
    p = s.copy()
    gamma = s @ s
    for k in range(iters):
        q = A @ p
        alpha = gamma / (q @ q)
        if np.linalg.norm(x + alpha * p) >= radius:
            xp, pp, xx = x @ p, p @ p, x @ x
            tau = (-xp + np.sqrt(xp * xp + pp * (radius * radius - xx))) / pp
            return x + tau * p, r - tau * q, k + 1, True
        x += alpha * p
        r -= alpha * q
        s = A.T @ r
        gnew = s @ s
        p = s + (gnew / gamma) * p
        gamma = gnew
    return x, r, iters, False


def test_trust_radius_and_column_scaling_follow_numpy():
    import scipy.sparse as sp
    import torch
    from examples.feasibility_trust_region import default_col_scale
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(4, 20, 7, 1, seed=4)
    torch, nlp, Z, c = _setup(batch)
    ref = oracle_batch(batch, nlp)
    d = default_col_scale(batch.N)
    d[3::20] = 0.0  # hold foot 1's x position fixed at every knot
    systems = [_system(batch, nlp, ref, b) for b in range(batch.B)]
    # radii chosen from numpy's own iterates so that problem b leaves the ball during iteration b+1; the last never does
    radii = []
    for b, (A, rho) in enumerate(systems):
        AD = A @ sp.diags(d)
        xb, *_ = _cgls(AD, rho, b + 1)
        xa, *_ = _cgls(AD, rho, b) if b else (np.zeros(AD.shape[1]),)
        radii.append(0.5 * (np.linalg.norm(xa) + np.linalg.norm(xb)) if b < batch.B - 1 else 1e30)
    info = torch.zeros(8 * batch.B, dtype=torch.float64, device="cuda")
    dZ = nlp.gauss_newton_step(Z, c, max_iters=4, rel_tol=0.0, radius=torch.tensor(radii, dtype=torch.float64, device="cuda"),
                               col_scale=torch.from_numpy(d).cuda(), info=info)
    torch.cuda.synchronize()
    dZ, info = dZ.cpu().numpy().reshape(batch.B, -1), info.cpu().numpy().reshape(batch.B, 8)
    for b, (A, rho) in enumerate(systems):
        x, r, k, hit = _steihaug_cgls(A @ sp.diags(d), rho, radii[b], 4)
        err = np.abs(dZ[b, : nlp.n_nlp] - d * x).max() / np.abs(d * x).max()
        print(f"problem {b}: radius {radii[b]:.3e}, {k} iterations, cut={hit}, rel err {err:.3e}")
        assert err <= 1e-8
        assert info[b, 0] == k and bool(info[b, 5]) == hit
        assert np.all(dZ[b, 3 : nlp.n_nlp : 20] == 0.0)
        assert abs(info[b, 6] - np.linalg.norm(x)) <= 1e-8 * np.linalg.norm(x)
        if hit:
            assert abs(info[b, 6] - radii[b]) <= 1e-12 * radii[b]
        assert abs(info[b, 3] - r @ r) <= 1e-7 * (r @ r)


def test_trust_region_loop_lowers_merit_and_violation_of_the_notebook_problem():
    """examples/feasibility_trust_region.py on the notebook problem (N = 61, k_trans = 21) from its initial guess."""
    import torch
    from examples.feasibility_trust_region import trust_region_feasibility
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.notebook_problem()
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.initial_guess()
    hist = trust_region_feasibility(nlp, Z, steps=25)
    torch.cuda.synchronize()
    print("notebook problem, max violation per step:", " ".join(f"{v:.2e}" for v, _ in hist))
    assert abs(hist[0][0] - 0.3132) < 1e-3  # KA5: max |c(Z0)| = 3.13e-01 (src/main.ipynb:232)
    merit = [m for _, m in hist]
    assert all(m1 <= m0 * (1 + 1e-12) for m0, m1 in zip(merit, merit[1:]))  # rejected steps leave Z untouched
    assert hist[-1][0] <= hist[0][0] / 10


def test_trust_region_loop_on_a_batch():
    import torch
    from examples.feasibility_trust_region import trust_region_feasibility
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(256, 40, 14, 1, seed=2, noise=0.0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    hist = trust_region_feasibility(nlp, Z, steps=20)
    torch.cuda.synchronize()
    print("B=256 N=40, max violation over the batch per step:", " ".join(f"{v:.2e}" for v, _ in hist))
    merit = [m for _, m in hist]
    assert all(m1 <= m0 * (1 + 1e-12) for m0, m1 in zip(merit, merit[1:]))
    assert hist[-1][1] <= hist[0][1] / 10


def test_satisfied_clearance_rows_are_left_out_and_nan_stays_local():
    from quadruped_landing_amd import problem_gen as PG

    batch = PG.make_batch(4, 10, 4, 1, seed=6)
    batch.Z[1, 1::20] = -0.5   # problem 1: body below the ground at every knot -> all clearance rows active
    batch.Z[2, 1::20] = 5.0    # problem 2: far above -> none active
    batch.Z[3, 7] = np.nan     # problem 3: a NaN in the state
    torch, nlp, Z, c = _setup(batch)
    ref = oracle_batch(batch, nlp)
    info = torch.zeros(8 * batch.B, dtype=torch.float64, device="cuda")
    dZ = nlp.gauss_newton_step(Z, c, max_iters=3, rel_tol=0.0, info=info)
    torch.cuda.synchronize()
    dZ = dZ.cpu().numpy().reshape(batch.B, -1)
    for b in (0, 1, 2):
        A, rho = _system(batch, nlp, ref, b)
        x, *_ = _cgls(A, rho, 3)
        assert np.abs(dZ[b, : nlp.n_nlp] - x).max() <= 1e-8 * np.abs(x).max()
    m = nlp.problem_dims(1)[0]
    assert np.all(ref["c"][nlp.c_off[1] + m - batch.N : nlp.c_off[1] + m] < 0)
    assert np.all(ref["c"][nlp.c_off[2] + nlp.problem_dims(2)[0] - batch.N : nlp.c_off[2] + nlp.problem_dims(2)[0]] > 0)
    assert np.all(np.isfinite(dZ[:3, : nlp.n_nlp]))


def test_problems_too_large_for_lds_are_refused():
    from quadruped_landing_amd import HybridNLP, problem_gen as PG
    from quadruped_landing_amd._lib import QLN_ERR_UNSUPPORTED, QlnError

    batch = PG.make_batch(1, 200, 60, 1, seed=0)
    torch, nlp, Z, c = _setup(batch)
    with pytest.raises(QlnError) as e:
        nlp.gauss_newton_step(Z, c)
    assert e.value.code == QLN_ERR_UNSUPPORTED and "LDS" in str(e.value)
    # the largest horizon that fits
    batch = PG.make_batch(2, 149, 60, 1, seed=0)
    torch, nlp, Z, c = _setup(batch)
    dZ = nlp.gauss_newton_step(Z, c, max_iters=3, rel_tol=0.0)
    torch.cuda.synchronize()
    assert torch.isfinite(dZ.view(2, -1)[:, : nlp.n_nlp]).all()


def test_gauss_newton_random_shapes_property():
    """Randomised shapes (hypothesis): both code paths of the step kernel (step blocks cached in registers for N <= 64,
    re-derived for larger N), ragged k_trans / init_mode, strides and alignments, with and without scaling / radius."""
    import scipy.sparse as sp
    import torch
    from hypothesis import given, settings, strategies as st
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    @settings(max_examples=int(__import__("os").environ.get("QLN_FUZZ_EXAMPLES", 12)), deadline=None)
    @given(B=st.integers(1, 6), N=st.integers(2, 100), pad=st.integers(0, 5), align=st.sampled_from([1, 2, 16]),
           scaled=st.booleans(), seed=st.integers(0, 10**6))
    def check(B, N, pad, align, scaled, seed):
        batch = PG.make_batch(B, N, seed=seed, ragged=True) if N > 3 else PG.make_batch(B, N, 2, 1 + seed % 2, seed=seed)
        nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                        z_stride=(20 * N - 5 + pad) if pad else 0, align=align)
        Z = nlp.upload_Z(batch.Z)
        c = nlp.eval_c(Z)
        ref = oracle_batch(batch, nlp)
        rng = np.random.default_rng(seed)
        d = rng.uniform(0.5, 2.0, size=nlp.n_nlp) if scaled else np.ones(nlp.n_nlp)
        if scaled:
            d[rng.integers(0, nlp.n_nlp, size=3)] = 0.0
        systems = [_system(batch, nlp, ref, b) for b in range(B)]
        radii = [0.7 * np.linalg.norm(_cgls(A @ sp.diags(d), rho, 2)[0]) if scaled else 1e30 for A, rho in systems]
        dZ = nlp.gauss_newton_step(Z, c, max_iters=2, rel_tol=0.0,
                                   radius=torch.tensor(radii, dtype=torch.float64, device="cuda") if scaled else None,
                                   col_scale=torch.from_numpy(d).cuda() if scaled else None)
        torch.cuda.synchronize()
        dZ = dZ.cpu().numpy().reshape(B, -1)
        for b, (A, rho) in enumerate(systems):
            x, *_ = _steihaug_cgls(A @ sp.diags(d), rho, radii[b], 2)
            assert np.abs(dZ[b, : nlp.n_nlp] - d * x).max() <= 1e-8 * np.abs(d * x).max(), (B, N, b)

    check()
