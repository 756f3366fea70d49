"""The opt-in leg-length ("kinematic") rows (qln_eval_kinematic_constraint).  NO REFERENCE ORACLE: the reference carries
this group only as commented-out code (src/constraints.jl:115-138, 276-288; src/nlp.jl:60,70) and never computes it, so
neither the oracle nor any reference-held number covers it.  The values follow the commented source's definition and are
checked against an independent numpy statement of it; the Jacobian is the mathematically correct one (the commented
one indexes the wrong state slots) and is checked by complex-step differentiation of the numpy statement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(Z, N):
    """d[2k] = |pb_k - p1_k|, d[2k+1] = |pb_k - p2_k| for one problem; works for complex Z (complex-step)."""
    X = np.concatenate([Z[: 20 * (N - 1)].reshape(N - 1, 20)[:, :15], Z[None, 20 * (N - 1) :]], axis=0)
    pb, p1, p2 = X[:, 0:2], X[:, 3:5], X[:, 5:7]
    n = lambda v: np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2)
    return np.stack([n(pb - p1), n(pb - p2)], axis=1).reshape(-1)


@pytest.mark.parametrize("B,N", [(7, 40), (3, 61), (2, 2)])
def test_kinematic_rows_and_their_jacobian(B, N):
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, N, min(14, N), 1, seed=N)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    d, jac, (lo, up) = nlp.kinematic_constraint(nlp.upload_Z(batch.Z))
    torch.cuda.synchronize()
    d, jac = d.cpu().numpy(), jac.cpu().numpy()
    m = batch.model
    assert lo == 0.0 and up == m.l1 + m.l2 + m.lb / 2
    for b in range(B):
        want = _rows(batch.Z[b], N)
        assert np.max(np.abs(d[b] - want)) <= 1e-15 * np.max(want)
        # complex-step derivative of the numpy statement, column by column of the columns the rows can depend on
        for k in range(N):
            for foot, cols in ((0, (0, 1, 3, 4)), (1, (0, 1, 5, 6))):
                for q, c in enumerate(cols):
                    Zc = batch.Z[b].astype(complex)
                    Zc[20 * k + c] += 1e-30j
                    deriv = _rows(Zc, N).imag / 1e-30
                    assert abs(deriv[2 * k + foot] - jac[b, 2 * k + foot, q]) <= 1e-13
                    # the row depends on nothing else in that column's knot but what the four slots cover
                    other = np.delete(deriv, 2 * k + foot)
                    if c in (0, 1):  # pb enters both rows of the knot
                        assert np.count_nonzero(other) <= 1
                    else:
                        assert np.count_nonzero(other) == 0


@pytest.mark.parametrize("B,N,ragged", [(9, 40, False), (64, 30, True), (2, 2, False)])
def test_friction_pyramid_rows(B, N, ragged):
    """The opt-in friction group (qln_eval_friction_cone).  NO REFERENCE ORACLE: the reference has no friction constraint.
    Values against a numpy statement of |F_x| <= mu F_y as two linear rows per standing foot, with the standing feet
    taken from the mode schedule of the dynamics rows (src/constraints.jl:23-37); the Jacobian is constant."""
    import torch
    from quadruped_landing_amd import HybridNLP, _lib, problem_gen as PG

    batch = PG.make_batch(B, N, min(14, N), 1, seed=N, ragged=ragged)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    mu = 0.7
    d, jac = nlp.friction_cone(Z, mu)
    d0, none = nlp.friction_cone(Z, mu, with_jacobian=False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(d, d0)
    d, jac = d.cpu().numpy(), jac.cpu().numpy()
    U = batch.Z[:, : 20 * (N - 1)].reshape(B, N - 1, 20)[:, :, 15:20]
    K = np.arange(1, N)[None, :]                                   # 1-based dynamics knot
    mode = np.where(K <= batch.k_trans[:, None] - 1, batch.init_mode[:, None], 3)
    on = np.stack([mode != 2, mode != 1], axis=2)                  # foot 1 / foot 2 stands on the ground
    want = np.zeros((B, N - 1, 4))
    wj = np.zeros((B, N - 1, 4, 2))
    for foot in (0, 1):
        fx, fy = U[:, :, 2 * foot], U[:, :, 2 * foot + 1]
        want[:, :, 2 * foot] = np.where(on[:, :, foot], mu * fy - fx, 0.0)
        want[:, :, 2 * foot + 1] = np.where(on[:, :, foot], mu * fy + fx, 0.0)
        wj[:, :, 2 * foot] = np.where(on[:, :, foot, None], [-1.0, mu], 0.0)
        wj[:, :, 2 * foot + 1] = np.where(on[:, :, foot, None], [1.0, mu], 0.0)
    assert np.array_equal(d, want) and np.array_equal(jac, wj)
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        with pytest.raises(_lib.QlnError) as e:
            nlp.friction_cone(Z, bad)
        assert e.value.code == _lib.QLN_ERR_INVALID_ARGUMENT


def test_solved_landings_against_the_friction_pyramid():
    """What the opt-in group says about the trajectories qln_solve returns for the reference NLP (which has no friction
    constraint): reported, and only the sign of the normal force of the standing feet is asserted loosely -- the NLP's
    cost pulls the vertical forces towards the weight-bearing reference (src/ref_traj.jl:19-34), so landed feet push."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(128, 40, 14, 1, seed=9, noise=0.0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z, info = nlp.solve(nlp.initial_guess())
    d, _ = nlp.friction_cone(Z, 1.0, with_jacobian=False)
    torch.cuda.synchronize()
    d = d.cpu().numpy()
    ok = info.cpu().numpy()[:, 5] == 0
    fy2 = (d[ok][:, :, 0] + d[ok][:, :, 1]) / 2  # mu F1y with mu = 1: the normal force of foot 1 (standing at every knot)
    frac_in_cone = float((d[ok].min(axis=2) >= -1e-9).mean())
    print(f"solved {int(ok.sum())} / 128; knots of the solutions inside the mu = 1 pyramid: {100 * frac_in_cone:.1f} %; "
          f"smallest normal force of the first foot after touchdown of both: {fy2[:, 14:].min():.2f} N")
    assert ok.mean() >= 0.98
    assert np.median(fy2[:, 14:]) > 0.0
