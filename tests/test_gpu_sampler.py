"""The synthetic workload generated on the device: qln_sample_drop_states draws the per-problem
drop states with the SAME PCG64 stream positions numpy.random.default_rng(seed) consumes in problem_gen.make_batch, so x0
is bit-identical to the host generator's; qln_perturb_point's Gaussian noise is Box-Muller on that kind of stream --
the recipe's distribution, not numpy's numbers (checked as a distribution)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,seed,im", [(1, 0, 1), (1000, 0, 1), (4097, 7, 2), (65536, 3, 1)])
def test_device_drop_states_are_bit_identical_to_the_host_generator(B, seed, im):
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    host = PG.make_batch(B, 12, 5, im, seed=seed, build_obj=False)
    nlp = HybridNLP(host.model, None, im, 5, 12, np.zeros((B, 15)), host.xf)  # created with placeholder x0
    x0 = nlp.sample_drop_states(PG.drop_state_sampler(seed, host.model))
    assert np.array_equal(x0, host.x0)
    # and everything built from x0 on the device is the host's: the notebook's initial guess
    Z0 = nlp.initial_guess().cpu().numpy().reshape(B, -1)
    from quadruped_landing_amd.ref_traj import reference_trajectory
    _, Uref = reference_trajectory(host.model, 12, host.k_trans, host.xf, host.init_mode, 0.009)
    assert np.array_equal(Z0, PG.initial_guess(12, host.k_trans, host.x0, host.xf, Uref))


def test_perturbed_point_has_the_recipes_distribution():
    """Z0 + N(0, 0.05^2) on every entry but h, h clipped to [0.001, 0.02]: mean / standard deviation / tails / lag-1
    correlation of the added noise over 4e6 draws, and the two h policies."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    B, N = 5000, 40
    host = PG.make_batch(B, N, 14, 1, seed=1, build_obj=False)
    nlp = HybridNLP(host.model, None, 1, 14, N, host.x0, host.xf)
    s = PG.drop_state_sampler(1, host.model, stream_offset=4 * B)
    Z0 = nlp.initial_guess()
    Z = nlp.perturb_point(Z0.clone(), s, sigma=0.05)
    torch.cuda.synchronize()
    d = (Z - Z0).cpu().numpy().reshape(B, -1)
    hcols = 19 + 20 * np.arange(N - 1)
    mask = np.ones(d.shape[1], dtype=bool)
    mask[hcols] = False
    g = d[:, mask].reshape(-1) / 0.05
    n = g.size
    assert abs(g.mean()) < 5 / np.sqrt(n) and abs(g.std() - 1) < 5 / np.sqrt(2 * n)
    assert abs(np.mean(np.abs(g) > 1.959964) - 0.05) < 5 * np.sqrt(0.05 * 0.95 / n)
    assert abs(np.mean(g ** 4) - 3) < 0.05 and abs(np.mean(g ** 3)) < 0.02
    assert abs(np.corrcoef(g[:-1], g[1:])[0, 1]) < 5 / np.sqrt(n)
    hz = Z.cpu().numpy().reshape(B, -1)[:, hcols]
    assert hz.min() >= 0.001 and hz.max() <= 0.02
    # deterministic, and a different stream position gives different noise
    Z2 = nlp.perturb_point(Z0.clone(), s, sigma=0.05)
    assert torch.equal(Z, Z2)
    s2 = PG.drop_state_sampler(1, host.model, stream_offset=4 * B + 2)
    assert not torch.equal(Z, nlp.perturb_point(Z0.clone(), s2, sigma=0.05))
    # the ragged workload's policy: h ~ U(0.001, 0.02)
    Zr = nlp.perturb_point(Z0.clone(), s, sigma=0.05, redraw_h=True).cpu().numpy().reshape(B, -1)[:, hcols]
    assert Zr.min() >= 0.001 and Zr.max() <= 0.02 and abs(Zr.mean() - 0.0105) < 1e-4


@pytest.mark.parametrize("B,N,seed", [(1001, 80, 0), (4096, 80, 7), (333, 12, 3), (65536, 80, 5)])
def test_ragged_workload_generated_on_the_device_is_the_host_generators(B, N, seed):
    """config 4: k_trans ~ U{2..N-1} and init_mode ~ U{1,2} drawn on the device with numpy's own
    algorithm and stream positions (qln_sample_bounded_integers), the drop states continuing the stream behind them: all
    three bit-identical to problem_gen.make_batch(ragged=True), with no rejection reported."""
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    host = PG.make_batch(B, N, seed=seed, ragged=True, build_obj=False)
    kt, im, off = PG.ragged_descriptors(seed, B, N, device=0)
    assert off == B and np.array_equal(kt, host.k_trans) and np.array_equal(im, host.init_mode)
    nlp = HybridNLP(host.model, None, im, kt, N, np.zeros((B, 15)), host.xf)
    x0 = nlp.sample_drop_states(PG.drop_state_sampler(seed, host.model, stream_offset=off))
    assert np.array_equal(x0, host.x0)      # incl. the mirrored feet of the init_mode 2 problems


def test_a_rejected_draw_is_reported_by_the_device_generator():
    """Seeds 6076 / 6979 make numpy reject one k_trans draw at B = 65 536, N = 80 (tests/test_host_logic.py): the device
    generator counts it, and the caller falls back to the host."""
    from quadruped_landing_amd import problem_gen as PG

    for seed in (6076, 6979):
        kt, im, off = PG.ragged_descriptors(seed, 65536, 80, device=0)
        assert off is None
        assert PG.ragged_descriptors(seed, 65536, 80)[2] is None        # the host check agrees
    assert PG.ragged_descriptors(6075, 65536, 80, device=0)[2] == 65536
