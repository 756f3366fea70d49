"""BASELINE.json's full-size configurations on the GPU: size-independent properties plus a sampled
comparison with the oracle, and (last test) EVERY problem of both full-size configurations against the batched oracle."""
import numpy as np
import pytest

from tests.helpers import oracle_model, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def _setup(B, N, ragged, seed=0):
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(B, N, 14, 1, seed=seed, ragged=ragged)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                    stream=torch.cuda.current_stream())
    return batch, nlp, nlp.upload_Z(batch.Z)


def _oracle_sample(batch, idx):
    from oracle import oracle as O

    out = []
    for b in idx:
        obj = batch.obj if batch.obj.ndim == 2 else batch.obj[b]
        o = O.OracleNLP(batch.N, int(batch.k_trans[b]), int(batch.init_mode[b]), batch.x0[b], batch.xf[b], obj,
                        oracle_model(batch.model))
        out.append((o.eval_c(batch.Z[b]), o.jac_c_coo(batch.Z[b]), o.eval_f(batch.Z[b]), o.grad_f(batch.Z[b])))
    return out


@pytest.mark.parametrize("B,N,ragged", [(65536, 40, False), (65536, 80, True)])
def test_full_size_configs(B, N, ragged):
    import torch

    batch, nlp, Z = _setup(B, N, ragged)
    nan = float("nan")
    c = torch.full((nlp.dims.c_total,), nan, dtype=torch.float64, device="cuda")
    v = torch.full((nlp.dims.j_total,), nan, dtype=torch.float64, device="cuda")
    nlp.eval_c_and_jac(Z, c, v, write_constants=True)
    f, g = nlp.eval_f(Z), nlp.grad_f(Z)
    torch.cuda.synchronize()

    # 1. every slot of every problem is written, padding is untouched
    m = np.array([18 * N - k + 16 for k in batch.k_trans])
    nz = np.array([300 * (N - 1) + N + 435 + 15 * (N - 1) + 3 * N - k + 3 for k in batch.k_trans])
    assert int(torch.isnan(c).sum()) == nlp.dims.c_total - int(m.sum())
    assert int(torch.isnan(v).sum()) == nlp.dims.j_total - int(nz.sum())

    # 2. sampled comparison with the oracle, incl. the first and last problem
    rng = np.random.default_rng(1)
    idx = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, size=126)]))
    ch, vh, fh, gh = c.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    worst = 0.0
    for b, (oc, ov, of, og) in zip(idx, _oracle_sample(batch, idx)):
        gc, gv = nlp.split_c(ch, b), nlp.split_vals(vh, b)
        assert np.array_equal(gv == 0, ov == 0)  # sparsity bit-exact
        worst = max(worst, rel_err(gc, oc, floor=1.0), rel_err(gv, ov, floor=1e-300), abs(fh[b] - of) / abs(of),
                    rel_err(gh[b * nlp.z_stride : b * nlp.z_stride + nlp.n_nlp], og, floor=1e-300))
    print(f"B={B} N={N}: worst rel err over {len(idx)} sampled problems = {worst:.3e}")
    assert worst <= RTOL

    # 3. idempotence and fused == separate entry points (bitwise)
    c2, v2 = nlp.eval_c(Z), nlp.jac_c(Z, write_constants=True)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(c, nan=0.0), c2)
    assert torch.equal(torch.nan_to_num(v, nan=0.0), v2)
    del c2, v2

    # 4. the step blocks are the derivative of the dynamics rows: J*d vs central differences of c,
    #    over the whole batch, on the GPU (uniform layout only)
    if not ragged:
        nb = 4096
        eps = 1e-6
        d = torch.randn(B, nlp.z_stride, dtype=torch.float64, device="cuda")
        d[:, nlp.n_nlp:] = 0
        d[:, 19::20] *= 1e-3  # keep h positive
        cp = nlp.eval_c((Z.view(B, -1) + eps * d).reshape(-1).contiguous())
        cm = nlp.eval_c((Z.view(B, -1) - eps * d).reshape(-1).contiguous())
        stride_c = nlp.c_off[1] - nlp.c_off[0]
        fd = ((cp - cm) / (2 * eps))[: nb * stride_c].view(nb, stride_c)[:, 29 : 29 + 15 * (N - 1)].reshape(nb, N - 1, 15)
        stride_j = nlp.j_off[1] - nlp.j_off[0]
        blocks = v[: nb * stride_j].view(nb, stride_j)[:, : 300 * (N - 1)].reshape(nb, N - 1, 20, 15)  # column-major 15x20
        dz = d[:nb, : 20 * (N - 1)].reshape(nb, N - 1, 20)
        nxt = (20 * (torch.arange(N - 1, device="cuda") + 1))[:, None] + torch.arange(15, device="cuda")[None, :]
        dxn = d[:nb][:, nxt]  # perturbation of x_{k+1}: (nb, N-1, 15)
        jd = torch.einsum("bkcr,bkc->bkr", blocks, dz) - dxn
        # quirk Q1: the reference zeroes the clock row of the jump knot's block; exclude that one entry
        kj = int(batch.k_trans[0]) - 2
        err = (jd - fd).abs()
        err[:, kj, 14] = 0
        assert float(err.max()) <= 1e-5 * max(1.0, float(fd.abs().max()))


def test_permutation_equivariance():
    """Problems are independent: evaluating a permuted batch permutes the outputs, bit for bit."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    batch = PG.make_batch(4096, 40, seed=2, ragged=True)
    perm = np.random.default_rng(0).permutation(batch.B)

    def run(ix):
        nlp = HybridNLP(batch.model, batch.obj[ix], batch.init_mode[ix], batch.k_trans[ix], batch.N, batch.x0[ix], batch.xf[ix])
        Z = nlp.upload_Z(batch.Z[ix])
        c, v = nlp.eval_c_and_jac(Z)
        f = nlp.eval_f(Z)
        torch.cuda.synchronize()
        return nlp, c.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy()

    n0, c0, v0, f0 = run(np.arange(batch.B))
    n1, c1, v1, f1 = run(perm)
    assert np.array_equal(f1, f0[perm])
    for i in (0, 1, 17, 4095):
        assert np.array_equal(n1.split_c(c1, i), n0.split_c(c0, perm[i]))
        assert np.array_equal(n1.split_vals(v1, i), n0.split_vals(v0, perm[i]))


def test_full_size_structural_format_and_products_agree_with_the_dense_blocks():
    """B = 65 536, N = 40 (BASELINE.json configs[2]), every problem: (1) the structural format holds bitwise the
    entries of the dense blocks at the positions qln_jacobian_structure lists, and nothing it leaves out is non-zero;
    (2) J v and J' lam (Jacobian re-derived in registers) equal the products formed from the evaluator's own stored
    Jacobian values."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    B, N = 65536, 40
    batch = PG.make_batch(B, N, 14, 1, seed=3)
    mk = lambda fmt: HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                               stream=torch.cuda.current_stream(), jac_format=fmt)
    nd, ns = mk("dense_blocks"), mk("structural")
    Z = nd.upload_Z(batch.Z)
    cd, vd = nd.eval_c_and_jac(Z, write_constants=True)
    cs, vs = ns.eval_c_and_jac(Z, write_constants=True)
    torch.cuda.synchronize()
    assert torch.equal(cd, cs)

    def per_problem(nlp, vals):
        stride = int(nlp.j_off[1] - nlp.j_off[0])
        nnz = nlp.problem_dims(0)[1]
        # the last problem's segment is not padded to the stride: view the first B-1 problems as a matrix
        return vals[: (B - 1) * stride].view(B - 1, stride)[:, :nnz], nnz

    Vd, nnz_d = per_problem(nd, vd)
    Vs, nnz_s = per_problem(ns, vs)
    rd, cd_ = nd.jacobian_structure(0)
    rs, cs_ = ns.jacobian_structure(0)
    n, m = nd.n_nlp, nd.problem_dims(0)[0]
    pos_d = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(rd, cd_))}
    idx = torch.tensor([pos_d[(int(r), int(c))] for r, c in zip(rs, cs_)], device="cuda")
    assert torch.equal(Vd[:, idx], Vs)                                  # (1) same values, bitwise, for 65 535 problems
    left_out = torch.ones(nnz_d, dtype=torch.bool, device="cuda")
    left_out[idx] = False
    assert int((Vd[:, left_out] != 0).sum()) == 0                       #     and only zeros are left out

    # (2) products against the stored Jacobian, all problems at once
    gen = torch.Generator(device="cuda").manual_seed(1)
    v = torch.randn(nd.dims.z_total, dtype=torch.float64, device="cuda", generator=gen)
    lam = torch.randn(nd.dims.c_total, dtype=torch.float64, device="cuda", generator=gen)
    y, g = ns.jac_vec(Z, v), ns.jac_t_vec(Z, lam)
    torch.cuda.synchronize()
    rows = torch.from_numpy(rs.astype(np.int64)).cuda()
    cols = torch.from_numpy(cs_.astype(np.int64)).cuda()
    cstride = int(nd.c_off[1] - nd.c_off[0])
    Vm = v.view(B, -1)[: B - 1, :n]
    Lm = lam[: (B - 1) * cstride].view(B - 1, cstride)[:, :m]
    y_ref = torch.zeros(B - 1, m, dtype=torch.float64, device="cuda").index_add_(1, rows, Vs * Vm[:, cols])
    y_abs = torch.zeros_like(y_ref).index_add_(1, rows, (Vs * Vm[:, cols]).abs())
    ey = ((y[: (B - 1) * cstride].view(B - 1, cstride)[:, :m] - y_ref).abs() / y_abs.clamp_min(1e-300)).max().item()
    del y_ref, y_abs
    g_ref = torch.zeros(B - 1, n, dtype=torch.float64, device="cuda").index_add_(1, cols, Vs * Lm[:, rows])
    g_abs = torch.zeros_like(g_ref).index_add_(1, cols, (Vs * Lm[:, rows]).abs())
    eg = ((g.view(B, -1)[: B - 1, :n] - g_ref).abs() / g_abs.clamp_min(1e-300)).max().item()
    print(f"full size: J v vs stored J: {ey:.3e}, J' lam vs stored J: {eg:.3e} (relative to sum |J||v|)")
    assert ey <= 1e-12 and eg <= 1e-12


def test_full_size_region_placed_buffer_holds_the_same_values():
    """B = 65 536, N = 40: the Jacobian buffer assembled by qln_vals_alloc_placed (physical chunks from different
    32-GiB regions of device memory behind one virtual range) receives bitwise what a plain allocation receives."""
    import torch

    batch, nlp, Z = _setup(65536, 40, False, seed=4)
    c = nlp.new_c()
    vals, ms = nlp.new_vals_regions(Z, c)
    vals.zero_()
    nlp.eval_c_and_jac(Z, c, vals, write_constants=True)
    c2, v2 = nlp.eval_c_and_jac(Z, write_constants=True)
    torch.cuda.synchronize()
    print(f"region-placed buffer: fused launch {ms:.3f} ms at setup")
    assert torch.equal(vals, v2) and torch.equal(c, c2)
    assert 0.5 < ms < 2.0


@pytest.mark.parametrize("B,N,ragged", [(65536, 40, False), (16384, 80, True)])
def test_gpu_jacobian_is_the_derivative_of_the_gpu_constraints(B, N, ragged):
    """A check of the Jacobian values that does not pass through the oracle, at full size: central differences of the GPU
    eval_c (whose values are bit-identical to the oracle's, which the notebook's printed numbers pin) along a random
    direction against (a) the stored step blocks of the hot kernel times the direction and (b) qln_eval_constraint_jvp, for
    every problem.  The one row where the reference's Jacobian is NOT the derivative -- the clock row of the transition
    knot, zeroed by jump*_jacobian (quirk Q1, src/planar_quadruped.jl:262-263) -- is checked to be exactly zero instead."""
    import torch

    batch, nlp, Z = _setup(B, N, ragged, seed=5)
    dev = Z.device
    gen = torch.Generator(device=dev).manual_seed(3)
    v = (torch.rand(nlp.dims.z_total, dtype=torch.float64, device=dev, generator=gen) * 2 - 1)
    eps = 1e-6
    fd = (nlp.eval_c(Z + eps * v) - nlp.eval_c(Z - eps * v)) / (2 * eps)
    c, vals = nlp.eval_c_and_jac(Z)
    jv = nlp.jac_vec(Z, v)
    torch.cuda.synchronize()
    c_off = torch.as_tensor(nlp.c_off, device=dev)
    j_off = torch.as_tensor(nlp.j_off, device=dev)
    kt = torch.as_tensor(batch.k_trans.astype(np.int64), device=dev)
    # (a) dynamics rows from the stored blocks: block_k (15 x 20, column-major) . v[x_k, u_k] - v[x_{k+1}]
    vz = v.view(B, -1)[:, : 20 * N - 5]
    vk = torch.cat([vz, vz.new_zeros(B, 5)], 1).view(B, N, 20)
    worst_a = 0.0
    for lo in range(0, B, 8192):
        hi = min(lo + 8192, B)
        idx = j_off[lo:hi, None] + torch.arange(300 * (N - 1), device=dev)[None, :]
        blocks = vals[idx].view(hi - lo, N - 1, 20, 15)
        bv = torch.einsum("bkcr,bkc->bkr", blocks, vk[lo:hi, :-1]) - vk[lo:hi, 1:, :15]
        ridx = c_off[lo:hi, None] + 29 + torch.arange(15 * (N - 1), device=dev)[None, :]
        f = fd[ridx].view(hi - lo, N - 1, 15)
        j = jv[ridx].view(hi - lo, N - 1, 15)
        q1 = torch.zeros(hi - lo, N - 1, 15, dtype=torch.bool, device=dev)
        kq = kt[lo:hi] - 2  # 0-based knot of the jump
        okq = (kq >= 0) & (kq < N - 1)
        q1[torch.nonzero(okq)[:, 0], kq[okq], 14] = True
        # Q1: the block's row is all zeros, what is left of the row is the -1 of the -I block
        bq = torch.nonzero(okq)[:, 0]
        assert float(blocks[bq, kq[okq], :, 14].abs().max()) == 0.0
        assert torch.equal(bv[q1], -vk[lo:hi, 1:, 14][q1[:, :, 14]]) and torch.equal(j[q1], bv[q1])
        scale = 1.0 + torch.einsum("bkcr,bkc->bkr", blocks.abs(), vk[lo:hi, :-1].abs())
        worst_a = max(worst_a, float(((bv - f).abs() / scale)[~q1].max()), float(((j - f).abs() / scale)[~q1].max()))
    # (b) every other row of c through the product kernel
    m = torch.as_tensor(np.array([nlp.problem_dims(0)[0]] * B if not ragged else [nlp.problem_dims(int(b))[0] for b in range(B)]), device=dev)
    worst_b = 0.0
    for name, start, length in (("init", 0, 15), ("term", 15, 14)):
        ridx = c_off[:, None] + start + torch.arange(length, device=dev)[None, :]
        worst_b = max(worst_b, float((jv[ridx] - fd[ridx]).abs().max()))
    tail0 = 29 + 15 * (N - 1)  # contact rows, final control, clearance: to the end of the problem's c
    maxlen = int((m - tail0).max())
    ar = torch.arange(maxlen, device=dev)[None, :]
    ridx = c_off[:, None] + tail0 + ar
    live = ar < (m - tail0)[:, None]
    # the clearance rows (the last N of a problem) are kinked at theta = 0 (|sin theta|): a knot whose theta lies within the
    # difference step of the kink has no derivative to compare with (a few dozen of the 2.6 million knots)
    theta = torch.cat([Z.view(B, -1)[:, : 20 * N - 5], Z.new_zeros(B, 5)], 1).view(B, N, 20)[:, :, 2]
    clr = ar - (m - tail0 - N)[:, None]  # knot of a clearance row, negative before the group
    near_kink = (clr >= 0) & live & (theta.gather(1, clr.clamp(0, N - 1)).abs() <= 2 * eps)
    print(f"clearance rows skipped at the kink: {int(near_kink.sum())}")
    live = live & ~near_kink
    ridx = torch.where(live, ridx, c_off[:, None])
    worst_b = max(worst_b, float(((jv[ridx] - fd[ridx]).abs() * live).max()))
    print(f"B={B} N={N} ragged={ragged}: |J v - central difference| / (1 + |J||v|): step blocks {worst_a:.2e}; other rows (absolute) {worst_b:.2e}")
    assert worst_a <= 1e-7 and worst_b <= 1e-7


def test_translation_and_clock_shift_invariance_at_full_size():
    """Two more properties of the model (src/planar_quadruped.jl:36-185) that need no reference data: nothing but the
    differences x_foot - xb enters the dynamics, and the clock state enters only its own row.  Shifting every horizontal
    position of BASELINE.json configs[2]'s batch by 0.25 m, and every clock by 1 s, must leave the dynamics residuals, the
    contact / final-control / clearance rows and every Jacobian value unchanged (to rounding: the shifted sums round
    differently), and move only the initial / terminal rows of the shifted entries -- by exactly the shift."""
    import torch

    B, N = 65536, 40
    batch, nlp, Z = _setup(B, N, False, seed=9)
    c0, v0 = nlp.eval_c_and_jac(Z)
    c0, v0 = c0.clone(), v0.clone()
    dx, dt = 0.25, 1.0
    shift = torch.zeros(20, dtype=torch.float64, device=Z.device)
    shift[[0, 3, 5]] = dx   # xb, x1, x2
    shift[14] = dt          # the clock
    Zs = Z.clone().view(B, -1)
    n_nlp = 20 * N - 5
    Zs[:, :n_nlp] += shift.repeat(N)[:n_nlp]
    c1, v1 = nlp.eval_c_and_jac(Zs.view(-1))
    torch.cuda.synchronize()
    sc = int(nlp.c_off[1] - nlp.c_off[0])
    m = nlp.problem_dims(0)[0]
    a = c0.as_strided((B, m), (sc, 1))
    b = c1.as_strided((B, m), (sc, 1))
    d = (b - a)
    want = torch.zeros(m, dtype=torch.float64, device=Z.device)
    want[[0, 3, 5]] = dx
    want[14] = dt
    want[[15 + 0, 15 + 3, 15 + 5]] = dx   # terminal rows x_N[1:14] - xf (the clock is not among them)
    err = (d - want).abs().max().item()
    scale = a.abs().max().item()
    jerr = ((v1 - v0).abs().max() / v0.abs().max()).item()
    print(f"rows: max |c(shifted) - c - expected shift| = {err:.2e} (largest |c| {scale:.1f}); Jacobian values: max relative change {jerr:.2e}")
    assert err <= 1e-12 * max(1.0, scale)
    assert jerr <= 1e-13


@pytest.mark.parametrize("B,N,ragged", [(65536, 40, False), (65536, 80, True)])
def test_every_problem_of_the_full_size_configs_against_the_oracle(B, N, ragged):
    """BASELINE.json configs[2] and configs[3] at FULL size, EVERY problem against the oracle (the C restatement evaluates a
    whole batch with OpenMP over the problems in a few seconds): NaN padding untouched; the equality rows of c, the objective
    and the gradient bit for bit; the clearance rows within 1 ulp (device sin); the exact-zero pattern of the Jacobian values
    identical; Jacobian entries within the north star's 1e-8 (closed form against dual numbers: ~4e-11).  Compared in slices
    of 8192 problems so that the host never holds more than one slice of the oracle's 6.7 / 14 GB Jacobian."""
    import os
    import torch
    from oracle import oracle as O

    batch, nlp, Z = _setup(B, N, ragged, seed=2)
    nan = float("nan")
    c = torch.full((nlp.dims.c_total,), nan, dtype=torch.float64, device="cuda")
    v = torch.full((nlp.dims.j_total,), nan, dtype=torch.float64, device="cuda")
    nlp.eval_c_and_jac(Z, c, v, write_constants=True)
    f, g = nlp.eval_f(Z), nlp.grad_f(Z)
    torch.cuda.synchronize()
    fh, gh = f.cpu().numpy(), g.cpu().numpy().reshape(B, -1)
    nthreads = min(os.cpu_count() or 1, 16)
    c_off, j_off = np.asarray(nlp.c_off), np.asarray(nlp.j_off)
    c_end = np.append(c_off[1:], nlp.dims.c_total)
    j_end = np.append(j_off[1:], nlp.dims.j_total)
    worst_j, worst_cl, n_eq, n_cl = 0.0, 0.0, 0, 0
    step = 8192
    for lo in range(0, B, step):
        hi = min(lo + step, B)
        c0, c1, j0, j1 = int(c_off[lo]), int(c_end[hi - 1]), int(j_off[lo]), int(j_end[hi - 1])
        Zs = np.zeros((hi - lo, nlp.z_stride))
        Zs[:, : nlp.n_nlp] = batch.Z[lo:hi]
        obj = batch.obj if batch.obj.ndim == 2 else batch.obj[lo:hi]
        ref = O.batch_eval(N, oracle_model(batch.model), batch.k_trans[lo:hi], batch.init_mode[lo:hi], batch.x0[lo:hi], batch.xf[lo:hi],
                           obj, Zs.reshape(-1), nlp.z_stride, c_off[lo:hi] - c0, j_off[lo:hi] - j0, c1 - c0, j1 - j0,
                           True, True, True, True, nthreads)
        cg, vg = c[c0:c1].cpu().numpy(), v[j0:j1].cpu().numpy()
        rc, rv = ref["c"], ref["vals"]
        # padding: NaN exactly where the oracle leaves its buffer untouched
        assert np.array_equal(np.isnan(cg), np.isnan(rc)) and np.array_equal(np.isnan(vg), np.isnan(rv)), lo
        # objective and gradient: bit for bit
        assert np.array_equal(fh[lo:hi], ref["f"]), lo
        assert np.array_equal(gh[lo:hi, : nlp.n_nlp], ref["grad"].reshape(hi - lo, -1)[:, : nlp.n_nlp]), lo
        # c: where the bits differ at all it is a clearance row (the last N rows of a problem), and then by <= 1 ulp
        diff = np.flatnonzero((cg != rc) & ~np.isnan(rc))
        if diff.size:
            owner = np.searchsorted(c_off[lo:hi] - c0, diff, side="right") - 1
            m = 18 * N - batch.k_trans[lo:hi][owner].astype(np.int64) + 16
            row = diff - (c_off[lo:hi] - c0)[owner]
            assert np.all((row >= m - N) & (row < m)), f"an equality row differs from the oracle (slice at {lo})"
            e = np.abs(cg[diff] - rc[diff]) / np.maximum(np.abs(rc[diff]), 0.25)
            worst_cl = max(worst_cl, float(e.max()))
            assert worst_cl <= 2.3e-16
        n_cl += int(diff.size)
        n_eq += int((~np.isnan(rc)).sum()) - int(diff.size)
        # Jacobian values: identical zero pattern, entries within RTOL
        ok = ~np.isnan(rv)
        assert np.array_equal(vg[ok] == 0, rv[ok] == 0), lo
        nzm = ok & (rv != 0)
        worst_j = max(worst_j, float((np.abs(vg[nzm] - rv[nzm]) / np.abs(rv[nzm])).max()))
        assert worst_j <= RTOL
        del ref, cg, vg, rc, rv, ok, nzm
    print(f"B={B} N={N}: every problem against the oracle -- {n_eq} entries of c bit-identical, {n_cl} clearance rows differ by <= "
          f"{worst_cl:.2e} relative, f and grad bit-identical, Jacobian entries within {worst_j:.2e}")


@pytest.mark.parametrize("fmt", ["dense_blocks", "structural"])
def test_full_size_one_launch_entry_points_give_the_bits_of_the_separate_ones(fmt):
    """B = 65 536, N = 40 (BASELINE.json configs[2]), both formats: qln_eval_all (f, grad, c, J from one read of Z) and
    qln_eval_objective_and_constraint (f, c) return, for every problem, the bits of the fused launch and of the separate
    objective / gradient kernels (which the test above holds to the oracle problem by problem)."""
    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    B, N = 65536, 40
    batch = PG.make_batch(B, N, 14, 1, seed=6)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, jac_format=fmt)
    Z = nlp.upload_Z(batch.Z)
    nan = float("nan")
    mk = lambda n: torch.full((n,), nan, dtype=torch.float64, device="cuda")
    zt, ct, jt = nlp.dims.z_total, nlp.dims.c_total, nlp.dims.j_total
    c0, v0 = nlp.eval_c_and_jac(Z, mk(ct), mk(jt))
    f0, g0 = nlp.eval_f(Z), nlp.grad_f(Z, mk(zt))
    f1, g1, c1, v1 = nlp.eval_all(Z, mk(B), mk(zt), mk(ct), mk(jt))
    torch.cuda.synchronize()
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert eq(c0, c1) and eq(v0, v1) and eq(f0, f1) and eq(g0, g1)
    del v1, g1, c1, f1
    f2, c2 = nlp.eval_f_and_c(Z, mk(B), mk(ct))
    torch.cuda.synchronize()
    assert eq(f0, f2) and eq(c0, c2)
    assert not torch.isnan(f0).any()
