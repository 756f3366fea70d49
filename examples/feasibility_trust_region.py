"""Demonstration (not a solver): a batched trust-region loop on the constraint violation built from the device
primitives -- qln_eval_constraint, qln_gauss_newton_step (CGLS in LDS with a Steihaug trust radius) -- for B landing
problems at once.  The reference hands its callbacks to Ipopt (src/moi.jl:46-103) and ends the notebook run with
"Restoration Failed" at violation 1.5e-6 after 428 iterations; this loop only shows the step primitive doing its
job in an outer iteration: the merit ||rho||^2 falls monotonically for every problem.  Variable bounds of solve()
(src/moi.jl:52-66) are not imposed here.

    python examples/feasibility_trust_region.py [B] [N] [steps] [cgls_iters]      (B = 0: the notebook problem, N = 61)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def default_col_scale(N):
    """Forces are O(1e2) N, time steps O(1e-2) s, everything else O(1): scale the unknowns accordingly."""
    d = np.ones(20 * N - 5)
    for k in range(N - 1):
        d[20 * k + 15 : 20 * k + 19] = 100.0
        d[20 * k + 19] = 0.01
    return d


def trust_region_feasibility(nlp, Z, steps=30, radius0=1.0, cgls_iters=300, col_scale=None, verbose=False):
    """Z is updated in place.  Returns the history of (max violation, max merit) over the batch, one entry per step."""
    import torch
    from quadruped_landing_amd._lib import GN_INFO_STRIDE

    B, stride = nlp.B, nlp.z_stride
    dev = Z.device
    dsc = torch.from_numpy(default_col_scale(nlp.N) if col_scale is None else np.asarray(col_scale, dtype=np.float64)).to(dev)
    radius = torch.full((B,), float(radius0), dtype=torch.float64, device=dev)
    c, ct = nlp.new_c(), nlp.new_c()
    dZ, Zt, scratch = nlp.new_Z(), nlp.new_Z(), nlp.new_Z()
    info = torch.zeros(B * GN_INFO_STRIDE, dtype=torch.float64, device=dev)
    info_t = torch.zeros_like(info)
    viol = nlp.new_f()
    hist = []
    for it in range(steps):
        nlp.eval_c(Z, c)
        nlp.constraint_violation(c, viol)
        nlp.gauss_newton_step(Z, c, dZ, max_iters=cgls_iters, rel_tol=1e-10, radius=radius, col_scale=dsc, info=info)
        I = info.view(B, GN_INFO_STRIDE)
        phi, pred = I[:, 4], I[:, 4] - I[:, 3]
        hist.append((viol.max().item(), phi.max().item()))
        torch.add(Z, dZ, out=Zt)
        nlp.eval_c(Zt, ct)
        nlp.gauss_newton_step(Zt, ct, scratch, max_iters=0, rel_tol=0.0, info=info_t)  # only ||rho(Zt)||^2 is wanted
        phi_t = info_t.view(B, GN_INFO_STRIDE)[:, 4]
        ratio = torch.where(pred > 0, (phi - phi_t) / pred, torch.full_like(pred, -1.0))
        accept = ratio > 0.05
        Z.view(B, stride)[accept] = Zt.view(B, stride)[accept]
        hit = I[:, 5] > 0
        radius = torch.where(ratio < 0.25, radius * 0.25, torch.where((ratio > 0.75) & hit, radius * 2.0, radius))
        if verbose:
            print(f"step {it:3d}: max violation {hist[-1][0]:.3e}  max merit {hist[-1][1]:.3e}  accepted {int(accept.sum())}/{B}")
    nlp.eval_c(Z, c)
    nlp.constraint_violation(c, viol)
    nlp.gauss_newton_step(Z, c, scratch, max_iters=0, rel_tol=0.0, info=info)
    hist.append((viol.max().item(), info.view(B, GN_INFO_STRIDE)[:, 4].max().item()))
    return hist


if __name__ == "__main__":
    import time

    import torch
    from quadruped_landing_amd import HybridNLP, problem_gen as PG

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    cgls_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    batch = PG.notebook_problem() if B == 0 else PG.make_batch(B, N, max(2, N // 3), 1, seed=0, noise=0.0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.initial_guess()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = trust_region_feasibility(nlp, Z, steps=steps, cgls_iters=cgls_iters, verbose=steps <= 40)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    viol = [v for v, _ in hist]
    below = next((i for i, v in enumerate(viol) if v <= 1.4928675395736724e-06), None)
    print(f"B={batch.B} N={batch.N}: {steps} trust-region steps (<= {cgls_iters} CGLS iterations each) in {dt * 1e3:.1f} ms "
          f"({dt / steps * 1e3:.2f} ms per step for the batch); max violation {viol[0]:.3e} -> {viol[-1]:.3e}; "
          f"first step with every problem at or below Ipopt's final violation on the notebook problem (1.49e-06): {below}")
    print("max violation every 10th step:", " ".join(f"{v:.1e}" for v in viol[::10]))
