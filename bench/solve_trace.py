"""Per-outer-iteration trace of qln_solve for one problem of a random batch (GPU box): the solve is deterministic, so
running it with max_outer = 1, 2, ... shows the state after each multiplier update.
   python bench/solve_trace.py N kt seed index B [exact]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from quadruped_landing_amd import HybridNLP, problem_gen as PG
N, kt, seed, idx, B = (int(a) for a in sys.argv[1:6])
exact = int(sys.argv[6]) if len(sys.argv) > 6 else 0
batch = PG.make_batch(B, N, kt, 1, seed=seed, noise=0.0)
nlp = HybridNLP(batch.model, batch.obj, 1, kt, N, batch.x0[idx], batch.xf[idx])
for mo in list(range(0, 16)) + [20, 30]:
    Z, info = nlp.solve(nlp.initial_guess(), max_outer=mo, exact_h_gradient=exact)
    inf = info.cpu().numpy()[0]
    print("max_outer %2d: outer %2.0f iters %4.0f f %12.6f viol %.3e rho %.0e status %.0f J %.6f alpha %.3g sum h %.4f mu %.1e" % (
        mo, inf[0], inf[1], inf[2], inf[3], inf[4], inf[5], inf[6], inf[7], inf[8], inf[9]))
