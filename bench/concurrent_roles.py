"""Would splitting the fused structural launch into a constraint role and a Jacobian role that run CONCURRENTLY pay?  Emulated with
two launches on two streams (eval_c! alone || jac_c! alone, structural format): Z is then read twice from HBM, which a one-launch
role split would avoid -- so this is a pessimistic estimate.  Measurement aid.
    python bench/concurrent_roles.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import build  # noqa: E402

batch, nlp, Z, c, vals = build("config3", 0, 0, placement_trials=8 if os.environ.get("QLN_ABLATE_PLACED") else 1, jac_format="structural")
from quadruped_landing_amd import HybridNLP  # noqa: E402

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x0, xf = nlp.boundary_states()
nlp_c = HybridNLP(batch.model, None, batch.init_mode, batch.k_trans, batch.N, x0, xf, jac_format="structural", stream=sa)
nlp_j = HybridNLP(batch.model, None, batch.init_mode, batch.k_trans, batch.N, x0, xf, jac_format="structural", stream=sb)
torch.cuda.synchronize()


def timed(fn, iters=20):
    out = []
    for i in range(iters + 3):
        torch.cuda.synchronize()
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        sa.wait_event(e0)
        sb.wait_event(e0)
        fn()
        ea.record(sa)
        eb.record(sb)
        torch.cuda.synchronize()
        if i >= 3:
            out.append(max(e0.elapsed_time(ea), e0.elapsed_time(eb)))
    return float(np.median(out))


def both():
    nlp_c.eval_c(Z, c)
    nlp_j.jac_c(Z, vals, write_constants=False)


def fused_on_a():
    nlp_c.eval_c_and_jac(Z, c, vals, write_constants=False)


def c_then_j_on_a():
    nlp_c.eval_c(Z, c)
    nlp_c.jac_c(Z, vals, write_constants=False)


for rnd in range(3):
    print(f"fused launch {timed(fused_on_a):.4f} ms | eval_c || jac_c on two streams {timed(both):.4f} ms | eval_c then jac_c on one stream "
          f"{timed(c_then_j_on_a):.4f} ms", flush=True)
