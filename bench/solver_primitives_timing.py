"""Measurement aid: launch times of the caller-side primitives (Jacobian products, Gauss-Newton step) on a bench
workload.  python bench/solver_primitives_timing.py [config3|config4|config2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

def t_ms(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
batch, nlp, Z, c, vals = build(wl, 0, 0)
del vals
nlp.eval_c(Z, c)
v = torch.randn(nlp.dims.z_total, dtype=torch.float64, device="cuda")
lam = torch.randn(nlp.dims.c_total, dtype=torch.float64, device="cuda")
y, g, dZ = nlp.new_c(), nlp.new_Z(), nlp.new_Z()
knots = batch.B * batch.N
zb, cb = 8 * nlp.n_nlp * batch.B, 8 * float(np.sum(18 * batch.N - batch.k_trans + 16))
t = t_ms(lambda: nlp.jac_vec(Z, v, y)); print(f"{wl} J v            : {t:.3f} ms  ({(2 * zb + cb) / t / 1e6:.0f} GB/s of Z + v + y)")
t = t_ms(lambda: nlp.jac_t_vec(Z, lam, g)); print(f"{wl} J' lam         : {t:.3f} ms  ({(2 * zb + cb) / t / 1e6:.0f} GB/s of Z + lam + g)")
info = torch.zeros(8 * batch.B, dtype=torch.float64, device="cuda")
for iters in (0, 10, 50, 200):
    t = t_ms(lambda: nlp.gauss_newton_step(Z, c, dZ, max_iters=iters, rel_tol=0.0, info=info), iters=5)
    done = info.view(batch.B, 8)[:, 0].mean().item()
    print(f"{wl} Gauss-Newton step, {iters:3d} CGLS iterations: {t:8.3f} ms  ({done:.0f} done; "
          f"{t * 1e3 / max(iters, 1) / 1:.1f} us per CGLS iteration of the whole batch, {batch.B * max(iters,1) / t / 1e3:.2f} M problem-iterations/s)")
