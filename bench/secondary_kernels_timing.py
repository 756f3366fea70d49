"""Launch times of the kernels an NLP iteration calls beside the hot one (HIP events, median of 20), on a bench
workload: eval_f, grad_f, eval_c alone, the fused c+J, the single-launch f+grad+c+J, J v, J' lam.
   python bench/secondary_kernels_timing.py [config3|config4|config2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

def t_ms(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
batch, nlp, Z, c, vals = build(wl, 0, 0, placement_trials=int(os.environ.get("PLACEMENT_TRIALS", "8")))
f, g = nlp.new_f(), nlp.new_Z()
v = torch.randn(nlp.dims.z_total, dtype=torch.float64, device="cuda")
lam = torch.randn(nlp.dims.c_total, dtype=torch.float64, device="cuda")
y, gz = nlp.new_c(), nlp.new_Z()
zb = 8.0 * nlp.n_nlp * batch.B
cb = 8.0 * float(np.sum(18 * batch.N - batch.k_trans + 16))
jb = 8.0 * (300 * (batch.N - 1) + batch.N) * batch.B
costb = 8.0 * 41 * batch.N * (batch.B if nlp.cost_batch > 1 else 1)
rows = [("eval_f            (k_objective)", lambda: nlp.eval_f(Z, f), zb + costb),
        ("grad_f            (k_objective_gradient)", lambda: nlp.grad_f(Z, g), 2 * zb + costb),
        ("eval_c            (k_constraint_jacobian, c only)", lambda: nlp.eval_c(Z, c), zb + cb),
        ("eval_c + jac_c    (k_constraint_jacobian, the hot launch)", lambda: nlp.eval_c_and_jac(Z, c, vals, write_constants=False), zb + cb + jb),
        ("f + grad + c + J  (qln_eval_all, one launch)", lambda: nlp.eval_all(Z, f, g, c, vals, write_constants=False), 2 * zb + cb + jb + costb),
        ("J v               (k_constraint_jvp)", lambda: nlp.jac_vec(Z, v, y), 2 * zb + cb),
        ("J' lam            (k_constraint_vjp)", lambda: nlp.jac_t_vec(Z, lam, gz), 2 * zb + cb)]
tot = {}
for name, fn, byts in rows:
    t = t_ms(fn)
    tot[name[:6]] = t
    print(f"{wl} {name:62s} {t:7.3f} ms   {byts / 1e9:6.3f} GB compulsory -> {byts / t / 1e6:6.0f} GB/s = {byts / t / 1e6 / 80:4.1f} % of 8 TB/s")
