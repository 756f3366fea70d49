"""Scan of qln_solve's penalty schedule on random landing problems (run on the GPU box): solved count, iterations, time.
   python bench/solve_schedule_scan.py [B]"""
import sys, time, itertools
import numpy as np
import torch
sys.path.insert(0, ".")
from quadruped_landing_amd import HybridNLP, problem_gen as PG

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = PG.make_batch(B, 40, 14, 1, seed=3, noise=0.0)
nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
Z0 = nlp.initial_guess()
nlp.solve(Z0.clone())
torch.cuda.synchronize()
rows = []
for mi, r0, rf in itertools.product((4, 6, 8, 12), (3.0, 10.0, 30.0, 100.0), (3.0, 5.0, 10.0)):
    Z = Z0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Z, info = nlp.solve(Z, max_inner=mi, rho0=r0, rho_factor=rf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    inf = info.cpu().numpy()
    viol = nlp.constraint_violation(nlp.eval_c(Z)).cpu().numpy()
    f = nlp.eval_f(Z).cpu().numpy()
    ok = int((viol <= 1.0001e-6).sum())
    rows.append((dt, mi, r0, rf, ok, np.median(inf[:, 1]), inf[:, 1].max(), np.median(f)))
    print(f"max_inner {mi:2d} rho0 {r0:5.0f} rho_factor {rf:4.0f}: solved {ok:5d}/{B}  iters med {np.median(inf[:,1]):4.0f} max {inf[:,1].max():5.0f}  f med {np.median(f):.3f}  {dt*1e3:7.1f} ms", flush=True)
print("fastest with everything solved:")
for r in sorted(r for r in rows if r[4] == B)[:5]:
    print("  max_inner %d rho0 %.0f rho_factor %.0f: %.1f ms, iters med %.0f max %.0f, f med %.3f" % (r[1], r[2], r[3], r[0] * 1e3, r[5], r[6], r[7]))
