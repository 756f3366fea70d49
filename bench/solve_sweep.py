"""Parameter sweep / timing of qln_solve on random landing problems (run on the GPU box).
   python bench/solve_sweep.py [B] [N] [k_trans] [full]     ("full": the option sets of profiles/r02_solve_sweep.txt as well)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from quadruped_landing_amd import HybridNLP, problem_gen as PG

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kt = int(sys.argv[3]) if len(sys.argv) > 3 else 14
batch = PG.make_batch(B, N, kt, 1, seed=3, noise=0.0)
nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
Z0 = nlp.initial_guess()
torch.cuda.synchronize()
sets = [dict()]
if "full" in sys.argv:
    sets += [dict(max_outer=30, max_inner=60, rho0=1.0, rho_factor=10.0), dict(h_prox=0.0),
             dict(exact_h_gradient=1, max_inner=30, max_outer=40)]
for opts in sets:
    Z = Z0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Z, info = nlp.solve(Z, **opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    inf = info.cpu().numpy()
    c = nlp.eval_c(Z)
    viol = nlp.constraint_violation(c).cpu().numpy()
    f = nlp.eval_f(Z).cpu().numpy()
    st = np.bincount(inf[:, 5].astype(int), minlength=3)
    print(f"{str(opts):70s} status {st}  iters med {np.median(inf[:,1]):5.0f} max {inf[:,1].max():5.0f}  viol max {viol.max():.2e} "
          f"p99 {np.quantile(viol, 0.99):.2e}  f med {np.median(f):8.3f}  {dt*1e3:8.1f} ms  {B/dt:9.0f} problems/s", flush=True)
Z, info = nlp.solve(Z0.clone())
torch.cuda.synchronize()
inf = info.cpu().numpy()
bad = np.nonzero(inf[:, 5] != 0)[0]
print("not converged:", bad[:40].tolist())
for b in bad[:12]:
    print(b, "outer %d iters %d viol %.2e rho %.0e status %d J %.6f alpha %.3g mu %.1e x0: th %.3f y2 %.3f vby %.3f w %.3f" % (
        inf[b,0], inf[b,1], inf[b,3], inf[b,4], inf[b,5], inf[b,6], inf[b,7], inf[b,9], batch.x0[b,2], batch.x0[b,6], batch.x0[b,8], batch.x0[b,9]))
tk = inf[:, 10:15]
print("phase ticks per iLQR iteration (median over problems): refresh %.0f  blocks %.0f  sweep %.0f  rollouts %.0f  accept %.0f  (s_memtime ticks)" % tuple(np.median(tk / inf[:, 1:2], axis=0)))
print("share: ", np.round(np.median(tk / tk.sum(axis=1, keepdims=True), axis=0), 3))
