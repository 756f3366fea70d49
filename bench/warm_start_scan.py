"""qln_solve warm-started from the reference's own solved trajectories (tests/golden/data_*.csv) under different
penalty schedules: where does it end relative to the file?  Development aid for tests/test_gpu_solve.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quadruped_landing_amd import HybridNLP, problem_gen as PG, nlp as NL

for i in (6, 3, 4, 5, 1, 2):
    Zf = np.loadtxt(os.path.join(ROOT, "tests", "golden", f"data_{i}.csv"))
    nb = PG.notebook_problem()
    if i != 6:
        nb.x0[0] = Zf[:15]
    nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
    Xf = np.concatenate([Zf, np.zeros(5)]).reshape(61, 20)
    for kw in (dict(), dict(rho0=1e2), dict(rho0=1e4), dict(rho0=1e6), dict(rho0=1e4, max_inner=30), dict(rho0=1e6, max_inner=30),
               dict(rho0=1e6, rho_max=1e9, max_inner=60), dict(rho0=1e4, exact_h_gradient=1, max_inner=30)):
        Z, info = nlp.solve(nlp.upload_Z(Zf[None, :]), **kw)
        torch.cuda.synchronize()
        inf = info.cpu().numpy()[0]
        viol = float(nlp.constraint_violation(nlp.eval_c(Z)).cpu()[0])
        f = float(nlp.eval_f(Z).cpu()[0])
        X = np.concatenate([Z.cpu().numpy()[:1215], np.zeros(5)]).reshape(61, 20)
        print(f"data_{i} {kw}: status {inf[5]:.0f} outer {inf[0]:.0f} iters {inf[1]:.0f} f {f:.6f} viol {viol:.3e} "
              f"dstate {np.abs(X[:, :14] - Xf[:, :14]).max():.3e} dforce {np.abs(X[:60, 15:19] - Xf[:60, 15:19]).max():.3e} "
              f"dh {np.abs(X[:60, 19] - Xf[:60, 19]).max():.2e}", flush=True)
