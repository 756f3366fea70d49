"""The reference's notebook, start to finish, without Ipopt: build the landing problem of src/main.ipynb, take its
initial guess Z0, solve the NLP on the GPU (qln_solve), judge the result with the evaluator, and write the trajectory in
the reference's own file format (one float per line, src/main.ipynb:881) so that its plot scripts read it unchanged.

    python examples/solve_on_gpu.py [out.csv]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from quadruped_landing_amd import HybridNLP, problem_gen as PG, trajectory_io  # noqa: E402

nb = PG.notebook_problem()                       # N = 61, k_trans = 21, init_mode = 1 (src/main.ipynb:92-161)
nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
Z = nlp.initial_guess()                          # Z0 = packZ(nlp, Xguess, Uref), built on the device
Z, info = nlp.solve(Z)                           # solve(Z0, nlp) of src/moi.jl:46, on the GPU
c = nlp.eval_c(Z)
viol = float(nlp.constraint_violation(c)[0])     # what Ipopt prints as "Constraint violation"
f = float(nlp.eval_f(Z)[0])
torch.cuda.synchronize()
i = info.cpu().numpy()[0]
print(f"status {i[5]:.0f} after {i[0]:.0f} multiplier updates / {i[1]:.0f} iLQR iterations")
print(f"objective            {f:.10f}   (the reference's Ipopt run: 116.08112892558562, 'Restoration Failed')")
print(f"constraint violation {viol:.3e}      (the reference's run:        1.4928675395736724e-06)")
out = sys.argv[1] if len(sys.argv) > 1 else "data_gpu.csv"
trajectory_io.save_trajectory(out, Z.cpu().numpy()[: nlp.n_nlp])
print(f"trajectory written to {out} in the reference's format ({nlp.n_nlp} lines)")
