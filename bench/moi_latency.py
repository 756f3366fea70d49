"""Measurement aid: wall-clock latency of the host-pointer (MOI-mode) callbacks for ONE problem (the notebook's
N=61), i.e. what Ipopt would see per callback through the drop-in boundary."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadruped_landing_amd import HybridNLP, moi, problem_gen as PG

nb = PG.notebook_problem()
nlp = HybridNLP(nb.model, nb.obj, nb.init_mode, nb.k_trans, nb.N, nb.x0, nb.xf)
Z = nb.Z[0]
m, n = nlp.num_duals(), nlp.num_primals()
g = np.zeros(m); gr = np.zeros(n); dense = np.zeros(m * n); _, nnz = nlp.problem_dims(0)
def t(fn, reps=200):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print("eval_objective            : %7.1f us" % t(lambda: moi.eval_objective(nlp, Z)))
print("eval_objective_gradient   : %7.1f us" % t(lambda: moi.eval_objective_gradient(nlp, gr, Z)))
print("eval_constraint           : %7.1f us" % t(lambda: moi.eval_constraint(nlp, g, Z)))
print("eval_constraint_jacobian (dense %dx%d, write-set scatter on host): %7.1f us" % (m, n, t(lambda: moi.eval_constraint_jacobian(nlp, dense, Z), 50)))
print("jac_c_host (block-COO, %d values)                               : %7.1f us" % (nnz, t(lambda: nlp.jac_c_host(Z), 100)))
