"""Measurement aid: which allocation does the launch time follow -- the handle's descriptor arrays or the big buffers?"""
import os, sys, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadruped_landing_amd import HybridNLP, problem_gen as PG

batch = PG.make_batch(65536, 40, 14, 1, seed=0)
def mk():
    return HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, stream=torch.cuda.current_stream())
def t(nlp, z, c, v):
    return float(np.median(nlp.time_c_and_jac(z, c, v, warmup=3, iters=20)))
print("A: buffers fixed, handle re-created")
nlp = mk(); z = nlp.upload_Z(batch.Z); c = nlp.new_c(); v = nlp.new_vals(); nlp.init_jacobian_constants(v)
for rep in range(6):
    print("  rep %d: %.3f ms" % (rep, t(nlp, z, c, v)))
    del nlp; gc.collect(); nlp = mk()
print("B: handle fixed, vals re-allocated (empty_cache in between)")
for rep in range(6):
    del v; gc.collect(); torch.cuda.empty_cache()
    v = nlp.new_vals(); nlp.init_jacobian_constants(v)
    print("  rep %d: %.3f ms  v=%x" % (rep, t(nlp, z, c, v), v.data_ptr()))
print("C: handle fixed, c re-allocated")
for rep in range(6):
    del c; gc.collect(); torch.cuda.empty_cache()
    c = nlp.new_c()
    print("  rep %d: %.3f ms  c=%x" % (rep, t(nlp, z, c, v), c.data_ptr()))
print("D: handle fixed, Z re-allocated")
for rep in range(6):
    del z; gc.collect(); torch.cuda.empty_cache()
    z = nlp.upload_Z(batch.Z)
    print("  rep %d: %.3f ms  z=%x" % (rep, t(nlp, z, c, v), z.data_ptr()))
print("E: nothing changes")
for rep in range(4):
    print("  rep %d: %.3f ms" % (rep, t(nlp, z, c, v)))
