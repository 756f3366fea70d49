"""Measurement aid: does the size of the allocation behind the Jacobian buffer decide its speed class?  A buddy-type
VRAM manager serves a 6.75-GB request from several smaller blocks but an 8-GiB request from one block.
python bench/alloc_size_class.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

batch, nlp, Z, c, vals = build("config3", 0, 0)
del vals
torch.cuda.empty_cache()
n = nlp.dims.j_total
held = []
for label, count in (("exact (6.29 GiB)", n), ("8 GiB", 2**30), ("16 GiB", 2**31), ("exact (6.29 GiB)", n), ("8 GiB", 2**30)):
    out = []
    for rep in range(3):
        t = torch.empty(count, dtype=torch.float64, device="cuda")
        v = t[:n]
        nlp.init_jacobian_constants(v)
        out.append(float(np.median(nlp.time_c_and_jac(Z, c, v, warmup=1, iters=3))))
        held.append(t)  # keep it, so that the next candidate is different memory
    print(f"{label:18s}", " ".join(f"{x:.3f}" for x in out), flush=True)
