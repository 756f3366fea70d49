"""Measurement aid: does it matter in which 32-GiB region of device memory the decision vector Z (read) and the
constraint vector c (written, 5 % of the traffic) lie relative to the region-placed Jacobian buffer?
python bench/z_placement.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

batch, nlp, Z, c, vals = build("config3", 0, 0, placement_trials=8)
print("vals at %#x  Z at %#x  c at %#x; fused launch %.3f ms" % (vals.data_ptr(), Z.data_ptr(), c.data_ptr(),
      float(np.median(nlp.time_c_and_jac(Z, c, vals, warmup=2, iters=10)))))
held = []
for i in range(7):
    held.append(torch.empty(30 * 2**30, dtype=torch.uint8, device="cuda"))   # move on by ~one region
    Z2 = Z.clone()
    c2 = torch.empty_like(c)
    tz = float(np.median(nlp.time_c_and_jac(Z2, c, vals, warmup=2, iters=10)))
    tc = float(np.median(nlp.time_c_and_jac(Z, c2, vals, warmup=2, iters=10)))
    tb = float(np.median(nlp.time_c_and_jac(Z2, c2, vals, warmup=2, iters=10)))
    print(f"copy {i}: Z at {Z2.data_ptr():#x}: {tz:.3f} ms   c at {c2.data_ptr():#x}: {tc:.3f} ms   both: {tb:.3f} ms", flush=True)
    held += [Z2, c2]
