"""Measurement aid: is the fast/slow placement of the Jacobian buffer a property of WHERE inside device memory it lies?
One large allocation, the fused launch timed on consecutive windows of it.  python bench/slab_windows.py [slab_GiB]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 64
batch, nlp, Z, c, vals = build("config3", 0, 0)
del vals
torch.cuda.empty_cache()
n = nlp.dims.j_total
slab = torch.empty(int(gib * 2**30 / 8), dtype=torch.float64, device="cuda")
step = (n + 262143) // 262144 * 262144  # windows start on 2-MiB boundaries
out = []
for w in range(slab.numel() // step):
    v = slab[w * step : w * step + n]
    nlp.init_jacobian_constants(v)
    ms = nlp.time_c_and_jac(Z, c, v, warmup=1, iters=3)
    out.append(float(np.median(ms)))
print(f"slab {gib:g} GiB at {slab.data_ptr():#x}: fused launch per {step * 8 / 2**30:.2f}-GiB window:", " ".join(f"{t:.3f}" for t in out))
# and the same number of separate allocations, for comparison
c2 = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(6)]
o2 = []
for v in c2:
    nlp.init_jacobian_constants(v)
    o2.append(float(np.median(nlp.time_c_and_jac(Z, c, v, warmup=1, iters=3))))
print("separate allocations:", " ".join(f"{t:.3f}" for t in o2))
