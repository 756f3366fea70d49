"""Measurement aid: the fused launch timed on windows of one large allocation at 2-GiB steps (where does the fast class
live?), then the re-allocation trick: free the slab, allocate a spacer of the best window's offset followed by the
buffer, and time that.   python bench/placement_scan.py [slab_GiB] [step_GiB]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 80
step_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2
batch, nlp, Z, c, vals = build("config3", 0, 0)
del vals
torch.cuda.empty_cache()
n = nlp.dims.j_total
slab = torch.empty(int(gib * 2**30), dtype=torch.uint8, device="cuda")
print(f"slab at {slab.data_ptr():#x}")
step = int(step_gib * 2**30)
times = []
for off in range(0, slab.numel() - 8 * n, step):
    v = slab[off : off + 8 * n].view(torch.float64)
    ms = nlp.time_c_and_jac(Z, c, v, warmup=1, iters=2)
    times.append((off, float(np.min(ms))))
print("offset GiB : ms   " + "  ".join(f"{o / 2**30:.0f}:{t:.3f}" for o, t in times))
best_off, best_t = min(times, key=lambda x: x[1])
print(f"best window at {best_off / 2**30:.1f} GiB: {best_t:.3f} ms")
del slab, v
torch.cuda.empty_cache()
spacer = torch.empty(best_off, dtype=torch.uint8, device="cuda") if best_off else None
v2 = nlp.new_vals()
t2 = float(np.min(nlp.time_c_and_jac(Z, c, v2, warmup=1, iters=3)))
print(f"re-allocated behind a {best_off / 2**30:.1f}-GiB spacer: {t2:.3f} ms (spacer at {spacer.data_ptr() if spacer is not None else 0:#x}, vals at {v2.data_ptr():#x})")
del spacer
torch.cuda.empty_cache()
t3 = float(np.min(nlp.time_c_and_jac(Z, c, v2, warmup=1, iters=3)))
print(f"after freeing the spacer: {t3:.3f} ms")
