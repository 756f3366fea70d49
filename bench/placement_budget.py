"""What a smaller transient budget for qln_vals_alloc_placed costs (measurement aid): config 3's Jacobian buffer placed with
64 GiB (the default: two region boundaries inside the scanned slab), 32 GiB (one), 8 GiB, 0 (no scan: a plain mapping), the
fused launch timed on each; two rounds, interleaved, one box.
    python bench/placement_budget.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import build  # noqa: E402

batch, nlp, Z, c, vals = build("config3", 0, 0, placement_trials=1)
del vals
torch.cuda.empty_cache()
for rnd in range(2):
    for gib in (64.0, 32.0, 8.0, 0.0):
        t0 = time.perf_counter()
        v, ms_scan = nlp.new_vals_regions(Z, c, transient_gib=gib)
        dt = time.perf_counter() - t0
        nlp.init_jacobian_constants(v)
        ms = nlp.time_c_and_jac(Z, c, v, warmup=3, iters=20)
        chunk, scanned, first = nlp.placed_info(v)
        print(f"transient budget {gib:5.1f} GiB: scan kept window starting at chunk {first:3d} of {scanned:3d} ({chunk >> 20} MiB chunks), "
              f"set-up {dt:5.2f} s; fused launch {float(np.mean(ms)):.4f} ms avg, {float(np.min(ms)):.4f} min", flush=True)
        del v
        torch.cuda.synchronize()
retired, cap = nlp.placed_address_space()
print(f"virtual address space retired by these {2 * 4} placed allocations: {retired / 2**30:.1f} GiB of a cap of {cap / 2**40:.0f} TiB")
