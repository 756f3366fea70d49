import sys, time, numpy as np
sys.path.insert(0,'.')
import torch
from bench import build
batch, nlp, Z, c, vals = build("config3", 0, 0, placement_trials=8)
for rnd in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); ms=nlp.time_c_and_jac(Z,c,vals,0,50); torch.cuda.synchronize(); w1=(time.perf_counter()-t0)*1e3/50
    torch.cuda.synchronize(); t0=time.perf_counter(); tot=nlp.time_c_and_jac_total(Z,c,vals,0,50); torch.cuda.synchronize(); w2=(time.perf_counter()-t0)*1e3/50
    print(f"per-launch events: kernel avg {ms.mean():.4f} ms, wall per launch {w1:.4f} ms | one event pair: {tot/50:.4f} ms per launch, wall per launch {w2:.4f} ms")
