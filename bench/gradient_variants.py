"""grad_f at config 3 through the tuning build's variants of k_objective_gradient_shared (QLN_GRAD_VARIANT / QLN_GRAD_PER_CU
are read once per process: every variant runs in a child process): slices of Z in flight per wave, waves per SIMD the
registers are sized for, persistent waves per CU.  HIP events, median of 30.
   python bench/gradient_variants.py            # on the GPU box; needs `make -C quadruped_landing_amd/csrc tuning`"""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from bench import build
batch, nlp, Z, c, vals = build("config3", 0, 0, placement_trials=0)
g = nlp.new_Z()
for _ in range(5): nlp.grad_f(Z, g)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); nlp.grad_f(Z, g); b.record()
torch.cuda.synchronize()
t = float(np.median([a.elapsed_time(b) for a, b in ev]))
byts = 2 * 8.0 * nlp.n_nlp * batch.B
print("%%.4f ms  %%5.1f %%%% of 8 TB/s  checksum %%.17g" %% (t, byts / t / 1e6 / 80, float(g.sum())))
''' % ROOT

NAMES = {0: "product (1 slice in flight, registers for 2 waves/SIMD)", 1: "2 slices in flight, 2 waves/SIMD", 2: "2 slices in flight, 3 waves/SIMD",
         3: "1 slice in flight, 3 waves/SIMD", 4: "3 slices in flight, 2 waves/SIMD", 5: "1 slice in flight, 4 waves/SIMD"}
PER_CU = {0: (8, 6, 4), 1: (8, 4), 2: (12, 8), 3: (12, 8), 4: (8, 4), 5: (16, 12, 8)}
tuning = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_tuning.so")
prev = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_prev.so")


def run(label, lib, env_extra):
    env = dict(os.environ, QLN_LIB_PATH=lib, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]
    print(f"{label:72s} {out}", flush=True)


if os.path.exists(prev):
    run("previous build (table in LDS, four waves per workgroup)", prev, {})
for var in sorted(NAMES):
    for pc in PER_CU[var]:
        run(f"variant {var}: {NAMES[var]}, {pc} waves/CU", tuning, {"QLN_GRAD_VARIANT": str(var), "QLN_GRAD_PER_CU": str(pc)})
if os.path.exists(prev):
    run("previous build (again)", prev, {})
