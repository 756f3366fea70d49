"""eval_f at config 3 through the tuning build's variants of k_objective_shared (QLN_OBJ_VARIANT / QLN_OBJ_PER_CU are read
once per process, so every variant runs in a child process): where the in-order sum reads its terms, slices of Z in flight
per wave, waves per SIMD, persistent waves per CU.  HIP events, median of 30.
   python bench/objective_variants.py            # on the GPU box; needs `make -C quadruped_landing_amd/csrc tuning`"""
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
f = nlp.new_f()
for _ in range(5): nlp.eval_f(Z, f)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); nlp.eval_f(Z, f); b.record()
torch.cuda.synchronize()
t = float(np.median([a.elapsed_time(b) for a, b in ev]))
byts = 8.0 * nlp.n_nlp * batch.B
print("%%.4f ms  %%5.1f %%%% of 8 TB/s  checksum %%.17g" %% (t, byts / t / 1e6 / 80, float(f.sum())))
''' % ROOT

NAMES = {0: "product (lane sum, 2 slices in flight, 2 waves/SIMD)", 1: "LDS sum, 2 slices in flight, 2 waves/SIMD",
         2: "lane sum, 1 slice in flight, 2 waves/SIMD", 3: "lane sum, 1 slice in flight, 3 waves/SIMD (spills)",
         4: "LDS sum, 1 slice in flight, 3 waves/SIMD (spills)", 5: "lane sum, 3 slices in flight, 2 waves/SIMD"}
tuning = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_tuning.so")
prev = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_prev.so")


def run(label, lib, env_extra):
    env = dict(os.environ, QLN_LIB_PATH=lib, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]
    print(f"{label:64s} {out}", flush=True)


if os.path.exists(prev):
    run("previous build (table + sum in LDS, 8 waves per workgroup)", prev, {})
for var in sorted(NAMES):
    for pc in ((12, 8) if var in (3, 4) else (8, 6, 4)):
        run(f"variant {var}: {NAMES[var]}, {pc} waves/CU", tuning, {"QLN_OBJ_VARIANT": str(var), "QLN_OBJ_PER_CU": str(pc)})
if os.path.exists(prev):
    run("previous build (again)", prev, {})
