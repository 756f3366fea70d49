"""The fused launch with an L2 prefetch of a LATER workgroup's slice of Z (k_constraint_jacobian, flags >> 8): launch time
against the prefetch distance (problems ahead on the XCD; 0 = off), structural and dense format, config 3 or 4.  The distance
is read once per process from QLN_PREFETCH_AHEAD by the `prefetchknob` build (make -C quadruped_landing_amd/csrc
prefetchknob), so every setting runs in a child process; settings alternate, two rounds.
   python bench/prefetch_ahead.py [config3|config4] [formats] [distances] [masks]          # on the GPU box"""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
from bench import build
wl, fmt = sys.argv[1], sys.argv[2]
batch, nlp, Z, c, vals = build(wl, 0, 0, jac_format=fmt, placement_trials=4)
def t_ms(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
out = ["c+J %%.4f ms" %% t_ms(lambda: nlp.eval_c_and_jac(Z, c, vals, write_constants=False))]
out.append("c only %%.4f ms" %% t_ms(lambda: nlp.eval_c(Z, c)))
f, g = nlp.new_f(), nlp.new_Z()
out.append("f+grad+c+J %%.4f ms" %% t_ms(lambda: nlp.eval_all(Z, f, g, c, vals, write_constants=False)))
vals.zero_(); c.zero_()   # the buffers' padding and (unwritten) constant tail hold whatever the placement scan left there
nlp.eval_c_and_jac(Z, c, vals, write_constants=False); torch.cuda.synchronize()
print("  ".join(out), " checksum %%.17g %%.17g" %% (float(vals.sum()), float(c.sum())))
''' % ROOT
lib = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_prefetchknob.so")
wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
FORMATS = sys.argv[2].split(",") if len(sys.argv) > 2 else ("structural", "dense_blocks")
AHEAD = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0, 64, 128, 192, 256, 320, 512, 1024)
MASKS = [int(a) for a in sys.argv[4].split(",")] if len(sys.argv) > 4 else (1,)  # 1 = slice of Z, 2 = boundary vectors, 4 = descriptor
for rnd in range(2):
    for fmt in FORMATS:
        for ahead, mask in [(a, m) for a in AHEAD for m in (MASKS if a else (1,))]:
            env = dict(os.environ, QLN_LIB_PATH=lib, QLN_PREFETCH_AHEAD=str(ahead), QLN_PREFETCH_MASK=str(mask))
            r = subprocess.run([sys.executable, "-c", CHILD, wl, fmt], env=env, capture_output=True, text=True)
            out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]
            print(f"{wl} {fmt:12s} ahead {ahead:5d} mask {mask}: {out}", flush=True)
