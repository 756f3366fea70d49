"""Would packing 64 knots into every wave pay for the structural format?  At N = 40 a wave of the structural launch uses 39 of its
64 lanes (lane = knot).  Probe: the same number of knot points as N = 65 problems -- 64 knots each, so that 64-knot chunks
(tuning variants 14 / 15) fill every lane -- against the shipping 40-knot chunks at N = 40 and at N = 65 (40 + 24).  Reports time,
ns per knot point and TB/s on the strict byte count.      QLN_LIB_PATH=<tuning build> python bench/packing_probe.py     (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import torch, bench
name, B, N, kt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bench.WORKLOADS[name] = dict(B=B, N=N, k_trans=kt, ragged=False, desc=name)
batch, nlp, Z, c, vals = bench.build(name, 0, 0, jac_format="structural", placement_trials=4)
def t_ms(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
ms = t_ms(lambda: nlp.eval_c_and_jac(Z, c, vals, write_constants=False))
by = float(np.sum(bench.strict_bytes(N, batch.k_trans)))
print("%%.4f ms  %%.3f ns per knot point  %%.2f TB/s on the strict bytes" %% (ms, ms * 1e6 / (B * N), by / ms / 1e9))
''' % ROOT
CASES = [("N40", 65536, 40, 14), ("N65", 40330, 65, 22)]
for rnd in range(2):
    for name, B, N, kt in CASES:
        for variant in (0, 14, 15):
            env = dict(os.environ, QLN_VARIANT=str(variant))
            r = subprocess.run([sys.executable, "-c", CHILD, name, str(B), str(N), str(kt)], env=env, capture_output=True, text=True)
            out = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]
            print(f"{name} B={B} variant {variant:2d} ({'shipping <KC 40>' if variant == 0 else '<KC 64, %d wave(s) per SIMD budget>' % (variant - 13)}): {out}", flush=True)
