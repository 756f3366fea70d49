"""Measurement aid: time the separate entry points on the bench workload (not part of the product)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build

def t_ms(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
fmt = sys.argv[2] if len(sys.argv) > 2 else "dense_blocks"
# QLN_ABLATE_PLACED=1: the Jacobian buffer placed across two 32-GiB regions (as bench.py does), else a plain allocation
batch, nlp, Z, c, vals = build(wl, 0, 0, placement_trials=8 if os.environ.get("QLN_ABLATE_PLACED") else 1, jac_format=fmt)
f = nlp.new_f(); g = nlp.new_Z()
print(wl, "fused c+J      : %.3f ms" % t_ms(lambda: nlp.eval_c_and_jac(Z, c, vals, write_constants=False)))
print(wl, "fused +consts  : %.3f ms" % t_ms(lambda: nlp.eval_c_and_jac(Z, c, vals, write_constants=True)))
gg = nlp.new_Z()
print(wl, "eval_all       : %.3f ms" % t_ms(lambda: nlp.eval_all(Z, f, gg, c, vals, write_constants=False)))
print(wl, "J only         : %.3f ms" % t_ms(lambda: nlp.jac_c(Z, vals, write_constants=False)))
print(wl, "c only         : %.3f ms" % t_ms(lambda: nlp.eval_c(Z, c)))
print(wl, "objective      : %.3f ms" % t_ms(lambda: nlp.eval_f(Z, f)))
print(wl, "f + c, 1 launch: %.3f ms" % t_ms(lambda: nlp.eval_f_and_c(Z, f, c)))
print(wl, "gradient       : %.3f ms" % t_ms(lambda: nlp.grad_f(Z, g)))
print(wl, "memset vals    : %.3f ms" % t_ms(lambda: vals.zero_()))
