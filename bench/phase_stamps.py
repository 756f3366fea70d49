"""Measurement aid: per-wave phase timing of the hot kernel from the -DQLN_DIAG build's s_memtime stamps.
Read the SHARES, not the absolute run time (the stamps fence the instruction stream)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_landing_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "quadruped_landing_amd", "csrc", "libqln_hip_diag.so")
import torch
from bench import build

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
fmt = sys.argv[2] if len(sys.argv) > 2 else "dense_blocks"
batch, nlp, Z, c, vals = build(wl, 0, 0, placement_trials=8 if os.environ.get("QLN_ABLATE_PLACED") else 1, jac_format=fmt)
L = _lib.lib()
stamps = torch.zeros(batch.B * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    nlp.eval_c_and_jac(Z, c, vals, write_constants=False)
torch.cuda.synchronize()
L.qln_diag_set_stamps.argtypes = [ctypes.c_void_p]
assert L.qln_diag_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); nlp.eval_c_and_jac(Z, c, vals, write_constants=False); e1.record()
torch.cuda.synchronize()
print("diag kernel time %.3f ms" % e0.elapsed_time(e1))
s = stamps.cpu().numpy().reshape(batch.B, 16).astype(np.float64)
t0 = s[:, 0].min()
if fmt == "structural":  # multi-chunk problems: the stamps of the last chunk
    names = {1: "stage (scalar loads + Z loads + c1/c2)", 2: "value phase (RK4, c rows, c drain)", 3: "Jacobian base quantities",
             4: "write the pattern's values to LDS", 5: "drain to HBM", 15: "tail"}
else:
  names = {1: "stage (scalar loads + Z loads + c1/c2)", 2: "value phase (RK4, c rows, c drain)", 3: "Jacobian base + zero tile",
           4: "sub-tile 0", 5: "sub-tile 1", 6: "sub-tile 2", 7: "sub-tile 3", 8: "sub-tile 4", 15: "tail"}
prev = s[:, 0]
tot = s[:, 15] - s[:, 0]
print("wave lifetime: median %.0f cycles, mean %.0f, p90 %.0f" % (np.median(tot), tot.mean(), np.percentile(tot, 90)))
for i in (1, 2, 3, 4, 5, 6, 7, 8, 15):
    if i not in names or not s[:, i].any():
        continue
    d = s[:, i] - prev
    print("%-42s median %8.0f  mean %8.0f  p90 %8.0f  share %.1f%%" % (names[i], np.median(d), d.mean(), np.percentile(d, 90), 100 * d.mean() / tot.mean()))
    prev = s[:, i]
span = s[:, 15].max() - t0
print("kernel span %.0f cycles; concurrent waves (sum lifetime / span) = %.0f" % (span, tot.sum() / span))
