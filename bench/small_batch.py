"""Measurement aid: fused c+J launch time against the batch size (small-batch regime), for the library / variant
selected by QLN_LIB_PATH / QLN_VARIANT.   python bench/small_batch.py [dense_blocks|structural] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadruped_landing_amd import HybridNLP, problem_gen as PG

fmt = sys.argv[1] if len(sys.argv) > 1 else "dense_blocks"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = []
for B in (64, 256, 1024, 2048, 4096, 8192, 16384):
    batch = PG.make_batch(B, N, max(2, N // 3 + 1), 1, seed=0)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, jac_format=fmt,
                    stream=torch.cuda.current_stream())
    Z, c, v = nlp.upload_Z(batch.Z), nlp.new_c(), nlp.new_vals()
    ms = nlp.time_c_and_jac(Z, c, v, warmup=10, iters=100)
    out.append("B=%d: %.1f us" % (B, 1e3 * float(np.median(ms))))
print("variant %s %s N=%d  " % (os.environ.get("QLN_VARIANT", "-"), fmt, N) + "  ".join(out))
