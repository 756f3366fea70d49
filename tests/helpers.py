"""Shared helpers for the parity tests: run the oracle on a LandingBatch with the same
strides/offsets as the product handle, and compare with a stated tolerance."""
import os
import subprocess

import numpy as np

from oracle import oracle as O


def oracle_model(model):
    m = O.Model()
    m.g, m.mb, m.mf, m.lb, m.l1, m.l2 = model.g, model.mb, model.mf, model.lb, model.l1, model.l2
    return m


def oracle_batch(batch, nlp, want_c=True, want_j=True, want_f=False, want_grad=False, nthreads=4):
    """Evaluate `batch` with the C oracle in the layout of the product handle `nlp`."""
    Zs = np.zeros((batch.B, nlp.z_stride))
    Zs[:, : nlp.n_nlp] = batch.Z
    return O.batch_eval(batch.N, oracle_model(batch.model), batch.k_trans, batch.init_mode, batch.x0, batch.xf,
                        batch.obj, Zs.reshape(-1), nlp.z_stride, nlp.c_off, nlp.j_off, nlp.dims.c_total,
                        nlp.dims.j_total, want_c, want_j, want_f, want_grad, nthreads)


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(|b|, floor) over entries where both are finite."""
    a, b = np.asarray(a), np.asarray(b)
    ok = np.isfinite(b)
    d = np.abs(a[ok] - b[ok])
    s = np.maximum(np.abs(b[ok]), floor)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(d == 0, 0.0, d / s)
    return float(r.max()) if r.size else 0.0


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_c_host(out_dir):
    """gcc-compile tests/host_c/moi_host.c against include/qln_evaluator.h and libqln_hip.so."""
    csrc = os.path.join(ROOT, "quadruped_landing_amd", "csrc")
    exe = os.path.join(str(out_dir), "moi_host")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_c", "moi_host.c"), "-L", csrc, "-lqln_hip", "-lm",
                           f"-Wl,-rpath,{csrc}", "-o", exe])
    return exe


def write_problem_file(path, nb, Z, Z0=None):
    m = nb.model
    with open(path, "w") as fh:
        fh.write(f"{nb.N} {int(nb.k_trans[0])} {int(nb.init_mode[0])}\n")
        for arr in ([m.g, m.mb, m.mf, m.lb, m.l1, m.l2], nb.x0[0], nb.xf[0], np.asarray(nb.obj).reshape(-1), Z) + (
                (Z0,) if Z0 is not None else ()):
            fh.write(" ".join(repr(float(x)) for x in np.asarray(arr).reshape(-1)) + "\n")
