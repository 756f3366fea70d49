"""Shared helpers for the parity tests: run the oracle on a LandingBatch with the same
strides/offsets as the product handle, and compare with a stated tolerance."""
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
