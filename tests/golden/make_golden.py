"""Generates tests/golden/*.npz from the C oracle AFTER it has passed the notebook known answers
(tests/test_oracle_known_answers.py).  These are "derived from a restatement validated by KA1/KA2",
not reference output: the reference (Julia) cannot be executed in this pipeline.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    nlp, xinit, xterm, Xref, Uref = O.notebook_problem()
    Zsol = np.loadtxt(os.path.join(HERE, "data_6.csv"))
    assert nlp.eval_f(Zsol) == 1.1608112892558562e02  # KA1, src/main.ipynb:710
    neq = nlp.cinds()[5][1]
    assert np.max(np.abs(nlp.eval_c(Zsol)[:neq])) == 1.4928675395736724e-06  # KA2, src/main.ipynb:712
    Z0 = O.notebook_initial_guess(61, 21, xinit, xterm, Uref)
    out = {"x0": xinit, "xf": xterm, "cost": nlp.cost, "Z0": Z0}
    for tag, Z in (("sol", Zsol), ("guess", Z0)):
        out[f"f_{tag}"] = np.array(nlp.eval_f(Z))
        out[f"grad_{tag}"] = nlp.grad_f(Z)
        out[f"c_{tag}"] = nlp.eval_c(Z)
        out[f"jac_{tag}"] = nlp.jac_c_coo(Z)
    r, c = nlp.jac_structure()
    out["rows"], out["cols"] = r, c
    np.savez_compressed(os.path.join(HERE, "notebook_N61.npz"), **out)

    # single-knot cases: 12 per mode plus jump knots (x, u -> x+, J)
    rng = np.random.default_rng(20240611)
    xs, us, modes, xn, Js = [], [], [], [], []
    for mode in (1, 2, 3):
        for _ in range(12):
            x = rng.normal(size=15)
            u = np.concatenate([rng.normal(size=4) * 60.0, [rng.uniform(0.001, 0.02)]])
            xs.append(x), us.append(u), modes.append(mode)
            xn.append(O.contact_dynamics_rk4(mode, x, u))
            Js.append(O.contact_jacobian(mode, x, u))
    np.savez_compressed(os.path.join(HERE, "single_knot.npz"), x=np.array(xs), u=np.array(us), mode=np.array(modes),
                        xn=np.array(xn), J=np.array(Js))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
