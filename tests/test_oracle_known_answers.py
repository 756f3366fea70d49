"""Pins the oracle: the known-answer values the reference's notebook printed for the run whose
solution is src/data_6.csv.  The reference itself (Julia) cannot run here."""
import os

import numpy as np
import pytest

from oracle import np_oracle as NP
from oracle import oracle as O


@pytest.fixture(scope="module")
def nb():
    nlp, xinit, xterm, Xref, Uref = O.notebook_problem()
    return dict(nlp=nlp, xinit=xinit, xterm=xterm, Xref=Xref, Uref=Uref)


def _Z(golden_dir, i):
    return np.loadtxt(os.path.join(golden_dir, f"data_{i}.csv"))


def test_KA3_dimensions(nb):
    # src/main.ipynb:217-226: 1215 variables, 1032 equalities, 61 inequalities, dense 1093 x 1215
    nlp = nb["nlp"]
    assert nlp.n_nlp == 1215 and nlp.m_nlp == 1093
    assert nlp.cinds() == [(1, 15), (16, 29), (30, 929), (930, 990), (991, 1031), (1032, 1032), (1033, 1093)]
    lb, ub = nlp.bounds()
    assert np.count_nonzero(np.isinf(ub)) == 61 and np.count_nonzero(ub == 0) == 1032 and not lb.any()
    assert 1253880 + 74115 == nlp.m_nlp * nlp.n_nlp
    assert 1253880 == 1032 * 1215 and 74115 == 61 * 1215


def test_KA1_objective(nb, golden_dir):
    # src/main.ipynb:710  Objective 1.1608112892558562e+02
    assert nb["nlp"].eval_f(_Z(golden_dir, 6)) == 1.1608112892558562e02


def test_KA2_constraint_violation(nb, golden_dir):
    # src/main.ipynb:712  Constraint violation 1.4928675395736724e-06 (max over equality rows)
    c = nb["nlp"].eval_c(_Z(golden_dir, 6))
    neq = nb["nlp"].cinds()[5][1]
    assert np.max(np.abs(c[:neq])) == 1.4928675395736724e-06
    assert c[neq:].min() >= 0  # the 61 clearance rows are feasible


def test_KA4_printed_residuals(nb, golden_dir):
    # src/main.ipynb cell 9 output (Z_sol[1:15] - xinit) and cell 11 (F1y+F2y = 98.10000000000001)
    Z = _Z(golden_dir, 6)
    c = nb["nlp"].eval_c(Z)
    printed = np.array([-3.4916514124461173e-14, -4.0967229608668276e-14, 1.4099832412739488e-14,
                        -1.7424461934630155e-9, 2.780551258041212e-15, 3.519406988061746e-14,
                        2.6922908347160046e-15, -4.790380523959208e-16, -5.329070518200751e-15,
                        6.661338147750939e-16, 0.0, 0.0, 1.0416919756127546e-15, 8.881784197001252e-16,
                        -5.2722935594050705e-18])
    assert np.array_equal(c[:15], printed)
    assert Z[-19] + Z[-17] == 98.10000000000001
    assert Z[-19] == 44.56221189408092 and Z[-17] == 53.53778810591909


def test_KA5_initial_guess_infeasibility(nb):
    # src/main.ipynb:232  iteration 0 inf_pr 3.13e-01 at Z0 = packZ(nlp, Xguess, Uref)
    Z0 = O.notebook_initial_guess(61, 21, nb["xinit"], nb["xterm"], nb["Uref"])
    c0 = nb["nlp"].eval_c(Z0)
    neq = nb["nlp"].cinds()[5][1]
    assert f"{np.max(np.abs(c0[:neq])):.2e}" == "3.13e-01"


@pytest.mark.parametrize("i", [1, 2, 3, 4, 5])
def test_other_solved_trajectories_are_feasible(nb, golden_dir, i):
    """data_1..5.csv: same N/k_trans/init_mode, x0 = their first 15 entries: they pin
    the constraint groups c2..c7 (their cost weights are unknown, so not the objective)."""
    Z = _Z(golden_dir, i)
    nlp = O.OracleNLP(61, 21, 1, Z[:15], nb["xterm"], nb["nlp"].cost)
    c = nlp.eval_c(Z)
    ci = nlp.cinds()
    assert np.max(np.abs(c[ci[2][0] - 1 : ci[2][1]])) <= 5e-6   # dynamics (Ipopt constr_viol_tol 1e-3 scaled)
    assert np.max(np.abs(c[ci[3][0] - 1 : ci[4][1]])) <= 1e-10  # both contact groups
    assert np.max(np.abs(c[ci[1][0] - 1 : ci[1][1]])) <= 1e-6   # terminal
    assert c[ci[6][0] - 1 :].min() >= 0                         # clearance
    if i >= 3:  # data_1/2 predate the final-control row (their F1y+F2y is not -mb*g)
        assert abs(c[ci[5][0] - 1]) <= 1e-12


def test_spot_value_of_rk4_step(nb):
    # contact1_dynamics_rk4(model, xinit, Uref[1])
    got = O.contact_dynamics_rk4(1, nb["xinit"], nb["Uref"][0])
    want = np.array([-0.2, 0.44728920668792749, -0.52512248392509375, 0, 0, -0.5, 0.19899509500000001, 0,
                     -6.2641839053463304, -1.4766203267948965, 0, 0, 0, -1.0098100000000001, 0.001])
    assert np.array_equal(got, want)


def test_two_restatements_agree(nb, golden_dir):
    Z = _Z(golden_dir, 6)
    assert np.array_equal(nb["nlp"].eval_c(Z), NP.eval_c(61, 21, 1, nb["xinit"], nb["xterm"], Z)) or \
        np.max(np.abs(nb["nlp"].eval_c(Z) - NP.eval_c(61, 21, 1, nb["xinit"], nb["xterm"], Z))) <= 1e-15
    assert abs(nb["nlp"].eval_f(Z) - NP.eval_f(61, nb["nlp"].cost, Z)) <= 1e-12 * 116.0


def test_KA6_iteration0_objective_at_the_bound_pushed_point(nb):
    """src/main.ipynb:232: Ipopt's iteration-0 objective 1.8380701e+00.  It is not eval_f(Z0) (= 1.5438468): Ipopt
    evaluates at Z0 pushed inside the (relaxed) variable bounds of solve() (src/moi.jl:51-67).  Reproducing all eight
    printed digits pins eval_f at a non-solution point AND the variable bounds incl. quirk Q6: with the bounds on
    F1y/F2y that the source comment describes, the value is 1.8371626 instead."""
    from quadruped_landing_amd import nlp as NL

    Z0 = O.notebook_initial_guess(61, 21, nb["xinit"], nb["xterm"], nb["Uref"])
    assert f"{nb['nlp'].eval_f(Z0):.7e}" == "1.5438468e+00"
    Zp = NL.ipopt_initial_point(Z0, *NL.variable_bounds(61))
    assert np.count_nonzero(Zp != Z0) == 120  # 60 time steps sitting on a bound, 60 x1_{k+1} = 0 on the Q6 bound
    assert f"{nb['nlp'].eval_f(Zp):.7e}" == "1.8380701e+00"
    assert f"{nb['nlp'].eval_f(NL.ipopt_initial_point(Z0, *NL.variable_bounds_forces(61))):.7e}" == "1.8371626e+00"
    # the pushed variables do not enter the worst constraint row: inf_pr stays 3.13e-01 (KA5)
    c = nb["nlp"].eval_c(Zp)
    assert f"{np.max(np.abs(c[: nb['nlp'].cinds()[5][1]])):.2e}" == "3.13e-01"
