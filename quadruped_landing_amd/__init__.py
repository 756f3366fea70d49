"""MI355X-native batched NLP evaluator for the hybrid planar-quadruped landing problem.

Scope: the evaluator hot path of zixinz990/quadruped_landing (src/planar_quadruped.jl,
src/constraints.jl, src/costs.jl, src/nlp.jl behind the MOI callbacks of src/moi.jl:1-33),
as hand-written gfx950 HIP kernels behind the C ABI of include/qln_evaluator.h.
"""
from .planar_quadruped import PlanarQuadruped, control_dim, state_dim  # noqa: F401
from .quadratic_cost import LQRCost, lqr_objective  # noqa: F401
from .ref_traj import reference_trajectory  # noqa: F401
from .nlp import (HybridNLP, cinds, constraint_bounds, num_duals, num_primals, packZ, uinds, unpackZ,  # noqa: F401
                  variable_bounds, xinds)
from . import moi, problem_gen, trajectory_io  # noqa: F401

__version__ = "0.1.0"
