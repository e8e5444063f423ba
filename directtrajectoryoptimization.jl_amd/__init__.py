"""MI355X-native sparse NLP callback + KKT engine for direct trajectory optimization.

Host-side mirror of the DirectTrajectoryOptimization.jl surface
(src/DirectTrajectoryOptimization.jl:22-35 exports) over the C-ABI in include/dto.h.
The directory name contains a dot, so import it through the repo-root alias `dto_amd`.
"""
from .model import Bound, Constraint, Cost, Dynamics, GeneralConstraint, linear_interpolation  # noqa: F401
from .symbolic.expr import dot, sin, cos, tan, ifelse, minimum, maximum  # noqa: F401
from .solver import NLPData, Options, Solver, get_trajectory, initialize_controls, initialize_states, solve  # noqa: F401
from . import problems  # noqa: F401
