"""MI355X-native low-rank Rosenbrock/ADI engine — host-side mirror of the reference API.

Import as `import dre_amd` (the directory name carries a dot and cannot be imported directly;
`dre_amd.py` at the repo root loads this package under that name).
"""
from . import _lib
from ._lib import DREError
from .device import Context, DenseMatrix, DeviceLDLt, Factor, Pencil, default_context, set_default_context
from .api import (ADI, ADISolver, Backslash, BlockLinearProblem, BlockLinearSolver, ShermanMorrisonWoodbury, init, step_, isdone, solve_, Callbacks, DRESolution, GALEProblem, GAREProblem, GDREProblem, GMRES, LDLt, LowRankUpdate, Newton, Ros1, Ros2, ScaledPencil, Shifts,
                  compress_, concatenate_, delta, dot, gare_residual, lyapunov_apply, solve_gmres, heuristic_shifts, lowrank, lr_update, norm, orthf, quadratic_forcing,
                  residual, solve, solve_gale, solve_gare, solve_gdre, superlinear_forcing)
from .steel_profile import SIZES, initial_value, steel_profile

__all__ = [n for n in dir() if not n.startswith("_")]
