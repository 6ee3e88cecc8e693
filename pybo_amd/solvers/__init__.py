"""Inner-loop solvers; `solve_lbfgs` as in pybo.solvers (DIRECT needs nlopt and is out of scope)."""
from .lbfgs import solve_lbfgs

__all__ = ['solve_lbfgs']
