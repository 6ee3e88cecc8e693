"""Inner-loop solvers (same public names as pybo.solvers; DIRECT needs nlopt and is out of scope)."""
from .lbfgs import *            # noqa: F401,F403
from . import lbfgs

__all__ = list(lbfgs.__all__)
