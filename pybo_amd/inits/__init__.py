"""Initial designs (same public names as pybo.inits)."""
from .methods import *          # noqa: F401,F403
from . import methods

__all__ = list(methods.__all__)
