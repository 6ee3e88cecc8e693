"""pybo_amd -- MI355X-native GP-posterior + acquisition engine behind pybo's plugin API."""
__version__ = '0.1.0'

from .bayesopt import solve_bayesopt, init_model      # noqa: E402,F401
from . import inits, models, policies, recommenders, solvers   # noqa: E402,F401

__all__ = ['solve_bayesopt', 'init_model']
