"""Initial designs; the exported names match pybo.inits (`init_middle|uniform|latin|sobol`)."""
from .methods import init_middle, init_uniform, init_latin, init_sobol

__all__ = ['init_middle', 'init_uniform', 'init_latin', 'init_sobol']
