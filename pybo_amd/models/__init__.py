"""Probabilistic models with the protocol pybo expects of `reggie` objects."""
from .gp import GP, make_gp, RFFSampleDevice      # noqa: F401
from .mcmc import MCMC                            # noqa: F401
from .sharded import ShardedGP                    # noqa: F401

__all__ = ['GP', 'make_gp', 'MCMC', 'ShardedGP']
