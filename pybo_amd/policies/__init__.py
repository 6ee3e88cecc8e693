"""Acquisition policies; the exported names match pybo.policies (`EI`, `PI`, `UCB`, `Thompson`)."""
from .simple import EI, PI, UCB, Thompson

__all__ = ['EI', 'PI', 'UCB', 'Thompson']
