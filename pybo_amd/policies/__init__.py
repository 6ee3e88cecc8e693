"""Acquisition policies (same public names as pybo.policies)."""
from .simple import *           # noqa: F401,F403
from . import simple

__all__ = list(simple.__all__)
