"""
Small host-side helpers.  Mirrors the one utility of pybo the BO loop depends on:
`rstate` (/root/reference/pybo/utils.py:16-25).  The shell/interactive query wrappers of the reference
(utils.py:28-58) are objective-side I/O and out of scope.
"""
import numpy as np

__all__ = ['rstate']


def rstate(rng=None):
    """Pass a RandomState through untouched; anything else (None, int seed, ...) seeds a new one."""
    return rng if isinstance(rng, np.random.RandomState) else np.random.RandomState(rng)
