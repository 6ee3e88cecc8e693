"""
Initial designs / candidate generators, same names, signatures and RandomState call sequences as
/root/reference/pybo/inits/methods.py:17-77, so that a fixed seed yields the same points
(tests/golden/inits.npz pins init_middle / init_uniform / init_latin bit-for-bit).

init_sobol is the one deliberate difference: the reference drives a 13k-line LGPL direction-number table
(pybo/inits/sobol.py) that must not be copied; here the sequence comes from scipy.stats.qmc.Sobol
(Joe-Kuo numbers as well, but a different generator convention), with the same `skip` draw from the rng.
Candidates are an *input* to the hot path, so bit-parity of this generator is not part of the contract.
"""
import numpy as np

from ..utils import rstate

__all__ = ['init_middle', 'init_uniform', 'init_latin', 'init_sobol', 'init_sobol_device',
           'init_uniform_device']


def _box(bounds):
    b = np.array(bounds, dtype=float, ndmin=2)
    return b[:, 0], b[:, 1] - b[:, 0], len(b)


def init_middle(bounds):
    """The single point at the centre of the box, shape (1, d)."""
    return np.mean(np.array(bounds, dtype=float, ndmin=2), axis=1)[None, :]


def init_uniform(bounds, n=None, rng=None):
    """n i.i.d. uniform points in the box (n defaults to 3*d).  One `rng.rand(n, d)` call."""
    rng = rstate(rng)
    lo, width, d = _box(bounds)
    n = 3 * d if n is None else n
    return lo + width * rng.rand(n, d)


def init_latin(bounds, n=None, rng=None):
    """Latin hypercube: one uniform jitter per cell, then an independent shuffle of every column."""
    rng = rstate(rng)
    lo, width, d = _box(bounds)
    n = 3 * d if n is None else n
    X = lo + width * (np.arange(n)[:, None] + rng.rand(n, d)) / n
    for k in range(d):
        X[:, k] = rng.permutation(X[:, k])
    return X


def init_sobol(bounds, n=None, rng=None):
    """Sobol points; `skip = rng.randint(100, 200)` leading points are discarded as in the reference."""
    from scipy.stats import qmc
    rng = rstate(rng)
    lo, width, d = _box(bounds)
    n = 3 * d if n is None else n
    skip = rng.randint(100, 200)
    eng = qmc.Sobol(d, scramble=False)
    if skip:
        eng.fast_forward(skip)
    return lo + width * eng.random(n)


# ---- grids generated and kept in HBM -------------------------------------------------
def init_sobol_device(bounds, n=None, rng=None, device=0):
    """The points of `init_sobol(bounds, n, rng)` -- same `skip` draw, bit-identical coordinates -- generated
    on the GPU and left there: returns a `DeviceGrid` for `solve_lbfgs(..., xgrid=...)`."""
    from .._lib import DeviceGrid, ShardedDeviceGrid
    rng = rstate(rng)
    d = len(np.array(bounds, dtype=float, ndmin=2))
    n = 3 * d if n is None else n
    skip = rng.randint(100, 200)
    if isinstance(device, (list, tuple)):          # one shard per listed device (pybo_amd.models.ShardedGP)
        return ShardedDeviceGrid('sobol', bounds, n, device, first=skip)
    return DeviceGrid('sobol', bounds, n, first=skip, device=device)


def init_uniform_device(bounds, n=None, rng=None, device=0):
    """n i.i.d. uniform points generated on the GPU (Philox4x32-10 keyed by one 62-bit draw from `rng`) -- the
    device counterpart of `init_uniform`; a different stream of numbers than numpy's MT19937."""
    from .._lib import DeviceGrid, ShardedDeviceGrid
    rng = rstate(rng)
    d = len(np.array(bounds, dtype=float, ndmin=2))
    n = 3 * d if n is None else n
    seed = (int(rng.randint(0, 2 ** 31 - 1)) << 31) | int(rng.randint(0, 2 ** 31 - 1))
    if isinstance(device, (list, tuple)):
        return ShardedDeviceGrid('uniform', bounds, n, device, seed=seed)
    return DeviceGrid('uniform', bounds, n, seed=seed, device=device)
