"""
Log-densities of the hyper-priors `init_model` attaches (`/root/reference/pybo/bayesopt.py:108-111`):
    like.sn2   'horseshoe', scale          kern.rho   'lognormal', mu, sigma
    kern.ell   'uniform', a, b             mean.bias  'normal', mu, s2
The densities themselves live in `reggie` (absent, unpinned); these are the textbook forms, up to additive
constants (a sampler only needs differences):
    horseshoe(scale): the tight closed-form bound of Carvalho et al.,  log( log(1 + 3 (scale/x)^2) )
    lognormal(mu, sigma): -(log x - mu)^2 / (2 sigma^2) - log x
    uniform(a, b): 0 inside [a, b], -inf outside (element-wise for vectors)
    normal(mu, s2): -(x - mu)^2 / (2 s2)          (s2 is a VARIANCE: init_model passes rho)
A parameter without a prior is flat on its support (positive for sn2 / rho / ell).
"""
import math

import numpy as np

__all__ = ['log_prior']


def log_prior(prior, x):
    """Sum of log-densities of `x` (scalar or vector) under `prior` = (kind, *args) or None.
    (Scalars take a plain-`math` path: the sampler calls this four times per state, and numpy's per-call overhead
    on 0-d values was the largest single cost of a hyper-parameter update once the likelihoods were batched.)"""
    if prior is None:
        return 0.0
    kind, args = prior[0], prior[1:]
    scalar = np.ndim(x) == 0
    if kind == 'uniform':
        if scalar and np.ndim(args[0]) == 0 and np.ndim(args[1]) == 0:
            return 0.0 if float(args[0]) <= x <= float(args[1]) else -np.inf
        x = np.atleast_1d(np.asarray(x, dtype=float))
        a, b = (np.broadcast_to(np.asarray(v, dtype=float), x.shape) for v in args)
        return 0.0 if np.all((x >= a) & (x <= b)) else -np.inf
    if kind == 'lognormal':
        mu, sigma = float(args[0]), float(args[1])
        if scalar:
            if x <= 0:
                return -np.inf
            lx = math.log(x)
            return -0.5 * ((lx - mu) / sigma) ** 2 - lx
        x = np.asarray(x, dtype=float)
        if np.any(x <= 0):
            return -np.inf
        return float(np.sum(-0.5 * ((np.log(x) - mu) / sigma) ** 2 - np.log(x)))
    if kind == 'normal':
        mu, s2 = float(args[0]), float(args[1])
        if scalar:
            return -0.5 * (x - mu) ** 2 / s2
        return float(np.sum(-0.5 * (np.asarray(x, dtype=float) - mu) ** 2 / s2))
    if kind == 'horseshoe':
        scale = float(args[0])
        if scalar:
            if x <= 0:
                return -np.inf
            return math.log(math.log1p(3.0 * (scale / x) ** 2))
        x = np.asarray(x, dtype=float)
        if np.any(x <= 0):
            return -np.inf
        return float(np.sum(np.log(np.log1p(3.0 * (scale / x) ** 2))))
    raise ValueError('unknown prior {!r}'.format(kind))
