"""
The Bayesian-optimisation driver with pybo's public surface
(/root/reference/pybo/bayesopt.py: `solve_bayesopt` :193-287, `init_model` :60-120,
`get_component` :125-176, checkpoint helpers :36-55), re-expressed for Python 3 on top of the
MI355X engine.  The loop body is the reference's: policy -> solver -> objective -> add_data ->
recommender -> checkpoint (bayesopt.py:262-276); everything numerically heavy happens inside the model
object, which here is `pybo_amd.models.GP` (HIP kernels behind a C-ABI) instead of `reggie`.

Differences from the reference, all deliberate:
  * `init_model` builds a fixed-hyper-parameter GP with the reference's heuristic initial values
    (bayesopt.py:98-102) instead of wrapping it in `reggie.MCMC(n=10, burn=100)` (bayesopt.py:115): the
    hyper-posterior sampler is reggie-internal, unpinned, and excluded from the hot path (SURVEY F10,
    R7).  The priors are recorded on the model's params (same `set_prior` calls) but not sampled.
  * checkpoints are binary pickles (the reference opens the file in text mode, a Python-2-ism).
  * `get_component` reports a bad component with a working format string (the reference's
    '{:r}' at bayesopt.py:138 is itself a ValueError).
"""
import collections
import functools
import inspect
import os.path
import pickle

import numpy as np

from . import inits
from . import policies
from . import recommenders
from . import solvers
from .utils import rstate

__all__ = ['solve_bayesopt', 'init_model']

Info = collections.namedtuple('Info', ['x', 'y', 'xbest'])


# -- checkpointing ---------------------------------------------------------------------------------
def safe_dump(model, info, filename=None):
    """Write (model, info) to `filename` if one was given (atomically: tmp file + rename)."""
    if filename is None:
        return
    tmp = filename + '.tmp'
    with open(tmp, 'wb') as fp:
        pickle.dump((model, info), fp)
    os.replace(tmp, filename)


def safe_load(filename=None):
    """Read a checkpoint; (None, empty Info) if there is none."""
    if filename is not None and os.path.exists(filename):
        with open(filename, 'rb') as fp:
            return pickle.load(fp)
    return None, Info([], [], [])


# -- model bootstrap -------------------------------------------------------------------------------
def init_model(f, bounds, ninit=None, design='latin', log=None, rng=None, kernel='se'):
    """Evaluate an initial design (resumable) and return a GP with heuristic hyper-parameters."""
    from . import models
    rng = rstate(rng)
    bounds = np.array(bounds, dtype=float, ndmin=2)
    ninit = 3 * len(bounds) if ninit is None else ninit
    model, info = safe_load(log)
    if model is not None:
        return model
    if len(info.x) == 0:
        design = getattr(inits, 'init_' + design)
        info.x.extend(design(bounds, ninit, rng))
        info.y.extend(np.nan for _ in range(ninit))

    for i, x in enumerate(info.x):
        if np.isnan(info.y[i]):
            info.y[i] = f(x)
        safe_dump(None, info, filename=log)

    # heuristic hyper-parameters, bayesopt.py:98-102
    sn2 = 1e-6
    rho = max(info.y) - min(info.y) if len(info.y) > 1 else 1.0
    rho = 1.0 if rho < 1e-1 else rho
    ell = 0.25 * (bounds[:, 1] - bounds[:, 0])
    bias = np.mean(info.y) if len(info.y) > 0 else 0.0

    model = models.make_gp(sn2, rho, ell, bias, kernel=kernel)
    model.params['like.sn2'].set_prior('horseshoe', 0.1)
    model.params['kern.rho'].set_prior('lognormal', np.log(rho), 1.0)
    model.params['kern.ell'].set_prior('uniform', ell / 100, ell * 10)
    model.params['mean.bias'].set_prior('normal', bias, rho)
    model.add_data(info.x, info.y)

    safe_dump(model, info, filename=log)
    return model


# -- plugin resolution -----------------------------------------------------------------------------
def get_component(value, module, rng, lstrip=''):
    """
    Resolve a component given as a name, a callable, or (name-or-callable, kwargs):
      * names are matched case-insensitively against `module.__all__` after stripping `lstrip`;
      * kwargs must be a subset of the callable's defaulted arguments, `rng` excluded;
      * `rng` is injected iff the callable has an argument called `rng`.
    Errors are ValueError, as in the reference (bayesopt.py:138,153,167).
    """
    kwargs = {}
    if isinstance(value, (list, tuple)):
        try:
            value, kwargs = value
            kwargs = dict(kwargs)
        except (ValueError, TypeError):
            raise ValueError('invalid component: {!r}'.format(value))

    if callable(value):
        func = value
    else:
        for fname in module.__all__:
            func = getattr(module, fname)
            short = fname[len(lstrip):] if fname.startswith(lstrip) else fname
            if short.lower() == value:
                break
        else:
            raise ValueError('invalid component: {!s}'.format(value))

    spec = inspect.getfullargspec(func)
    valid = set(spec.args[-len(spec.defaults):]) if spec.defaults else set()
    valid.discard('rng')
    if not valid.issuperset(kwargs.keys()):
        raise ValueError('unknown arguments for {:s}: {:s}'.format(
            getattr(func, '__name__', repr(func)), ', '.join(kwargs.keys())))

    if 'rng' in spec.args:
        kwargs['rng'] = rng
    return functools.partial(func, **kwargs) if kwargs else func


# -- verbose formatting ----------------------------------------------------------------------------
int2str = '{:03d}'.format
float2str = '{: .3f}'.format


def array2str(a):
    return np.array2string(np.asarray(a), formatter=dict(float=float2str, int=int2str))


# -- the meta solver -------------------------------------------------------------------------------
def solve_bayesopt(objective, bounds, model=None, niter=100, policy='ei', solver='lbfgs',
                   recommender='latent', ninit=None, verbose=False, log=None, rng=None):
    """
    Maximise `objective` over the box `bounds` ((d,2) array-like) by Bayesian optimisation.

    `policy`, `solver`, `recommender` are each a name, a callable, or a (name-or-callable, kwargs)
    pair; `model` is any object with the model protocol (copy / add_data / predict / get_improvement /
    get_tail / sample_f), by default a `pybo_amd.models.GP` built by `init_model`.

    Returns (xbest, model, Info(x, y, xbest)) with the Info fields as arrays (bayesopt.py:285-287).
    """
    rng = rstate(rng)
    bounds = np.array(bounds, dtype=float, ndmin=2)

    policy = get_component(policy, policies, rng)
    solver = get_component(solver, solvers, rng, lstrip='solve_')
    recommender = get_component(recommender, recommenders, rng, lstrip='best_')

    model_, info = safe_load(log)
    if model is None and model_ is None:
        # NOTE (kept from the reference, bayesopt.py:243-259): `info` was loaded BEFORE init_model ran,
        # so the initial design lives in the model but not in the returned trace, and the "single point
        # in the middle" below is evaluated as well.  Policies therefore see info.x without the design.
        model = init_model(objective, bounds, ninit, log=log, rng=rng)
    else:
        model = model_ if model_ is not None else model.copy()

    # a user-supplied empty model is started from the centre of the box (bayesopt.py:253-259)
    if len(info.x) == 0:
        x = inits.init_middle(bounds)[0]
        y = objective(x)
        info.x.append(x)
        info.y.append(y)
        model.add_data(x, y)
        safe_dump(model, info, filename=log)

    xbest = info.xbest[-1] if len(info.xbest) else None
    for i in range(len(info.xbest), niter):
        index = policy(model, bounds, info.x)
        x, _ = solver(index, bounds)

        y = objective(x)
        model.add_data(x, y)
        xbest = recommender(model, bounds, info.x)

        info.x.append(x)
        info.y.append(y)
        info.xbest.append(xbest)
        safe_dump(model, info, filename=log)

        if verbose:
            print('i={:s}, x={:s}, y={:s}, xbest={:s}'.format(
                int2str(i), array2str(x), float2str(y), array2str(xbest)))

    info = Info(*[np.array(_) for _ in info])
    return xbest, model, info
