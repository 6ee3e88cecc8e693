"""
Bayesian-optimisation driver with pybo's public surface, running on the MI355X engine.

Public names and semantics follow /root/reference/pybo/bayesopt.py:
    solve_bayesopt(objective, bounds, model=None, niter=100, policy='ei', solver='lbfgs',
                   recommender='latent', ninit=None, verbose=False, log=None, rng=None)      (:193-287)
    init_model(f, bounds, ninit=None, design='latin', log=None, rng=None)                     (:60-120)
    get_component(value, module, rng, lstrip='')                                               (:125-176)
    Info, safe_dump, safe_load                                                                 (:36-55)
One step of the loop is: policy -> solver -> objective -> model.add_data -> recommender -> checkpoint
(:262-276); every numerically heavy call lands in the model object, here `pybo_amd.models.GP` (HIP kernels
behind a C-ABI) instead of `reggie`.

Behaviour kept on purpose (it decides which points get queried, so it is part of parity):
  * a model handed in by the caller is copied and, if the trace is empty, first fed the centre of the box
    (:249-259);
  * with the default model, `init_model` evaluates a 3d-point latin design that lives in the model but NOT
    in the returned trace, because the trace was read before the model was built (:243-246), and the box
    centre is evaluated on top of it;
  * the loop resumes at `len(trace.xbest)` after a checkpoint reload (:262).
Deliberate differences:
  * `init_model` builds the GP with the reference's heuristic initial values and priors (:98-111) and wraps
    it in `pybo_amd.models.MCMC(n=10, burn=100)` like the reference (:115) -- but the sampler is this
    build's own (random-direction slice sampling, models/mcmc.py): reggie's is absent and unpinned;
  * checkpoints are binary pickles written atomically (the reference opens the file in text mode);
  * a malformed component tuple raises a ValueError with a working message (the reference's '{:r}' format at
    :138 is itself an error).
"""
import collections
import functools
import inspect
import os
import pickle

import numpy as np

from . import inits
from . import policies
from . import recommenders
from . import solvers
from .utils import rstate

__all__ = ['solve_bayesopt', 'init_model']

Info = collections.namedtuple('Info', ['x', 'y', 'xbest'])


# ----------------------------------------------------------------------------------------------------
# checkpointing
# ----------------------------------------------------------------------------------------------------
def safe_dump(model, info, filename=None):
    """Persist (model, trace); a None filename disables checkpointing."""
    if filename is None:
        return
    if _SPMD['on'] and _spmd_rank() != 0:    # SPMD runs: the ranks hold the same state, one of them writes it
        return
    # the trace goes to disk as three arrays, not as lists of thousands of small ones (pickling a list of 8192
    # (d,) arrays costs ~25 ms per iteration, more than a warm BO step); safe_load turns them back into lists
    packed = tuple(np.array(col, dtype=float) for col in info)
    scratch = '{}.part{}'.format(filename, os.getpid())
    with open(scratch, 'wb') as fh:
        pickle.dump((model, _PackedTrace(*packed)), fh, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(scratch, filename)


def _load_local(filename):
    if filename is None or not os.path.exists(filename):
        return None, Info([], [], [])
    with open(filename, 'rb') as fh:
        model, info = pickle.load(fh)
    if isinstance(info, _PackedTrace):
        info = Info(list(info.x), [float(v) for v in info.y], list(info.xbest))
    return model, info


def safe_load(filename=None):
    """Return the stored (model, trace), or (None, empty trace) when there is nothing to resume.

    SPMD runs: only rank 0 writes the checkpoint, so only rank 0 READS it -- what it found (possibly nothing) is
    broadcast, and every rank resumes from the same state.  (Each rank reading the file on its own raced rank 0's next
    write, and needed a shared file system: ranks could load different traces, issue different numbers of objective
    exchanges and record rank 0's pair under the wrong index.)"""
    if filename is not None and _SPMD['on']:
        from . import dist as pdist
        d = pdist._dist()
        if d is not None and d.get_world_size(_SPMD['group']) > 1:
            box = [None]
            if _spmd_rank() == 0:
                # a checkpoint rank 0 cannot read must fail EVERY rank: the others are about to block in the broadcast
                try:
                    box = [('ok', _load_local(filename))]
                except BaseException as exc:      # noqa: BLE001 -- re-raised below, on every rank
                    box = [('error', '%s: %s' % (type(exc).__name__, exc))]
            d.broadcast_object_list(box, src=0, group=_SPMD['group'])
            status, payload = box[0]
            if status != 'ok':
                raise RuntimeError('rank 0 could not load the checkpoint %r: %s' % (filename, payload))
            model, info = payload
            return model, Info(list(info.x), list(info.y), list(info.xbest))
    return _load_local(filename)


_PackedTrace = collections.namedtuple('_PackedTrace', ['x', 'y', 'xbest'])


class _Rows(list):
    """The trace columns x / xbest: a list of (d,) points, as in the reference (`info.x.append(x)`,
    pybo/bayesopt.py:271), that ALSO keeps its rows in one growing array and hands that to numpy (`__array__`):
    every policy and recommender call starts with np.array(X) of the whole trace -- 3 ms for a list of 8192 small
    arrays, 10 us for the mirror."""

    def __init__(self, rows=()):
        list.__init__(self)
        self._buf, self._n = None, 0
        self.extend(rows)

    def append(self, x):
        x = np.asarray(x, dtype=float)
        if self._buf is None:
            self._buf = np.empty((16,) + x.shape)
        if x.shape != self._buf.shape[1:]:
            raise ValueError('trace rows must have the same shape')
        if self._n == len(self._buf):
            grown = np.empty((2 * len(self._buf),) + self._buf.shape[1:])
            grown[:self._n] = self._buf[:self._n]
            self._buf = grown
        self._buf[self._n] = x
        self._n += 1
        list.append(self, self._buf[self._n - 1])

    def extend(self, rows):
        for x in rows:
            self.append(x)

    def __array__(self, dtype=None, copy=None):
        out = self._buf[:self._n] if self._buf is not None else np.empty((0,))
        return out if dtype is None else out.astype(dtype)

    def __reduce__(self):
        return (_Rows, (list(np.array(self)),))

    # every OTHER way a list can change (the loop itself only appends): apply it, then rebuild the mirror from the
    # list's contents, so that np.array(trace.x) -- what the policies and recommenders read -- never goes stale
    def _rebuild(self):
        rows = [np.array(r, dtype=float) for r in list.__iter__(self)]
        list.clear(self)
        self._buf, self._n = None, 0
        for r in rows:
            self.append(r)

    def _mutator(name):                      # noqa: N805 (class-body helper)
        plain = getattr(list, name)

        def method(self, *args, **kwargs):
            out = plain(self, *args, **kwargs)
            self._rebuild()
            return self if name in ('__iadd__', '__imul__') else out
        method.__name__ = name
        return method

    for _name in ('__setitem__', '__delitem__', 'insert', 'pop', 'remove', 'clear', 'sort', 'reverse', '__iadd__',
                  '__imul__'):
        locals()[_name] = _mutator(_name)
    del _name, _mutator


# SPMD is OPT-IN (solve_bayesopt(..., spmd=True) / init_model(..., spmd=True)): a process that merely has
# torch.distributed initialised -- e.g. tuning something inside a data-parallel training job, a different objective per
# rank -- runs its own, independent loop and is never pulled into a collective.
_SPMD = {'on': False, 'group': None}


def _spmd_rank():
    from . import dist as pdist
    d = pdist._dist()
    return d.get_rank(_SPMD['group']) if d is not None else 0


def _spmd_group(spmd):
    """None / False -> (False, None); True -> the default process group; a ProcessGroup -> that group."""
    if spmd is None or spmd is False:
        return False, None
    from . import dist as pdist
    if pdist._dist() is None:
        raise RuntimeError('spmd=%r needs an initialised torch.distributed process group' % (spmd,))
    return True, (None if spmd is True else spmd)


# ----------------------------------------------------------------------------------------------------
# model bootstrap
# ----------------------------------------------------------------------------------------------------
def _heuristic_hypers(y, bounds):
    """Initial hyper-parameters from the design values (reference heuristics, bayesopt.py:98-102)."""
    spread = (max(y) - min(y)) if len(y) > 1 else 1.0
    return dict(sn2=1e-6,
                rho=spread if spread >= 1e-1 else 1.0,
                ell=0.25 * (bounds[:, 1] - bounds[:, 0]),
                bias=float(np.mean(y)) if len(y) else 0.0)


def init_model(f, bounds, ninit=None, design='latin', log=None, rng=None, kernel='se', devices=None, spmd=None):
    """Evaluate an initial design (resumable through `log`) and return a GP fitted to it.  `devices`: a list of
    GPUs for a one-process multi-device model (models.ShardedGP).  `spmd`: see solve_bayesopt."""
    from . import models
    from .dist import spmd_objective, broadcast_seed
    on, group = _spmd_group(spmd)
    if on and not getattr(f, 'spmd', False):
        rng = broadcast_seed(rng, group)
        f = spmd_objective(f, group)             # SPMD runs: rank 0 evaluates, everyone receives (x, y)
    saved = dict(_SPMD)
    if on:
        _SPMD.update(on=True, group=group)
    try:
        return _init_model(f, bounds, ninit, design, log, rng, kernel, devices)
    finally:
        _SPMD.update(saved)


def _init_model(f, bounds, ninit, design, log, rng, kernel, devices):
    from . import models
    rng = rstate(rng)
    bounds = np.array(bounds, dtype=float, ndmin=2)
    stored_model, trace = safe_load(log)
    if stored_model is not None:
        return stored_model

    if not trace.x:                      # fresh start: lay out the design, values still unknown
        npts = 3 * len(bounds) if ninit is None else ninit
        trace.x.extend(getattr(inits, 'init_' + design)(bounds, npts, rng))
        trace.y.extend([np.nan] * npts)
    exchange = getattr(f, 'exchange', None)
    for k, point in enumerate(trace.x):  # (re)evaluate whatever is still missing, saving as we go
        if np.isnan(trace.y[k]):
            if exchange is not None:     # SPMD: the point rank 0 evaluated is the point every rank records
                trace.x[k], trace.y[k] = exchange(point)
            else:
                trace.y[k] = f(point)
        safe_dump(None, trace, filename=log)

    hyp = _heuristic_hypers(trace.y, bounds)
    gp = models.make_gp(hyp['sn2'], hyp['rho'], hyp['ell'], hyp['bias'], kernel=kernel, devices=devices)
    gp.params['like.sn2'].set_prior('horseshoe', 0.1)
    gp.params['kern.rho'].set_prior('lognormal', np.log(hyp['rho']), 1.0)
    gp.params['kern.ell'].set_prior('uniform', hyp['ell'] / 100, hyp['ell'] * 10)
    gp.params['mean.bias'].set_prior('normal', hyp['bias'], hyp['rho'])
    gp.add_data(trace.x, trace.y)
    model = models.MCMC(gp, n=10, burn=100, rng=rng)     # hyper-parameter marginalisation, as the reference
    safe_dump(model, trace, filename=log)
    return model


# ----------------------------------------------------------------------------------------------------
# plugin resolution
# ----------------------------------------------------------------------------------------------------
def _lookup(name, module, prefix):
    """Find `name` among module.__all__, ignoring `prefix` and case of the exported names."""
    for exported in module.__all__:
        key = exported[len(prefix):] if exported.startswith(prefix) else exported
        if key.lower() == name:
            return getattr(module, exported)
    raise ValueError('invalid component: {!s}'.format(name))


def get_component(value, module, rng, lstrip=''):
    """
    Turn a component spec into a callable.  A spec is a name, a callable, or a pair (name-or-callable,
    kwargs).  Rules (reference bayesopt.py:125-176): names match `module.__all__` case-insensitively after
    stripping `lstrip`; kwargs must be a subset of the callable's defaulted arguments other than `rng`;
    `rng` is bound iff the callable has an argument of that name.  Violations raise ValueError.
    """
    extra = {}
    if isinstance(value, (list, tuple)):
        if len(value) != 2:
            raise ValueError('invalid component: {!r}'.format(value))
        value, extra = value
        try:
            extra = dict(extra)
        except (ValueError, TypeError):
            raise ValueError('invalid component: {!r}'.format(value))

    target = value if callable(value) else _lookup(value, module, lstrip)

    spec = inspect.getfullargspec(target)
    ndef = len(spec.defaults or ())
    tunable = set(spec.args[len(spec.args) - ndef:]) - {'rng'}
    unknown = [k for k in extra if k not in tunable]
    if unknown:
        raise ValueError('unknown arguments for {:s}: {:s}'.format(
            getattr(target, '__name__', repr(target)), ', '.join(extra)))
    if 'rng' in spec.args:
        extra['rng'] = rng
    return functools.partial(target, **extra) if extra else target


# ----------------------------------------------------------------------------------------------------
# progress line
# ----------------------------------------------------------------------------------------------------
int2str = '{:03d}'.format
float2str = '{: .3f}'.format


def array2str(a):
    return np.array2string(np.asarray(a), formatter={'float': float2str, 'int': int2str})


def _report(i, x, y, xbest):
    print('i={:s}, x={:s}, y={:s}, xbest={:s}'.format(int2str(i), array2str(x), float2str(y),
                                                     array2str(xbest)))


# ----------------------------------------------------------------------------------------------------
# the meta solver
# ----------------------------------------------------------------------------------------------------
def _bo_step(model, trace, objective, bounds, policy, solver, recommender):
    """One iteration: choose, evaluate, absorb, recommend.  Mutates `model` and `trace`."""
    index = policy(model, bounds, trace.x)          # acquisition closure over a model copy
    if _SPMD['on'] and hasattr(index, 'topk'):      # SPMD: every rank sweeps its slice of the grid, ONE all-gather
        from .dist import ShardedIndex
        if not isinstance(index, ShardedIndex):
            index = ShardedIndex(index, _SPMD['group'])
    x, _ = solver(index, bounds)                    # grid sweep + top-k + refinement
    del index                                       # drop the policy's model copy: add_data below may then
    announce = getattr(model, 'anticipate', None)   # extend the factorisation in place instead of refitting;
    if announce is not None:                        # device models start the value-independent part of that
        announce(x)                                 # update now, while the black box is being evaluated
    exchange = getattr(objective, 'exchange', None)
    if exchange is not None:                        # SPMD: rank 0's query point and value, on every rank
        x, y = exchange(x)
    else:
        y = objective(x)
    model.add_data(x, y)                            # refit
    xbest = recommender(model, bounds, trace.x)     # NB: trace.x does not contain x yet (as in the reference)
    trace.x.append(x)
    trace.y.append(y)
    trace.xbest.append(xbest)
    return x, y, xbest


def solve_bayesopt(objective, bounds, model=None, niter=100, policy='ei', solver='lbfgs',
                   recommender='latent', ninit=None, verbose=False, log=None, rng=None, spmd=None):
    """
    Maximise `objective` over the box `bounds` ((d,2) array-like) by Bayesian optimisation.

    `policy`, `solver`, `recommender`: a name, a callable, or (name-or-callable, kwargs); see
    `get_component`.  `model`: any object with the model protocol (copy / add_data / predict /
    get_improvement / get_tail / sample_f); default: a `pybo_amd.models.GP` from `init_model`.
    Returns (xbest, model, Info(x, y, xbest)) with the Info fields as arrays (np.array of the trace columns; inside the
    loop the columns x / xbest are `_Rows`: lists that mirror their rows in one array).

    `spmd` (not in the reference, which is single-process): None / False (default) -- this process runs its own loop,
    whatever torch.distributed state it lives in.  True, or a torch.distributed ProcessGroup: EVERY rank of the group
    calls solve_bayesopt with the same arguments (one rank per GPU); rank 0 evaluates the objective and broadcasts the
    query point and the value (the reference evaluates once per iteration, bayesopt.py:268), `rng=None` is replaced by
    one broadcast seed, the solver's grid stage is sharded over the ranks (pybo_amd.dist.ShardedIndex), and rank 0
    writes the checkpoints: the replicated models stay bitwise equal.
    """
    on, group = _spmd_group(spmd)
    saved = dict(_SPMD)
    if on:
        from .dist import spmd_objective, broadcast_seed
        rng = broadcast_seed(rng, group)
        objective = spmd_objective(objective, group)
        _SPMD.update(on=True, group=group)
    try:
        return _solve_bayesopt(objective, bounds, model, niter, policy, solver, recommender, ninit, verbose, log, rng,
                               spmd if on else None)
    finally:
        _SPMD.update(saved)


def _solve_bayesopt(objective, bounds, model, niter, policy, solver, recommender, ninit, verbose, log, rng, spmd):
    rng = rstate(rng)
    bounds = np.array(bounds, dtype=float, ndmin=2)
    policy = get_component(policy, policies, rng)
    solver = get_component(solver, solvers, rng, lstrip='solve_')
    recommender = get_component(recommender, recommenders, rng, lstrip='best_')

    resumed, trace = safe_load(log)
    if resumed is not None:
        model = resumed
    elif model is None:
        model = init_model(objective, bounds, ninit, log=log, rng=rng, spmd=spmd)   # trace stays as loaded above
    else:
        model = model.copy()             # never mutate the caller's model
    trace = Info(_Rows(trace.x), list(trace.y), _Rows(trace.xbest))

    if not trace.x:                      # seed the trace with the centre of the box
        x0 = inits.init_middle(bounds)[0]
        y0 = objective(x0)               # (the box centre is the same point on every rank)
        trace.x.append(x0)
        trace.y.append(y0)
        model.add_data(x0, y0)
        safe_dump(model, trace, filename=log)

    xbest = trace.xbest[-1] if trace.xbest else None
    for i in range(len(trace.xbest), niter):
        x, y, xbest = _bo_step(model, trace, objective, bounds, policy, solver, recommender)
        safe_dump(model, trace, filename=log)
        if verbose:
            _report(i, x, y, xbest)

    return xbest, model, Info(*(np.array(col) for col in trace))
