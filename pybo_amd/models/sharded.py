"""
One process, several GPUs: a GP whose whole-grid work is sharded over P device handles.

pybo is single-process: `solve_bayesopt` evaluates the objective ONCE per iteration
(/root/reference/pybo/bayesopt.py:268) and asks the model for everything else.  `ShardedGP` keeps that shape on a
multi-GPU node -- it is the drop-in under `solve_bayesopt(model=make_gp(..., devices=[0, 1, ..., 7]))`:

  * one `pybo_amd.models.GP` replica (= one gpx handle) per entry of `devices`, driven from host threads (ctypes
    releases the GIL; handles are independent, include/gpx.h threading note);
  * `add_data` fits / appends on EVERY replica concurrently -- the fit is deterministic, so the replicas hold
    bitwise-identical factors (tests/test_gpu_multirank.py) and nothing has to be communicated;
  * grid-sized calls -- `index.topk(xgrid, k)` of the solver (pybo/solvers/lbfgs.py:50-51), `predict` /
    `get_improvement` / `get_tail` on many rows -- split the rows into contiguous shards, one per replica, run them
    concurrently and merge: top-k by the deterministic rule of pybo_amd.dist.merge_topk (value descending, then
    global index ascending), row results by concatenation.  Per-candidate results do not depend on how the grid is
    chunked (bit-identical, tested), so the sharded answers equal the single-handle ones bit for bit;
  * small calls (the ~10-row gradient batches of the L-BFGS refinement, the closed forms at the data) run on
    replica 0 alone.

`devices` may name one GPU several times ([0, 0]): P handles on one device -- how the 1-GPU test box proves the path.
The multi-PROCESS layout (one rank per GPU under torch.distributed, pybo_amd.dist) stays available; this class is the
one that needs no launcher.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .._lib import DeviceGrid, ShardedDeviceGrid
from ..dist import merge_topk, shard_bounds
from ..utils import rstate
from .gp import GP, RFFSampleDevice

__all__ = ['ShardedGP']

_POOL = ThreadPoolExecutor(max_workers=16, thread_name_prefix='pybo_amd-shard')
SPLIT_ROWS = 8192      # row batches below this stay on replica 0 (a launch per device would cost more than it saves)


def _run(jobs):
    """Run the thunks concurrently (one host thread per replica); results in order; the first error is raised
    after every thread has finished (no replica is left mid-call)."""
    if len(jobs) == 1:
        return [jobs[0]()]
    futs = [_POOL.submit(j) for j in jobs]
    out, err = [], None
    for f in futs:
        try:
            out.append(f.result())
        except BaseException as exc:      # noqa: collected, re-raised below
            out.append(None)
            err = err or exc
    if err is not None:
        raise err
    return out


def _sharded_topk(calls, grid, k, devices=None):
    """calls[p](sub_grid, k) -> (values, local indices) on shard p of `grid`; returns the merged global top-k.
    `devices[p]`: the GPU replica p computes on.  A ShardedDeviceGrid is used in place only when shard p LIVES on
    devices[p] (a grid built for [0, 1] under a model on [1, 0] or [2, 3] would hand a replica a pointer into another
    GPU's memory); any other layout goes through the host."""
    P = len(calls)
    M = len(grid)
    in_place = isinstance(grid, ShardedDeviceGrid) and len(grid.shards) == P and devices is not None and \
        all(g is None or int(g.device) == int(dv) for (_, _, g), dv in zip(grid.shards, devices))
    if in_place:
        parts = [(lo, hi, g) for lo, hi, g in grid.shards]
    else:
        if isinstance(grid, (DeviceGrid, ShardedDeviceGrid)):
            grid = np.asarray(grid)          # resident elsewhere / laid out differently: through the host
        bounds = [shard_bounds(M, p, P) for p in range(P)]
        parts = [(lo, hi, grid[lo:hi]) for lo, hi in bounds]
    jobs = []
    for p, (lo, hi, sub) in enumerate(parts):
        if hi > lo:
            jobs.append(lambda p=p, sub=sub, n=hi - lo: calls[p](sub, min(int(k), n)))
        else:
            jobs.append(lambda: (np.empty(0), np.empty(0, dtype=np.int64)))
    res = _run(jobs)
    vals = np.concatenate([np.asarray(r[0], dtype=float) for r in res])
    idx = np.concatenate([np.where(np.asarray(r[1], dtype=np.int64) >= 0, np.asarray(r[1], dtype=np.int64) + lo, -1)
                          for r, (lo, _, _) in zip(res, parts)])
    return merge_topk(vals, idx, int(k))


class ShardedRFFSample(object):
    """One Thompson draw (pybo/policies/simple.py:48) evaluated by every replica on its shard of the grid."""

    def __init__(self, samples, devices=None):
        self._samples = samples
        self._devices = devices

    def get(self, X, grad=False):
        return self._samples[0].get(X, grad)

    def topk(self, xgrid, k):
        return _sharded_topk([s.topk for s in self._samples], xgrid, k, self._devices)

    __call__ = get


class ShardedGP(object):
    def __init__(self, sn2, rho, ell, bias=0.0, kernel='se', devices=(0,)):
        devices = [int(dv) for dv in devices]
        if not devices:
            raise ValueError('ShardedGP needs at least one device')
        self.devices = devices
        self._reps = [GP(sn2, rho, ell, bias, kernel, device=dv) for dv in devices]

    # -- hyper-parameters: one set of values, mirrored on every replica ---------------------------------------
    def _get(self, name):
        return getattr(self._reps[0], name)

    def _set(self, name, value):
        for r in self._reps:
            setattr(r, name, np.array(value, dtype=float, ndmin=1) if name == 'ell' else value)

    sn2 = property(lambda self: self._get('sn2'), lambda self, v: self._set('sn2', float(v)))
    rho = property(lambda self: self._get('rho'), lambda self, v: self._set('rho', float(v)))
    bias = property(lambda self: self._get('bias'), lambda self, v: self._set('bias', float(v)))
    ell = property(lambda self: self._get('ell'), lambda self, v: self._set('ell', v))
    kernel = property(lambda self: self._get('kernel'))

    @property
    def params(self):
        return self._reps[0].params

    @property
    def ndata(self):
        return self._reps[0].ndata

    @property
    def data(self):
        return self._reps[0].data

    @property
    def replicas(self):
        return list(self._reps)

    def copy(self):
        new = ShardedGP.__new__(ShardedGP)
        new.devices = list(self.devices)
        new._reps = [r.copy() for r in self._reps]       # cheap: every copy shares its replica's device state
        return new

    def __getstate__(self):
        return dict(devices=self.devices, gp=self._reps[0].__getstate__())

    def __setstate__(self, st):
        self.devices = list(st['devices'])
        self._reps = []
        for dv in self.devices:
            g = GP.__new__(GP)
            g.__setstate__(dict(st['gp'], device=dv))
            self._reps.append(g)

    def hyper_vector(self):
        return self._reps[0].hyper_vector()

    def set_hyper_vector(self, theta):
        for r in self._reps:
            r.set_hyper_vector(theta)

    def loglikelihood(self):
        return self._reps[0].loglikelihood()

    def loglik_at(self, thetas):
        return self._reps[0].loglik_at(thetas)

    # -- protocol -----------------------------------------------------------------------------------------------
    def add_data(self, X, Y):
        """Every replica absorbs the observation(s) concurrently: the same deterministic fit / rank-1 extension on
        every device, no broadcast of factors ('every rank factorises redundantly')."""
        _run([lambda r=r: r.add_data(X, Y) for r in self._reps])

    def anticipate(self, x):
        """Announce the next query point to every replica (see GP.anticipate)."""
        return any(_run([lambda r=r: r.anticipate(x) for r in self._reps]))

    def _rows(self, call, X):
        """`call(replica, rows)` over contiguous row shards; outputs (arrays or tuples of arrays) concatenated."""
        X = np.array(X, ndmin=2, dtype=float)
        P = len(self._reps)
        if P == 1 or len(X) < SPLIT_ROWS:
            return call(self._reps[0], X)
        bounds = [shard_bounds(len(X), p, P) for p in range(P)]
        res = _run([lambda p=p, lo=lo, hi=hi: call(self._reps[p], X[lo:hi]) for p, (lo, hi) in enumerate(bounds)])
        if isinstance(res[0], tuple):
            return tuple(np.concatenate([r[i] for r in res]) for i in range(len(res[0])))
        return np.concatenate(res)

    def predict(self, X, grad=False):
        return self._rows(lambda r, Z: r.predict(Z, grad), X)

    def predict_mean(self, X, grad=False):
        return self._reps[0].predict_mean(X, grad)

    def mean_topk(self, xgrid, k):
        if not isinstance(xgrid, (DeviceGrid, ShardedDeviceGrid)):
            xgrid = np.array(xgrid, ndmin=2, dtype=float)
            if self.ndata and self._reps[0]._data_rows(xgrid) is not None:
                return self._reps[0].mean_topk(xgrid, k)          # closed form at the data: nothing to shard
        return self.acq_topk('mean', None, xgrid, k)

    def get_improvement(self, target, X, grad=False):
        return self._rows(lambda r, Z: r.get_improvement(target, Z, grad), X)

    def get_tail(self, target, X, grad=False):
        return self._rows(lambda r, Z: r.get_tail(target, Z, grad), X)

    def acq_values(self, kind, param, xgrid):
        return self._rows(lambda r, Z: r.acq_values(kind, param, Z), xgrid)

    def posterior_mean_at_data(self):
        return self._reps[0].posterior_mean_at_data()

    def acq_topk(self, kind, param, xgrid, k):
        """The solver's grid stage (pybo/solvers/lbfgs.py:50-51) over all devices: replica p sweeps shard p of the grid
        (a `ShardedDeviceGrid` built for the same device list is already laid out that way and stays in HBM; a host
        grid is sliced and uploaded per shard), the P x k (value, global index) pairs are merged on the host."""
        return _sharded_topk([lambda g, kk, r=r: r.acq_topk(kind, param, g, kk) for r in self._reps], xgrid, k,
                             self.devices)

    def sample_f(self, n, rng=None):
        """ONE posterior function sample (the host draws and the weight posterior come from replica 0), evaluated by
        every replica on its shard of the grid."""
        first = self._reps[0].sample_f(n, rstate(rng))
        rest = [RFFSampleDevice(r, first.W, first.b, first.theta) for r in self._reps[1:]]
        for s in rest:
            s.bias = first.bias
        return ShardedRFFSample([first] + rest, self.devices)
