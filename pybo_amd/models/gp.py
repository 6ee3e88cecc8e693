"""
Device-backed exact GP with the duck-typed protocol pybo expects from `reggie` models
(call sites: /root/reference/pybo/bayesopt.py:105-115,258,269; pybo/policies/simple.py:20-64;
pybo/recommenders.py:22-34):

    make_gp(sn2, rho, ell, bias)   copy()   add_data(X, Y)   predict(X, grad=False)
    get_improvement(target, X, grad)   get_tail(target, X, grad)   sample_f(n, rng).get(X, grad)
    params[name].set_prior(kind, *args)   picklable

All GP arithmetic runs in libgpx.so (HIP, gfx950) through pybo_amd._lib.Engine; this file is host
logic only: data bookkeeping, copy-on-write sharing of the device state between `copy()`s, pickling
(hyper-parameters + data only, device state rebuilt lazily), and the closed-form elementwise EI/PI
gradient chain rule on top of device-computed (mu, s2, dmu, ds2).
There is NO CPU fallback: without the library / a GPU, the first device call raises.
"""
import threading
import weakref

import numpy as np
from scipy.special import erfc

from .. import _lib
from .._lib import DeviceGrid
from ..utils import rstate

__all__ = ['GP', 'make_gp']

_KERNELS = ('se', 'matern5', 'matern3', 'matern1')
_ENGINE_POOL = []      # engines whose last owner died; reused so steady-state BO never re-allocates
_POOL_LOCK = threading.RLock()      # models are created and dropped from several host threads (models/sharded.py)
_POOL_SMALL_N = 1024   # handles whose factor buffers are ALLOCATED for at most ~this many rows count as small (<= ~40 MB)


def _is_small(engine):
    """Pool class of a handle by what it keeps allocated (gpx_capacity), not by its last fit: buffers never shrink,
    so a handle that once held a large model stays large however small its current one is."""
    return engine.capacity() <= _POOL_SMALL_N + 256


class _DeviceState(object):
    """A fitted engine shared by a model and its copies (copy-on-write in GP._own_state)."""

    def __init__(self, device, ndata=0):
        self.engine = None
        self.key = None            # hyper-parameters of the fit the engine holds
        self.cache_grid = None     # the DeviceGrid whose sweep sums the engine keeps (warm BO step)
        # a pooled handle of this device, preferably of the model's size class (a small model that borrows a
        # multi-GB handle keeps it out of reach of the next large model, which then allocates afresh)
        with _POOL_LOCK:
            _ENGINE_POOL[:] = [e for e in _ENGINE_POOL if e._h]          # never hand out a closed handle
            want_small = ndata <= _POOL_SMALL_N
            mine = [e for e in _ENGINE_POOL if e.device == device]       # other devices' handles wait for their models
            pick = next((e for e in reversed(mine) if _is_small(e) == want_small), mine[-1] if mine else None)
            if pick is not None:
                _ENGINE_POOL.remove(pick)
                self.engine = pick
        if self.engine is None:
            self.engine = _lib.Engine(device)
        self.nrefs = 1

    def release(self):
        self.nrefs -= 1
        if self.nrefs == 0 and self.engine is not None:
            # keep the handle for the next model: creating and destroying one costs ~8 ms each (HSA queue,
            # allocations), which was 80 % of a default solve_bayesopt run -- the hyper-parameter sampler turns over
            # ~30 member / proposal models per iteration.  Handles of small models are cheap to keep (a few MB);
            # at most 4 large ones (their factor and sweep buffers stay allocated) wait in the pool.
            with _POOL_LOCK:
                room = False
                if self.engine._h:
                    small = _is_small(self.engine)
                    room = (sum(1 for e in _ENGINE_POOL if _is_small(e)) < 64) if small \
                        else (sum(1 for e in _ENGINE_POOL if not _is_small(e)) < 4)
                if room:
                    try:
                        self.engine.set_option('sweep_cache', -1)       # the next owner starts without a cache
                    except Exception:
                        pass
                    _ENGINE_POOL.append(self.engine)
                self.engine = None
                self.cache_grid = None


class Param(object):
    """A named hyper-parameter with an optional prior record (priors are stored, not sampled)."""

    def __init__(self, owner, attr):
        # weak back-reference: a strong one would make every GP part of a reference cycle, and the
        # cyclic collector runs GP.__del__ and Engine.__del__ in arbitrary order
        self._owner, self._attr = weakref.proxy(owner), attr
        self.prior = None

    @property
    def value(self):
        return getattr(self._owner, self._attr)

    def set_prior(self, kind, *args):
        self.prior = (kind,) + tuple(np.array(a, dtype=float) for a in args)


class RFFSampleDevice(object):
    """One RFF posterior function sample; `.get(X, grad=False)` evaluates on the device."""

    def __init__(self, model, W, b, theta):
        # the sample is a fixed function: the mean offset is captured now, later hyper-parameter changes of
        # the model do not alter it
        self._model, self.W, self.b, self.theta, self.bias = model, W, b, theta, float(model.bias)

    def _eng(self):
        # evaluating the feature expansion needs a device handle but no fit (a prior sample of a model
        # without data is a valid Thompson index)
        return self._model._any_engine()

    def get(self, X, grad=False):
        X = np.array(X, ndmin=2, dtype=float)
        eng = self._eng()
        if grad:
            return eng.rff_eval_grad(self.W, self.b, self.theta, self.bias, X)
        out = eng.rff_sweep(self.W[None], self.b[None], self.theta[None], self.bias, X, k=0)
        return out['vals'][0]

    def topk(self, xgrid, k):
        eng = self._eng()
        if isinstance(xgrid, DeviceGrid) and int(xgrid.device) != int(eng.device):
            xgrid = np.asarray(xgrid)                # resident on ANOTHER GPU: its pointer means nothing here
        if isinstance(xgrid, DeviceGrid):            # grid already resident in HBM
            tv, ti = eng.rff_sweep_dev(self.W[None], self.b[None], self.theta[None], self.bias,
                                       xgrid.ptr, len(xgrid), int(k))
            return tv[0], ti[0]
        out = eng.rff_sweep(self.W[None], self.b[None], self.theta[None], self.bias, xgrid, k=int(k),
                            want_all=False)
        return out['top_val'][0], out['top_idx'][0]

    __call__ = get


class GP(object):
    APPEND_MAX = 16     # add_data with at most this many new rows extends the factorisation in place
    MEAN_DIRECT_ROWS = 256      # predict_mean of at most this many points: k(x, X).alpha, 16 points per device call

    def __init__(self, sn2, rho, ell, bias=0.0, kernel='se', device=0):
        if kernel not in _KERNELS:
            raise ValueError('unknown kernel {!r}; choose from {}'.format(kernel, _KERNELS))
        self.sn2 = float(sn2)
        self.rho = float(rho)
        self.ell = np.array(ell, dtype=float, ndmin=1)
        self.bias = float(bias)
        self.kernel = kernel
        self.device = int(device)
        self._X = np.empty((0, len(self.ell)))
        self._Y = np.empty(0)
        self._state = None          # _DeviceState, shared with copies
        self._fitted = False
        self.params = {'like.sn2': Param(self, 'sn2'), 'kern.rho': Param(self, 'rho'),
                       'kern.ell': Param(self, 'ell'), 'mean.bias': Param(self, 'bias')}

    # -- bookkeeping ---------------------------------------------------------------------------
    @property
    def ndata(self):
        return len(self._X)

    @property
    def data(self):
        return self._X, self._Y

    def __del__(self):
        st = getattr(self, '_state', None)
        if st is not None:
            try:
                st.release()
            except Exception:
                pass

    def __getstate__(self):
        return dict(sn2=self.sn2, rho=self.rho, ell=self.ell, bias=self.bias, kernel=self.kernel,
                    device=self.device, X=self._X, Y=self._Y,
                    priors={k: p.prior for k, p in self.params.items()})

    def __setstate__(self, st):
        self.__init__(st['sn2'], st['rho'], st['ell'], st['bias'], st['kernel'], st['device'])
        self._X, self._Y = st['X'], st['Y']
        for k, pr in st['priors'].items():
            self.params[k].prior = pr

    def copy(self):
        """Cheap: hyper-parameters and host data are copied, the fitted device state is shared."""
        new = GP(self.sn2, self.rho, self.ell.copy(), self.bias, self.kernel, self.device)
        new._X, new._Y = self._X, self._Y          # arrays are replaced, never mutated in place
        for k, p in self.params.items():
            new.params[k].prior = p.prior
        if self.ndata > 0 and (not self._fitted or self._stale()):
            # a model that has data but no current fit (just unpickled, or its hyper-parameters were assigned) is
            # fitted HERE, so that the copy shares the fit instead of paying for its own and leaving this model
            # unfitted (every policy call starts with model.copy(), pybo/policies/simple.py:20,34,57)
            try:
                self._engine()
            except np.linalg.LinAlgError:
                pass                               # surfaces where the copy is first used
        if self._state is not None and self._fitted:
            self._state.nrefs += 1
            new._state, new._fitted = self._state, True
        return new

    def _own_state(self):
        """A device state this model may overwrite (fresh one if the current is shared)."""
        if self._state is not None and self._state.nrefs > 1:
            self._state.release()
            self._state = None
        if self._state is None:
            self._state = _DeviceState(self.device, self.ndata)
        return self._state

    def _hyper_key(self):
        return (self.sn2, self.rho, self.bias, self.kernel, tuple(np.ravel(self.ell).tolist()))

    def _stale(self):
        """True when the device state was fitted with other hyper-parameters than the model now carries
        (the attributes sn2 / rho / ell / bias may be assigned directly)."""
        return self._fitted and self._state is not None and self._state.key != self._hyper_key()

    def _engine(self):
        if self.ndata == 0:
            raise RuntimeError('the model has no data yet')
        if self._stale():
            self._fitted = False
        if not self._fitted:
            st = self._own_state()
            st.engine.fit(self._X, self._Y, self.kernel, self.ell, self.rho, self.sn2, self.bias)
            st.key = self._hyper_key()
            self._fitted = True
        return self._state.engine

    def _any_engine(self):
        """A device handle for work that needs no fit (RFF feature evaluation): the fitted engine when there is
        data, otherwise this model's (possibly still empty) device state."""
        if self.ndata > 0:
            return self._engine()
        if self._state is None:
            self._state = _DeviceState(self.device)
        return self._state.engine

    # -- protocol ------------------------------------------------------------------------------
    def add_data(self, X, Y):
        d = len(self.ell)
        X = np.array(X, dtype=float)
        X = X.reshape(-1, d)
        Y = np.array(Y, dtype=float).reshape(-1)
        if len(X) != len(Y):
            raise ValueError('X and Y must have the same number of rows')
        can_append = (self._fitted and not self._stale() and self._state is not None and self._state.nrefs == 1
                      and 0 < len(X) <= self.APPEND_MAX)
        self._X = np.vstack([self._X, X])
        self._Y = np.hstack([self._Y, Y])
        if can_append:
            # sole owner of a fitted device state: rank-1 extension per new row (O(N^2)) instead of the
            # O(N^3) refit (the library grows the factor by a block when the padding of the last one is used up)
            eng = self._state.engine
            done = 0
            try:
                for xr, yr in zip(X, Y):
                    if not eng.append(xr, yr):
                        break
                    done += 1
            except Exception:
                self._fitted = False        # device and host data now disagree: force a refit next time
                raise
            if done == len(X):
                return
        self._fitted = False
        self._engine()              # refit now: a non-PD Gram matrix must surface here (LinAlgError)

    def anticipate(self, x):
        """The loop knows the next query point before it knows its value (`x, _ = solver(...)`; `y = objective(x)`,
        pybo/bayesopt.py:265-268): tell the device now.  Everything of the coming `add_data(x, y)` that does not depend
        on y -- k(X, x), the triangular passes, the N*M covariance evaluations that keep the warm sweep cache current --
        runs while the objective is being evaluated (gpx_append_begin); `add_data` with the same x then only finishes.
        Returns True when work was started.  Never required: `add_data` gives bit-identical results without it."""
        st = self._state
        if not (self._fitted and st is not None and st.nrefs == 1 and st.cache_grid is not None) or self._stale():
            return False
        try:
            return bool(st.engine.append_begin(np.array(x, dtype=float).reshape(-1)))
        except Exception:
            return False

    def predict(self, X, grad=False):
        X = np.array(X, ndmin=2, dtype=float)
        if X.shape[0] == 0:                      # empty in, empty out (numpy semantics; no device call)
            e1, e2 = np.zeros(0), np.zeros((0, X.shape[1]))
            return (e1, e1.copy(), e2, e2.copy()) if grad else (e1, e1.copy())
        if self.ndata == 0:
            M, d = X.shape
            mu, s2 = np.full(M, self.bias), np.full(M, self.rho)
            return (mu, s2, np.zeros((M, d)), np.zeros((M, d))) if grad else (mu, s2)
        eng = self._engine()
        if not grad:
            rows = self._data_rows(X)
            if rows is not None:
                # posterior moments AT the training inputs have closed forms: mu = y - sn2*alpha,
                # s2 = sn2 - sn2^2 [K^-1]_ii (two O(N^2) passes) -- a sweep over X_obs would be an N x N x N
                # product, 3x the Cholesky.  EI / PI ask for it once per policy call (simple.py:21,35), the
                # recommenders once per iteration (recommenders.py:22-34).
                return eng.mean_at_obs()[0][rows], eng.var_at_obs()[rows]
        return eng.predict(X, grad=grad)

    def _data_rows(self, X):
        """Row range of `X` in the model's data when X is a contiguous run of the data rows, else None.  The trace of
        a BO run is one: pybo/bayesopt.py:243-259 puts the initial design in the model but not in the trace, and the
        recommender sees the trace BEFORE the newest point is appended to it (:270-271) -- so X is the data, the data
        without its head, or without its last row."""
        n, N = len(X), len(self._X)
        if not 0 < n <= N or X.shape[1] != self._X.shape[1]:
            return None
        for o in (N - n, max(N - n - 1, 0), 0):              # the usual offsets first
            if np.array_equal(X, self._X[o:o + n]):
                return slice(o, o + n)
        for o in np.flatnonzero(np.all(self._X[:N - n + 1] == X[0], axis=1)):
            if np.array_equal(X, self._X[o:o + n]):
                return slice(int(o), int(o) + n)
        return None

    def predict_mean(self, X, grad=False):
        """Posterior mean only: `model.predict(X)[0]` (with grad: `model.predict(X, True)[0::2]`) without the variance
        nobody reads (EI / PI targets, both recommenders).  Closed form at the model's own data; a few points or a
        gradient: `k(x, X).alpha` on the device, no pass over the factor's inverse; a large batch: the sweep."""
        X = np.array(X, ndmin=2, dtype=float)
        if X.shape[0] == 0:
            return (np.zeros(0), np.zeros((0, len(self.ell)))) if grad else np.zeros(0)
        if self.ndata == 0:
            mu = np.full(len(X), self.bias)
            return (mu, np.zeros(X.shape)) if grad else mu
        if grad:
            return self._engine().predict_mean(X, True)
        rows = self._data_rows(X)
        if rows is not None:
            return self._engine().mean_at_obs()[0][rows]
        if len(X) <= self.MEAN_DIRECT_ROWS:
            return self._engine().predict_mean(X)
        return self._engine().predict(X)[0]

    def mean_topk(self, xgrid, k):
        """(values, indices) of the k largest posterior means over `xgrid` -- the grid stage of the latent
        recommender (pybo/recommenders.py:22-27 runs solve_lbfgs with xgrid = X_obs): closed form + host ranking
        when the grid is the model's data, the device sweep otherwise."""
        if not isinstance(xgrid, DeviceGrid):
            xgrid = np.array(xgrid, ndmin=2, dtype=float)
            if self.ndata and self._data_rows(xgrid) is not None:
                mu = self.predict_mean(xgrid)
                v = np.where(np.isnan(mu), -np.inf, mu)
                order = np.lexsort((np.arange(len(v)), -v))[:int(k)]
                return mu[order], order
        return self.acq_topk('mean', None, xgrid, k)

    # -- hyper-parameter access (used by the MCMC meta-model) ---------------------------------------------
    def hyper_vector(self):
        """[log sn2, log rho, log ell_1..d, bias]"""
        return np.concatenate([[np.log(self.sn2), np.log(self.rho)], np.log(self.ell), [self.bias]])

    def set_hyper_vector(self, theta):
        theta = np.asarray(theta, dtype=float)
        d = len(self.ell)
        self.sn2, self.rho = float(np.exp(theta[0])), float(np.exp(theta[1]))
        self.ell = np.exp(theta[2:2 + d])
        self.bias = float(theta[2 + d])
        self._fitted = False                     # same data, new hyper-parameters: refit on next use

    def loglikelihood(self):
        """log p(y | X, hyper-parameters) of the current fit, computed on the device."""
        return self._engine().loglik()

    def loglik_at(self, thetas):
        """log p(y | X, theta_b) for the rows of `thetas` ([log sn2, log rho, log ell_1..d, bias]) in ONE batched
        device call (gpx_loglik_batch) on the resident data; the model and its fit are left untouched.  -inf
        where the covariance is not positive definite.  This is what the hyper-parameter sampler evaluates per
        proposal."""
        thetas = np.array(thetas, dtype=float, ndmin=2)
        d = len(self.ell)
        hyp = np.column_stack([np.exp(thetas[:, 0]), np.exp(thetas[:, 1]), np.exp(thetas[:, 2:2 + d]), thetas[:, 2 + d]])
        eng = self._engine()
        # gpx_loglik_batch takes B <= 64 vectors and at most 16e9 bytes of batch buffers (2 B Np^2 doubles):
        # larger requests go in sub-batches (down to one vector at a time; the values do not depend on the grouping)
        Np = -(-self.ndata // 128) * 128
        per = int(max(1, min(64, 16e9 // (16.0 * Np * Np))))
        if len(hyp) <= per:
            return eng.loglik_batch(hyp)
        return np.concatenate([eng.loglik_batch(hyp[i:i + per]) for i in range(0, len(hyp), per)])

    def acq_values(self, kind, param, xgrid):
        """Acquisition values over a whole grid (device sweep, values copied back)."""
        return self._engine().sweep(kind, param, np.array(xgrid, ndmin=2, dtype=float), k=0)['acq']

    def posterior_mean_at_data(self):
        return self._engine().mean_at_obs()[0]

    def _acq(self, kind, target, X, grad):
        X = np.array(X, ndmin=2, dtype=float)
        if self.ndata == 0:
            raise RuntimeError('the model has no data yet')
        if X.shape[0] == 0:
            return (np.zeros(0), np.zeros((0, X.shape[1]))) if grad else np.zeros(0)
        if not grad:
            return self._engine().sweep(kind, target, X, k=0)['acq']
        mu, s2, dmu, ds2 = self._engine().predict(X, grad=True)
        s = np.sqrt(s2)
        z = (mu - target) / s
        cdf = 0.5 * erfc(-z * 0.70710678118654752440)
        pdf = 0.39894228040143267794 * np.exp(-0.5 * z * z)
        if kind == 'ei':
            val = (mu - target) * cdf + s * pdf
            return val, cdf[:, None] * dmu + (0.5 * pdf / s)[:, None] * ds2
        dz = dmu / s[:, None] - (0.5 * z / s2)[:, None] * ds2
        return cdf, pdf[:, None] * dz

    def get_improvement(self, target, X, grad=False):
        return self._acq('ei', target, X, grad)

    def get_tail(self, target, X, grad=False):
        return self._acq('pi', target, X, grad)

    def acq_topk(self, kind, param, xgrid, k):
        """Whole-grid acquisition + top-k on the device: (values (k,), grid indices (k,)).  `xgrid` is a
        host array (uploaded) or a `DeviceGrid` (already in HBM)."""
        if isinstance(xgrid, DeviceGrid) and int(xgrid.device) != int(self._engine().device):
            xgrid = np.asarray(xgrid)                # resident on ANOTHER GPU: through the host, never its pointer
        if isinstance(xgrid, DeviceGrid):
            # Warm BO step: a grid resident in HBM that this device state has swept before is only RE-SCORED --
            # the per-candidate sums were kept current by every add_data since (gpx_append's rank-1 correction),
            # so an iteration costs O(N M) instead of the O(N^2 M) sweep.  Any refit (new hyper-parameters, a
            # diverging copy) drops the cache and the next call sweeps in full again.
            eng = self._engine()
            st = self._state
            if st.cache_grid is xgrid and eng.sweep_cache_size() == len(xgrid):
                r = eng.sweep_update(kind, param, k=int(k), want_all=False)
                return r['top_val'], r['top_idx']
            eng.set_option('sweep_cache', 1)
            try:
                out = eng.sweep_dev(kind, param, xgrid.ptr, len(xgrid), int(k))
            finally:
                eng.set_option('sweep_cache', 0)      # other sweeps of this engine must not overwrite the cache
            st.cache_grid = xgrid
            return out
        xgrid = np.array(xgrid, ndmin=2, dtype=float)
        out = self._engine().sweep(kind, param, xgrid, k=int(k), want_all=False)
        return out['top_val'], out['top_idx']

    def topk_engine(self):
        """The device handle whose HBM holds the (value, index) pairs of the last `acq_topk`."""
        return self._engine()

    def sample_f(self, n, rng=None):
        """RFF posterior function sample.  Host draws (order fixed: randn(n,d), [chisquare], rand(n),
        randn(n)); the O(N n^2) feature Gram and the n x n weight posterior run on the device (n <= 4096 features; a
        noise-free model solves on the host)."""
        rng = rstate(rng)
        d = len(self.ell)
        W = rng.randn(n, d)
        if self.kernel != 'se':
            nu = {'matern5': 2.5, 'matern3': 1.5, 'matern1': 0.5}[self.kernel]
            u = rng.chisquare(2.0 * nu, size=n)
            W = W * np.sqrt(2.0 * nu / u)[:, None]
        W = W / self.ell
        b = rng.rand(n) * 2.0 * np.pi
        z = rng.randn(n)
        sc = np.sqrt(2.0 * self.rho / n)
        if self.ndata == 0:
            return RFFSampleDevice(self, W, b, sc * z)
        if n <= 4096 and self.sn2 > 0.0:    # feature Gram AND the n x n weight posterior on the device (gpx_rff_posterior)
            theta = self._engine().rff_posterior(W[None], b[None], z[None], sc)[0]
            return RFFSampleDevice(self, W, b, theta)
        A, v = self._engine().rff_gram(W, b)           # wide feature maps / a noise-free model: the n x n solve on the host
        Am = (sc * sc) * A + self.sn2 * np.eye(n)
        L = np.linalg.cholesky(Am)
        mean = np.linalg.solve(L.T, np.linalg.solve(L, sc * v))
        noise = np.sqrt(self.sn2) * np.linalg.solve(L.T, z)
        return RFFSampleDevice(self, W, b, sc * (mean + noise))


def make_gp(sn2, rho, ell, bias=0.0, kernel='se', device=0, devices=None):
    """`reggie.make_gp(sn2, rho, ell, bias)` (pybo/bayesopt.py:105) plus kernel family and device.
    `devices=[0, 1, ...]`: one process, several GPUs -- a `ShardedGP` with one handle per listed device (the fit
    replicated, every grid-sized call sharded over them; models/sharded.py)."""
    if devices is not None and len(devices) > 0:
        from .sharded import ShardedGP
        return ShardedGP(sn2, rho, ell, bias, kernel, devices)
    return GP(sn2, rho, ell, bias, kernel, device)
