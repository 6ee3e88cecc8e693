"""
Hyper-parameter marginalisation by MCMC: the meta-model pybo wraps its default GP in,
`reggie.MCMC(model, n=10, burn=100, rng=rng)` (/root/reference/pybo/bayesopt.py:115), refreshed on every
`add_data`.  It exposes the same model protocol as a single GP (copy / add_data / predict /
get_improvement / get_tail / sample_f), with every quantity averaged over `n` posterior samples of the
hyper-parameters.

reggie's sampler is not available (absent, unpinned), so the algorithm here is this build's own and is
stated in full:
  * state  theta = [log sn2, log rho, log ell_1..d, bias];  target  log p(y | theta) + log prior(theta)
    (+ the log-Jacobian of the log transform), priors as recorded on `model.params` (priors.py);
  * one sample = one slice-sampling update (Neal 2003) along a random direction  scale * N(0, I) with a
    STATE-INDEPENDENT scale (1 for the log-parameters; for the bias the signal std sqrt(rho) of the model the
    ensemble was built from, frozen at construction -- a scale that moved with the chain's own rho would make
    the update irreversible): bracket by stepping out in unit steps (at most 16 expansions, split at random
    between the sides as in Neal's fig. 3), then shrinkage;
  * `burn` updates are discarded at construction, then `n` are kept; `add_data` continues the chain from
    its last state and keeps the next `n` samples (no new burn-in).
Each log-likelihood evaluation is one fit of the member model; for the device model `pybo_amd.models.GP` the
sampler's evaluations go through `loglik_at` -> `gpx_loglik_batch` (batched Gram + Cholesky + forward substitution
on the resident data, the two ends of a stepping-out round in one call), the kept members are fitted with
`gpx_fit`.  This file is host logic only and works with ANY member model that offers
`hyper_vector() / set_hyper_vector(theta) / loglikelihood() / params / copy() / add_data()`, which is how
the CPU tests drive it with the oracle model.
"""
import math

import numpy as np
from scipy.special import erfc

from ..utils import rstate
from .priors import log_prior

__all__ = ['MCMC']


def _log_prior_part(model, theta):
    """log prior density + log-Jacobian of the transformed hyper-parameters (-inf outside the support)."""
    d = len(theta) - 3
    if not np.all(np.isfinite(theta)) or np.any(np.abs(theta[:2 + d]) > 60.0):
        return -np.inf
    sn2, rho, ell, bias = math.exp(theta[0]), math.exp(theta[1]), np.exp(theta[2:2 + d]), float(theta[2 + d])
    pr = model.params
    lp = (log_prior(pr['like.sn2'].prior, sn2) + log_prior(pr['kern.rho'].prior, rho) +
          log_prior(pr['kern.ell'].prior, ell) + log_prior(pr['mean.bias'].prior, bias))
    if not math.isfinite(lp):
        return -np.inf
    return lp + float(np.sum(theta[:2 + d]))      # Jacobian of x = exp(theta)


def _log_targets(model, thetas):
    """log posterior densities of several hyper-parameter states.  A device model evaluates all likelihoods in ONE
    batched call that leaves the model untouched (`loglik_at` -> gpx_loglik_batch: one launch chain and one host
    synchronisation instead of a refit + log-likelihood round trip per state); any other model is refitted per
    state.  The values do not depend on how the states are grouped into calls."""
    out = np.array([_log_prior_part(model, th) for th in thetas], dtype=float)
    live = np.flatnonzero(np.isfinite(out))
    if len(live) == 0:
        return out
    batch = getattr(model, 'loglik_at', None)
    if batch is not None:
        out[live] += batch(np.array([thetas[i] for i in live]))
        return out
    for i in live:
        try:
            model.set_hyper_vector(thetas[i])
            out[i] += model.loglikelihood()
        except np.linalg.LinAlgError:
            out[i] = -np.inf
    return out


def _log_target(model, theta):
    return float(_log_targets(model, [theta])[0])


def _slice_update(logp, theta, lp, rng, scale, max_out=16, logp_many=None, spec=4):
    """One slice-sampling update (Neal 2003, fig. 3 + fig. 5) of `theta` along a random direction
    scale * N(0, I); returns (theta', lp').  `logp` is the log target density (callable), `lp` its value at
    `theta`.  `scale` is a vector that must NOT depend on the current state: the direction has to be drawn
    from the same distribution at theta and at theta' for the update to leave the target invariant.
    The bracket grows by stepping out in unit steps with the total number of expansions capped at `max_out`
    and split at random between the two sides (Neal's J/K rule), then shrinks towards the current point.

    `logp_many` (optional): evaluates several states in one call (a batched device call).  Both phases are then
    SPECULATED without changing the chain: the stepping-out points of a side are lo-1, lo-2, ... whatever the
    densities turn out to be, and a shrinkage candidate only depends on the SIGN of the previous one (t < 0 moves
    the lower end, otherwise the upper one), never on its density -- so the next `spec` points of either phase
    are known in advance, are evaluated together, and the first decisive one is used.  The random stream is
    rewound to exactly what the sequential procedure would have consumed: same states, same draws, same result."""
    direction = scale * rng.randn(len(theta))
    level = lp + np.log(rng.rand())
    r = rng.rand()
    lo, hi = -r, 1.0 - r
    nlo = int(np.floor(max_out * rng.rand()))
    nhi = (max_out - 1) - nlo
    if logp_many is None:
        while nlo > 0 and logp(theta + lo * direction) > level:
            lo -= 1.0
            nlo -= 1
        while nhi > 0 and logp(theta + hi * direction) > level:
            hi += 1.0
            nhi -= 1
    else:
        grow_lo, grow_hi = nlo > 0, nhi > 0
        per_side = max(1, spec // 2)
        while grow_lo or grow_hi:
            klo = min(per_side, nlo) if grow_lo else 0
            khi = min(per_side, nhi) if grow_hi else 0
            pts = [theta + (lo - i) * direction for i in range(klo)] + [theta + (hi + i) * direction for i in range(khi)]
            vals = np.asarray(logp_many(pts))
            for v in vals[:klo]:                 # the ends lo, lo-1, ...: expand while the density is above the level
                if v > level:
                    lo -= 1.0
                    nlo -= 1
                else:
                    grow_lo = False
                    break
            grow_lo = grow_lo and nlo > 0
            for v in vals[klo:]:
                if v > level:
                    hi += 1.0
                    nhi -= 1
                else:
                    grow_hi = False
                    break
            grow_hi = grow_hi and nhi > 0
    while True:
        if logp_many is None:
            t = lo + (hi - lo) * rng.rand()
            cand = theta + t * direction
            lpc = logp(cand)
            if lpc > level:
                return cand, lpc
            if t < 0:
                lo = t
            else:
                hi = t
            if hi - lo < 1e-12:                  # numerically collapsed bracket: stay put
                return theta, lp
            continue
        # speculate the next `spec` shrinkage candidates (each follows from the sign of the one before)
        state = rng.get_state()
        l2, h2, ts, dead = lo, hi, [], -1
        for i in range(spec):
            t = l2 + (h2 - l2) * rng.rand()
            ts.append(t)
            if t < 0:
                l2 = t
            else:
                h2 = t
            if h2 - l2 < 1e-12:
                dead = i                         # the sequential procedure would give up after rejecting this one
                break
        vals = np.asarray(logp_many([theta + t * direction for t in ts]))
        hit = next((i for i, v in enumerate(vals) if v > level), -1)
        used = (hit + 1) if hit >= 0 else len(ts)
        rng.set_state(state)
        rng.rand(used)                           # consume exactly the draws the sequential procedure would have
        if hit >= 0:
            return theta + ts[hit] * direction, float(vals[hit])
        if dead >= 0:
            return theta, lp
        lo, hi = l2, h2


class MCMC(object):
    def __init__(self, model, n=10, burn=100, rng=None):
        self._proto = model.copy()               # work-horse whose hyper-parameters the chain moves
        self._n = int(n)
        self._rng = rstate(rng)
        self._theta = np.array(self._proto.hyper_vector(), dtype=float)
        self._scale = np.ones(len(self._theta))
        self._scale[-1] = np.sqrt(np.exp(self._theta[1]))   # bias moves on the INITIAL signal std (frozen)
        self._lp = None
        self._members = []
        if self._proto.ndata > 0:
            self._advance(int(burn), keep=False)
            self._advance(self._n, keep=True)

    # The whole-grid top-k hooks only exist for device-backed members (policies probe them with getattr).  They are
    # properties, not attributes bound at construction: an instance that stores its own bound method is a reference
    # cycle, and every BO step makes several ensemble copies -- their 10 member handles each then wait for the cyclic
    # collector instead of going back to the handle pool at once.
    @property
    def acq_topk(self):
        if not hasattr(self._proto, '_engine'):      # single-handle device members only (gpx_ensemble_sweep)
            raise AttributeError('acq_topk')
        return self._acq_topk

    @property
    def topk_engine(self):
        if not hasattr(self._proto, '_engine'):
            raise AttributeError('topk_engine')
        return self._lead_engine

    def _lead_engine(self):
        return self._engines()[0]                # the ensemble's lead handle ranks the average

    @property
    def mean_topk(self):
        if not hasattr(self._proto, '_engine'):
            raise AttributeError('mean_topk')
        return self._mean_topk

    def predict_mean(self, X, grad=False):
        """Mixture mean only (= predict(X)[0]): the members' closed forms at the data where they have one
        (EI / PI targets and the recommenders ask for the mean at the observed points, pybo/policies/simple.py:21,35,
        pybo/recommenders.py:22-34 -- as a sweep that is n members x an N x N x N product)."""
        if grad:        # all members' moments and gradients come from ONE device call (gpx_ensemble_predict)
            post = self.predict(X, True)
            return post[0], post[2]
        members = self._need()
        if all(hasattr(m, 'predict_mean') for m in members):
            return np.mean([m.predict_mean(X) for m in members], axis=0)
        return self.predict(X)[0]

    def _mean_topk(self, xgrid, k):
        from .._lib import DeviceGrid
        if not isinstance(xgrid, DeviceGrid):
            xgrid = np.array(xgrid, ndmin=2, dtype=float)
            if self._members and self._members[0]._data_rows(xgrid) is not None:
                mu = self.predict_mean(xgrid)
                v = np.where(np.isnan(mu), -np.inf, mu)
                order = np.lexsort((np.arange(len(v)), -v))[:int(k)]
                return mu[order], order
        return self._acq_topk('mean', None, xgrid, k)

    # -- sampling ----------------------------------------------------------------------------------
    def _advance(self, nsteps, keep):
        if self._lp is None:
            self._lp = _log_target(self._proto, self._theta)
            if not np.isfinite(self._lp):
                raise ValueError('MCMC: the initial hyper-parameters have zero posterior density')
        kept = []
        logp = lambda th: _log_target(self._proto, th)      # noqa: E731
        many = (lambda ths: _log_targets(self._proto, ths)) if hasattr(self._proto, 'loglik_at') else None
        for _ in range(nsteps):
            self._theta, self._lp = _slice_update(logp, self._theta, self._lp, self._rng, self._scale,
                                                  logp_many=many)
            kept.append(self._theta.copy())
        if keep:
            members = []
            for th in kept:
                m = self._proto.copy()
                m.set_hyper_vector(th)
                m.loglikelihood()                # forces the (re)fit of this member
                members.append(m)
            self._members = members
        self._proto.set_hyper_vector(self._theta)

    @property
    def samples(self):
        """(n, 3 + d) array of the kept hyper-parameter states [log sn2, log rho, log ell.., bias]."""
        return np.array([m.hyper_vector() for m in self._members])

    # -- model protocol ----------------------------------------------------------------------------
    @property
    def ndata(self):
        return self._proto.ndata

    @property
    def data(self):
        return self._proto.data

    @property
    def params(self):
        return self._proto.params

    def copy(self):
        new = MCMC.__new__(MCMC)
        new._proto = self._proto.copy()
        new._n = self._n
        new._rng = self._rng                     # shared stream, as a chain continued from a copy would
        new._theta = self._theta.copy()
        new._scale = self._scale.copy()
        new._lp = self._lp
        new._members = [m.copy() for m in self._members]
        return new

    def add_data(self, X, Y):
        self._proto.add_data(X, Y)
        self._lp = None                          # the target changed with the data
        self._advance(self._n, keep=True)

    def _need(self):
        if not self._members:
            raise RuntimeError('the model has no data yet')
        return self._members

    def _member_grads(self, X):
        """(mu, s2, dmu, ds2) of every member, member-major: one device call for device members
        (gpx_ensemble_predict: their kernels overlap on the members' streams), a loop otherwise."""
        X = np.array(X, ndmin=2, dtype=float)
        engines = self._engines() if len(X) else None
        if engines is not None:
            from .._lib import Engine
            return Engine.ensemble_predict(engines, X)
        posts = [m.predict(X, True) for m in self._need()]
        return tuple(np.array([p[i] for p in posts]) for i in range(4))

    def predict(self, X, grad=False):
        engines = None if grad else self._engines()
        if engines is not None:                      # mixture moments formed on the device
            from .._lib import Engine
            out = Engine.ensemble_sweep(engines, 'mean', None, np.array(X, ndmin=2, dtype=float), k=0,
                                        want_all=False, want_moments=True)
            return out['mu'], out['s2']
        if grad:
            mus, s2s, dmus, ds2s = self._member_grads(X)
        else:
            posts = [m.predict(X, False) for m in self._need()]
            mus = np.array([p[0] for p in posts])
            s2s = np.array([p[1] for p in posts])
        mu = mus.mean(axis=0)
        s2 = np.maximum((s2s + mus ** 2).mean(axis=0) - mu ** 2, 0.0)
        if not grad:
            return mu, s2
        dmu = dmus.mean(axis=0)
        ds2 = (ds2s + 2.0 * mus[:, :, None] * dmus).mean(axis=0) - 2.0 * mu[:, None] * dmu
        return mu, s2, dmu, ds2

    def _mean_of(self, name, target, X, grad):
        engines = None if grad else self._engines()
        if engines is not None:
            from .._lib import Engine
            kind = {'get_improvement': 'ei', 'get_tail': 'pi'}[name]
            return Engine.ensemble_sweep(engines, kind, target, np.array(X, ndmin=2, dtype=float), k=0)['acq']
        if grad and self._engines() is not None and len(np.atleast_2d(X)):
            # device members: all members' moments and gradients in one call, the closed forms vectorised over them
            mu, s2, dmu, ds2 = self._member_grads(X)
            s = np.sqrt(s2)
            z = (mu - target) / s
            cdf = 0.5 * erfc(-z * 0.70710678118654752440)
            pdf = 0.39894228040143267794 * np.exp(-0.5 * z * z)
            if name == 'get_improvement':
                val = (mu - target) * cdf + s * pdf
                g = cdf[:, :, None] * dmu + (0.5 * pdf / s)[:, :, None] * ds2
            else:
                val = cdf
                g = pdf[:, :, None] * (dmu / s[:, :, None] - (0.5 * z / s2)[:, :, None] * ds2)
            return val.mean(axis=0), g.mean(axis=0)
        outs = [getattr(m, name)(target, X, grad) for m in self._need()]
        if not grad:
            return np.mean(outs, axis=0)
        return np.mean([o[0] for o in outs], axis=0), np.mean([o[1] for o in outs], axis=0)

    def get_improvement(self, target, X, grad=False):
        return self._mean_of('get_improvement', target, X, grad)

    def get_tail(self, target, X, grad=False):
        return self._mean_of('get_tail', target, X, grad)

    def sample_f(self, n, rng=None):
        rng = rstate(rng)
        members = self._need()
        return members[rng.randint(len(members))].sample_f(n, rng)

    def _engines(self):
        """The members' device engines when every member is a device GP (pybo_amd.models.GP), else None."""
        members = self._need()
        if not all(hasattr(m, '_engine') for m in members):
            return None
        return [m._engine() for m in members]

    def _acq_topk(self, kind, param, xgrid, k):
        """Ensemble acquisition over a grid + top-k in one device call (gpx_ensemble_sweep): every member
        sweeps the grid, the n value vectors are averaged in HBM (mean for EI/PI/mean, mixture moments for
        UCB) and only the k winners come back.  `xgrid`: host array or `DeviceGrid`."""
        from .._lib import Engine, DeviceGrid
        if not isinstance(xgrid, DeviceGrid):
            xgrid = np.array(xgrid, ndmin=2, dtype=float)
        out = Engine.ensemble_sweep(self._engines(), kind, param, xgrid, k=int(k), want_all=False)
        return out['top_val'], out['top_idx']

    # -- pickling: hyper-parameter states + data, members are rebuilt on load -------------------------
    def __getstate__(self):
        return dict(proto=self._proto, n=self._n, rng=self._rng, theta=self._theta, scale=self._scale,
                    samples=[np.array(m.hyper_vector()) for m in self._members])

    def __setstate__(self, st):
        self._proto, self._n, self._rng, self._theta = st['proto'], st['n'], st['rng'], st['theta']
        self._scale = st['scale']
        self._lp = None
        self._members = []
        for th in st['samples']:
            m = self._proto.copy()
            m.set_hyper_vector(th)
            self._members.append(m)
