"""
Acquisition policies EI / PI / UCB / Thompson with pybo's signatures
`policy(model, bounds, X, **kw[, rng]) -> index`, `index(X, grad=False)`
(/root/reference/pybo/policies/simple.py:16-74).  Behaviour kept on purpose:

  * EI / PI copy the model, then take `target = max posterior mean at the observed X + xi`
    (simple.py:20-21, 34-35) -- one `model.predict(X)` per policy call;
  * UCB's `d` is the NUMBER OF OBSERVATIONS len(X), not the input dimension (simple.py:58-60):
    beta = xi*2*log(pi^2/(3 delta)) + xi*(4+N)*log(N+1);
  * Thompson returns ONE posterior function sample's `.get` (simple.py:48); `n` is the number of random
    Fourier features.

New: when the model is the device-backed `pybo_amd.models.GP`, the returned index also carries
`index.topk(xgrid, k)` -- the whole-grid evaluation + top-k done on the GPU in one call -- which
`pybo_amd.solvers.solve_lbfgs` uses instead of `argsort(f(xgrid))` (pybo/solvers/lbfgs.py:50-51).
Any other model (e.g. a test stub) gets plain closures, exactly as in the reference.
"""
import numpy as np

__all__ = ['EI', 'PI', 'UCB', 'Thompson']


def _attach_topk(index, model, kind, param):
    fast = getattr(model, 'acq_topk', None)
    if fast is not None:
        index.topk = lambda xgrid, k: fast(kind, param, xgrid, k)
        owner = getattr(model, 'topk_engine', None)
        if owner is not None:              # which device handle holds the top-k pairs (pybo_amd.dist, comm=)
            index.topk_engine = owner
    return index


def _incumbent_target(model, X, xi):
    """Copy the model (the caller's must stay untouched) and compute best-posterior-mean-at-the-data + xi."""
    snapshot = model.copy()
    mean_only = getattr(snapshot, 'predict_mean', None)      # device models: no variance sweep nobody reads
    mu = mean_only(X) if mean_only is not None else snapshot.predict(X)[0]
    return snapshot, mu.max() + xi


def EI(model, _, X, xi=0.0):
    """Expected improvement over (best posterior mean at the data) + xi."""
    model, target = _incumbent_target(model, X, xi)

    def index(X, grad=False):
        return model.get_improvement(target, X, grad)

    return _attach_topk(index, model, 'ei', target)


def PI(model, _, X, xi=0.05):
    """Probability of improvement over (best posterior mean at the data) + xi."""
    model, target = _incumbent_target(model, X, xi)

    def index(X, grad=False):
        return model.get_tail(target, X, grad)

    return _attach_topk(index, model, 'pi', target)


def Thompson(model, _, __, n=100, rng=None):
    """Thompson sampling: the index is one posterior function sample built from n random features."""
    sample = model.sample_f(n, rng)
    fast = getattr(sample, 'topk', None)
    if fast is None:
        return sample.get

    def index(X, grad=False):
        return sample.get(X, grad)

    index.topk = fast                     # device sample: grid evaluation + top-k stay on the GPU
    return index


def UCB(model, _, X, delta=0.1, xi=0.2):
    """GP-UCB.  `delta`: failure probability of the bound; `xi`: scale of the exploration term.
    NB the reference's `d` is len(X), the number of observations (simple.py:58)."""
    model = model.copy()
    nobs = len(X)
    beta = xi * 2 * np.log(np.pi ** 2 / 3 / delta) + xi * (4 + nobs) * np.log(nobs + 1)

    def index(X, grad=False):
        if not grad:
            mu, s2 = model.predict(X)
            return mu + np.sqrt(beta * s2)
        mu, s2, dmu, ds2 = model.predict(X, grad=True)
        width = np.sqrt(beta * s2)
        return mu + width, dmu + 0.5 * np.sqrt(beta / s2[:, None]) * ds2

    return _attach_topk(index, model, 'ucb', beta)
