"""
Recommenders (/root/reference/pybo/recommenders.py:14-35): where to point the user after each step.
`best_latent` maximises the posterior mean starting from the observed points (through solve_lbfgs with
`xgrid=X`), `best_incumbent` returns the observed point with the highest posterior mean.
"""
import numpy as np

from . import solvers

__all__ = ['best_latent', 'best_incumbent']


def best_latent(model, bounds, X):
    def mean(X, grad=False):
        if grad:
            post = model.predict(X, True)
            return post[0], post[2]
        return model.predict(X)[0]

    fast = getattr(model, 'acq_topk', None)
    if fast is not None:
        mean.topk = lambda xgrid, k: fast('mean', None, xgrid, k)
    xbest, _ = solvers.solve_lbfgs(mean, bounds, xgrid=X)
    return xbest


def best_incumbent(model, _, X):
    mu, _ = model.predict(X)
    return np.asarray(X)[int(np.argmax(mu))]
