"""
Recommenders (/root/reference/pybo/recommenders.py:14-35): where to point the user after each step.
`best_latent` maximises the posterior mean starting from the observed points (through solve_lbfgs with
`xgrid=X`), `best_incumbent` returns the observed point with the highest posterior mean.
"""
import numpy as np

from . import solvers

__all__ = ['best_latent', 'best_incumbent']


def best_latent(model, bounds, X):
    mean_only = getattr(model, 'predict_mean', None)     # device models: no variance, no pass over the factor

    def mean(X, grad=False):
        if mean_only is not None:
            return mean_only(X, grad) if grad else mean_only(X)
        if grad:
            post = model.predict(X, True)
            return post[0], post[2]
        return model.predict(X)[0]

    fast = getattr(model, 'mean_topk', None)        # device models: closed form at the data, else the device sweep
    if fast is not None:
        mean.topk = fast
    xbest, _ = solvers.solve_lbfgs(mean, bounds, xgrid=X)
    return xbest


def best_incumbent(model, _, X):
    mean_only = getattr(model, 'predict_mean', None)
    mu = mean_only(X) if mean_only is not None else model.predict(X)[0]
    return np.asarray(X)[int(np.argmax(mu))]
