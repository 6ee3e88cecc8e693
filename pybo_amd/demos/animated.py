"""
Headless version of the reference's animated 1-D demo (/root/reference/pybo/demos/animated.py:23-78):
Gramacy & Lee's test function on [0.5, 2.5], a GP with priors wrapped in an MCMC ensemble of 20
hyper-parameter samples, EI with xi = 0.1, the incumbent recommender and the grid + L-BFGS solver -- the same
loop body, minus the plotting, on the MI355X engine.

    python -m pybo_amd.demos.animated [niter]
"""
import sys

import numpy as np

from .. import inits, policies, recommenders, solvers
from ..models import MCMC, make_gp

XOPT = 0.54856343          # animated.py:36-37
BOUNDS = np.array([[0.5, 2.5]])


def gramacy_lee(x):
    """-sin(10 pi x)/(2x) - (x-1)^4, maximised (animated.py:23-29)."""
    x = float(np.ravel(x)[0])
    return -(np.sin(10 * np.pi * x) / (2 * x) + (x - 1) ** 4)


def run(niter=30, rng=0, verbose=True):
    rng = np.random.RandomState(rng)
    X = list(inits.init_latin(BOUNDS, 3, rng))                       # animated.py:42
    Y = [gramacy_lee(x) for x in X]
    gp = make_gp(0.01, 1.9, 0.1, 0.0)                                # animated.py:47
    gp.params['like.sn2'].set_prior('lognormal', -2, 1)              # animated.py:51-54
    gp.params['kern.rho'].set_prior('lognormal', 0, 1)
    gp.params['kern.ell'].set_prior('lognormal', -2, 1)
    gp.params['mean.bias'].set_prior('normal', 0, 20)
    gp.add_data(X, Y)
    model = MCMC(gp, n=20, rng=rng)                                  # animated.py:57
    xbest = None
    for i in range(niter):
        index = policies.EI(model, BOUNDS, X, xi=0.1)                # animated.py:64
        xbest = recommenders.best_incumbent(model, BOUNDS, X)        # animated.py:67
        xnext, _ = solvers.solve_lbfgs(index, BOUNDS, rng=rng)       # animated.py:68
        del index
        ynext = gramacy_lee(xnext)
        model.add_data(xnext, ynext)                                 # animated.py:78
        X.append(xnext)
        Y.append(ynext)
        if verbose:
            print('i=%02d  x=%.5f  y=% .5f  incumbent=%.5f' % (i, xnext[0], ynext, np.ravel(xbest)[0]))
    return np.array(X), np.array(Y), xbest


if __name__ == '__main__':
    Xs, Ys, xb = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
    print('best observed x = %.6f (optimum %.6f), y = %.6f' % (Xs[np.argmax(Ys)][0], XOPT, Ys.max()))
