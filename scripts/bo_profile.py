"""Where the time of a DEFAULT solve_bayesopt run goes (pybo's defaults: MCMC(gp, n=10, burn=100) model, EI,
lbfgs solver with 10000 uniform candidates, latent recommender).  cProfile, cumulative."""
import os, sys, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import pybo_amd
from helpers import branin
bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
f = lambda x: -branin(np.atleast_2d(x))[0] / 10.0
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.perf_counter()
pybo_amd.solve_bayesopt(f, bounds, niter=5, rng=0)      # warm-up (library load, first allocations)
print('warm-up run (5 iterations): %.2f s' % (time.perf_counter() - t0))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
xbest, model, info = pybo_amd.solve_bayesopt(f, bounds, niter=niter, rng=0)
pr.disable()
el = time.perf_counter() - t0
print('%d iterations: %.2f s = %.1f ms per iteration; best f = %.4f' % (niter, el, el / niter * 1e3, max(info.y)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
