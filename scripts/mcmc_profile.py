"""cProfile of MCMC.add_data with device members (where do the milliseconds go)."""
import os, sys, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from helpers import synth_problem
from pybo_amd import models
N, d = 100, 3
X, y, ell = synth_problem(N + 8, d, seed=2)
m = models.make_gp(1e-3, 1.2, ell, 0.1)
m.params['like.sn2'].set_prior('horseshoe', 0.1)
m.params['kern.rho'].set_prior('lognormal', 0.0, 1.0)
m.params['kern.ell'].set_prior('uniform', [0.02] * d, [3.0] * d)
m.params['mean.bias'].set_prior('normal', 0.0, 4.0)
m.add_data(X[:N], y[:N])
mc = models.MCMC(m, n=10, burn=100, rng=0)
mc.add_data(X[N], y[N]); mc.add_data(X[N + 1], y[N + 1])
pr = cProfile.Profile()
pr.enable()
for i in range(2, 7):
    mc.add_data(X[N + i], y[N + i])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
