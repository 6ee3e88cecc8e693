"""Log-likelihood evaluations per second as the hyper-parameter sampler issues them: the round-1 path (set the
hyper-parameters, refit with gpx_fit, gpx_loglik) against gpx_loglik_batch with batch 1 (one proposal), 2 (the two
ends of a stepping-out round) and 10 (an ensemble's worth), at the sizes pybo's default model works at."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from helpers import synth_problem
from pybo_amd import models

for N in (30, 100, 300, 500, 1000):
    d = 3
    X, y, ell = synth_problem(N, d, seed=1)
    gp = models.make_gp(1e-3, 1.2, ell, 0.1)
    gp.add_data(X, y)
    th0 = gp.hyper_vector()
    rng = np.random.RandomState(0)
    ths = th0 + 0.1 * rng.randn(40, len(th0))
    g2 = gp.copy()
    g2.set_hyper_vector(ths[0]); g2.loglikelihood()
    t0 = time.perf_counter()
    for th in ths:
        g2.set_hyper_vector(th); g2.loglikelihood()
    t_old = (time.perf_counter() - t0) / len(ths)
    res = []
    for B in (1, 2, 10):
        gp.loglik_at(ths[:B])
        t0 = time.perf_counter()
        n = 0
        for i in range(0, len(ths) - B + 1, B):
            gp.loglik_at(ths[i:i + B]); n += B
        res.append((time.perf_counter() - t0) / n)
    print('N %5d: refit+loglik %7.1f us/eval | loglik_batch B=1 %7.1f  B=2 %7.1f  B=10 %7.1f us/eval | speed-up %4.1fx %4.1fx %4.1fx'
          % (N, t_old * 1e6, res[0] * 1e6, res[1] * 1e6, res[2] * 1e6, t_old / res[0], t_old / res[1], t_old / res[2]))

# the whole default-model refresh of pybo (MCMC(gp, n=10, burn=100) at construction, n=10 updates per add_data)
from pybo_amd.models import gp as gpmod
for N in (30, 100, 300):
    d = 3
    X, y, ell = synth_problem(N + 5, d, seed=2)
    times = {}
    for label in ('batched', 'sequential'):
        saved = None
        if label == 'sequential':
            saved = gpmod.GP.loglik_at
            del gpmod.GP.loglik_at
        try:
            m = models.make_gp(1e-3, 1.2, ell, 0.1)
            m.params['like.sn2'].set_prior('horseshoe', 0.1)
            m.params['kern.rho'].set_prior('lognormal', 0.0, 1.0)
            m.params['kern.ell'].set_prior('uniform', [0.02] * d, [3.0] * d)
            m.params['mean.bias'].set_prior('normal', 0.0, 4.0)
            m.add_data(X[:N], y[:N])
            t0 = time.perf_counter()
            mc = models.MCMC(m, n=10, burn=100, rng=0)
            t1 = time.perf_counter()
            for i in range(5):
                mc.add_data(X[N + i], y[N + i])
            t2 = time.perf_counter()
            times[label] = (t1 - t0, (t2 - t1) / 5, mc.samples.copy())
        finally:
            if saved is not None:
                gpmod.GP.loglik_at = saved
    same = np.allclose(times['batched'][2], times['sequential'][2], rtol=1e-6, atol=1e-8)
    print('N %4d: MCMC(n=10, burn=100) %6.1f ms batched vs %6.1f ms sequential; add_data %5.1f vs %5.1f ms; same chain: %s'
          % (N, times['batched'][0] * 1e3, times['sequential'][0] * 1e3, times['batched'][1] * 1e3,
             times['sequential'][1] * 1e3, same))
