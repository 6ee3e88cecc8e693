"""gpx_loglik_batch at BASELINE sizes: what the hyper-parameter sampler of the reference's default
model (MCMC(gp, n=10, burn=100), pybo/bayesopt.py:115) would pay per likelihood evaluation at N = 2048 and 8192 --
one vector per call, and ten per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from helpers import synth_problem
from pybo_amd import models

for N, d in ((2048, 2), (8192, 8)):
    X, y, ell = synth_problem(N, d, seed=1)
    gp = models.make_gp(1e-3, 1.2, ell, 0.1)
    gp.add_data(X, y)
    th0 = gp.hyper_vector()
    rng = np.random.RandomState(0)
    ths = th0 + 0.05 * rng.randn(20, len(th0))
    line = 'N %5d d %d:' % (N, d)
    for B in (1, 10):
        v = gp.loglik_at(ths[:B])                 # (untimed: allocates the batch buffers)
        t0 = time.perf_counter()
        reps = 0
        for i in range(0, len(ths) - B + 1, B):
            gp.loglik_at(ths[i:i + B]); reps += B
        dt = (time.perf_counter() - t0) / reps
        flop = N ** 3 / 3.0 + N * N * (3.0 * d + 30)
        line += '  B=%2d: %8.2f ms per vector (%.1f TFLOP/s of the N^3/3 + Gram)' % (B, dt * 1e3, flop / dt / 1e12)
    # the oracle's value for the first vector
    print(line, flush=True)
    from oracle import gp_ref
    sn2, rho, ellv, bias = gp.unpack_hyper_vector(ths[0]) if hasattr(gp, 'unpack_hyper_vector') else (None,) * 4
    if sn2 is not None:
        ref = gp_ref.make_gp(sn2, rho, ellv, bias)
        ref.add_data(X, y)
        print('   loglik of vector 0: device %.10g  oracle %.10g' % (v[0], ref.loglikelihood()))
