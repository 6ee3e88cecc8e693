"""Beyond BASELINE's sizes: N = 32768 (4 factor buffers of 8.6 GB, 1.07e9 elements each -- 64-bit indexing everywhere) and
N = 24576, fit + posterior at 96 points against the oracle on the box's host, and the stage timers.
    python scripts/big_n_check.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gp_ref                      # noqa: E402  (a checking script, not the product)
from pybo_amd._lib import Engine               # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [24576, 32768]:
    d = 8
    rng = np.random.RandomState(N)
    X = rng.rand(N, d)
    y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell, rho, bias = 0.25 * np.ones(d), float(np.var(y)), float(np.mean(y))
    sn2 = 1e-4 * rho
    Z = rng.rand(96, d)
    e = Engine(0)
    t0 = time.time()
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    r = e.sweep('ei', float(y.max()), Z, k=5, want_moments=True)
    e.sync()
    tm = e.timers(reset=True)
    print('N=%d device: fit + 96-point sweep %.2f s wall; cholesky %.1f ms (%.1f TFLOP/s), inverse %.1f ms'
          % (N, time.time() - t0, tm['cholesky'], N ** 3 / 3.0 / tm['cholesky'] / 1e9, tm['trtri']), flush=True)
    t0 = time.time()
    ref = gp_ref.make_gp(sn2, rho, ell, bias)
    ref.add_data(X, y)
    mr, sr = ref.predict(Z)
    print('N=%d oracle on the host: %.1f s' % (N, time.time() - t0), flush=True)
    em = np.max(np.abs(r['mu'] - mr) / (1e-6 * np.abs(mr) + 1e-9 * np.sqrt(rho)))
    es = np.max(np.abs(r['s2'] - sr) / (1e-6 * sr + 1e-10 * rho))
    print('N=%d max error in units of the stated tolerance: mu %.2e, s2 %.2e -> %s'
          % (N, em, es, 'OK' if em <= 1 and es <= 1 else 'FAIL'), flush=True)
    e.close()
