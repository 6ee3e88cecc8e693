import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
from pybo_amd._lib import Engine
rng = np.random.RandomState(0)
for N in (8192, 4096):
    X = rng.rand(N, 8); y = np.sin(X.sum(1))
    e = Engine(0)
    e.fit(X, y, 'se', np.full(8, 0.5), 1.0, 1e-3, 0.0)
    Z = rng.rand(1, 8)
    e.predict(Z, grad=True)
    for mb in (1, 10):
        Z = rng.rand(mb, 8)
        e.sync(); t0 = time.perf_counter()
        for _ in range(50): r = e.predict(Z, grad=True)
        dt = (time.perf_counter() - t0) / 50
        print('N=%d  predict(grad) of %2d points: %.1f us per call  (checksum %.17g)' % (N, mb, dt * 1e6, float(np.sum(r[3]))))
    e.close()
