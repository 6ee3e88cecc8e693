"""Median time of the triangular inverse (lazy: forced by eager_inverse) at size N."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine
N = int(sys.argv[1])
rng = np.random.RandomState(N)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
e.set_option('eager_inverse', 1)
ts = []
for r in range(6):
    e.timers(reset=True)
    e.fit(X, y, 'se', ell, rho, sn2, bias); e.sync()
    ts.append(e.timers(reset=True)['trtri'])
print('N=%d trtri median %.3f min %.3f ms' % (N, np.median(ts[1:]), min(ts[1:])), flush=True)
