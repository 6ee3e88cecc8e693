"""Host wall clock of the fit up to the factor (gram + Cholesky), no timer calls in between: python scripts/ab/chol_wall.py N [opt=v ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine
N = int(sys.argv[1])
rng = np.random.RandomState(N)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, int(v))
ts = []
for r in range(8):
    e.sync(); t0 = time.perf_counter()
    e.fit(X, y, 'se', ell, rho, sn2, bias, stage=2); e.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
tm = e.timers(reset=True)
print('N=%d %s  wall of fit(stage 2) median %.3f min %.3f ms; timers: cholesky %.3f per fit' % (N, ' '.join(sys.argv[2:]), np.median(ts[2:]), min(ts[2:]), tm['cholesky'] / 8), flush=True)
