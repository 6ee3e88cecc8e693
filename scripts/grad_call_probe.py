"""Wall time of one predict-with-gradients call (gpx_predict) against the number of rows in the call."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = sys.argv[3:]
rng = np.random.RandomState(N)
X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
for kv in opts:
    k, v = kv.split('='); e.set_option(k, int(v))
e.fit(X, y, 'se', ell, rho, sn2, bias)
for M in (1, 2, 4, 9, 16):
    Z = rng.rand(M, d)
    e.predict(Z, grad=True)
    ts = []
    for r in range(60):
        t0 = time.perf_counter(); e.predict(Z, grad=True); ts.append(time.perf_counter() - t0)
    print('N=%d d=%d %s rows=%2d: median %.1f us  min %.1f us' % (N, d, ' '.join(opts), M, 1e6 * np.median(ts), 1e6 * min(ts)), flush=True)
Z = rng.rand(1, d)
ts = []
for r in range(60):
    t0 = time.perf_counter(); e.predict_mean(Z, grad=True); ts.append(time.perf_counter() - t0)
print('predict_mean rows=1: median %.1f us' % (1e6 * np.median(ts)))
