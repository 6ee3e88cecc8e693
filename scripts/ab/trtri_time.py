"""Median time of the triangular inverse at size N: python scripts/ab/trtri_time.py N [opt=v ...]   (trtri_ahead=0: the inverse alone,
not the part of it left after a factorisation it rode behind)."""
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
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, int(v))
ts, tc = [], []
for r in range(6):
    e.timers(reset=True)
    e.fit(X, y, 'se', ell, rho, sn2, bias); e.sync()
    tm = e.timers(reset=True)
    ts.append(tm['trtri']); tc.append(tm['cholesky'])
print('N=%d %s trtri median %.3f min %.3f ms; cholesky median %.3f; sum %.3f' % (N, ' '.join(sys.argv[2:]), np.median(ts[1:]), min(ts[1:]), np.median(tc[1:]), np.median(ts[1:]) + np.median(tc[1:])), flush=True)
