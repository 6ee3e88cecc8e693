"""A/B of factorisation options on one box: python scripts/chol_ab.py N name=v[,name=v] [name=v ...]; prints the
HIP-event time of the Cholesky (median / min of 7) per option set and max |L - L_default| (bit-identity check)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine

N = int(sys.argv[1])
sets = [''] + sys.argv[2:]
rng = np.random.RandomState(1)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
L0 = None
for rep in range(2):
    for opts in sets:
        e = Engine(0)
        for kv in filter(None, opts.split(',')):
            k, v = kv.split('=')
            e.set_option(k, int(v))
        ts = []
        for r in range(8):
            e.timers(reset=True)
            e.fit(X, y, 'se', ell, rho, sn2, bias)
            e.sync()
            ts.append(e.timers(reset=True)['cholesky'])
        L = e.get_matrix('L') if (rep == 0 and N <= 8192) else None
        if L0 is None and L is not None:
            L0 = L
        diff = '' if L is None else '   max|L-L_default| %.2e' % np.abs(L - L0).max()
        print('N=%d  %-28s median %.3f ms  min %.3f%s' % (N, opts or 'default', np.median(ts[1:]), min(ts[1:]), diff), flush=True)
        e.close()
