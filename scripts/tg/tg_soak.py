"""Soak of the task-graph Cholesky: many factorisations per size, every factor compared bit for bit with the first, fallbacks
to the stream schedule counted (there must be none).  python scripts/tg/tg_soak.py [reps]"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for N in (1536, 2048, 3001, 4096, 8192, 12288):
    rng = np.random.RandomState(N)
    X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
    e = Engine(0)
    first, bad = None, 0
    n = reps if N <= 4096 else max(20, reps // 6)
    e.timers(reset=True)
    for r in range(n):
        e.fit(X, y, 'se', ell, rho, sn2, bias, stage=2)
        if r % 10 == 0 or r == n - 1:
            dg = hashlib.sha256(np.ascontiguousarray(e.get_matrix('L')).tobytes()).hexdigest()
            if first is None: first = dg
            bad += dg != first
    tm = e.timers(reset=True)
    print('N=%5d  %4d factorisations  mismatching factors %d  fallbacks %d  mean %.3f ms' % (N, n, bad, int(tm.get('chol_fallbacks', 0)), tm['cholesky'] / n), flush=True)
    e.close()
