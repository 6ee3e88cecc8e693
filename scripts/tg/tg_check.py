"""First-light check of the task-graph Cholesky on the GPU: bit-identity against the stream schedule and timings.
python scripts/tg/tg_check.py [sizes...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [129, 256, 300, 640, 1024, 2048, 4096, 8192]
opts = [a for a in sys.argv[1:] if '=' in a]
for N in sizes:
    rng = np.random.RandomState(N)
    X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
    res = {}
    for tg in (0, 1):
        e = Engine(0)
        e.set_option('chol_tg', tg)
        if tg:
            e.set_option('chol_tg_tmo_ms', 300)
            for kv in opts:
                k, v = kv.split('=')
                e.set_option(k, int(v))
        ts = []
        Ls = []
        for r in range(6):
            e.timers(reset=True)
            t0 = time.time()
            e.fit(X, y, 'se', ell, rho, sn2, bias, stage=2)
            e.sync()
            wall = (time.time() - t0) * 1e3
            ts.append(e.timers(reset=True)['cholesky'])
            if N <= 8192 and r in (0, 5):
                Ls.append(e.get_matrix('L'))
        res[tg] = (ts, Ls)
        e.close()
    ts0, L0 = res[0]; ts1, L1 = res[1]
    same = all(np.array_equal(L, L0[0]) for L in L1) if L0 else None
    nbad = int((L1[0] != L0[0]).sum()) if L0 and not same else 0
    print('N=%6d  streams: median %.3f min %.3f ms   task-graph: median %.3f min %.3f ms (first %.3f)   bit-identical: %s %s'
          % (N, np.median(ts0[1:]), min(ts0[1:]), np.median(ts1[1:]), min(ts1[1:]), ts1[0], same, nbad or ''), flush=True)
