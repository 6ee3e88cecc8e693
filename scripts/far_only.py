"""Far trailing updates of the factorisation ALONE (x_skip = 6) under option sets -- does a larger K per far tile (two-panel
accumulation, wider panels) make the far part faster?  (No: 3.49 vs 3.63 ms at N = 8192 with chol_merge; the wider-panel
rows of the table move work into the near updates, which this mode skips.)  python scripts/far_only.py"""
import sys, os
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from pybo_amd._lib import Engine
for N in (8192, 16384):
    rng = np.random.RandomState(0)
    X = rng.rand(N, 8); y = rng.randn(N); ell = np.full(8, 1e-3)
    e = Engine(0)
    for opts in ({}, {'chol_merge': 1}, {'chol_merge': 1, 'chol_w': 8}, {'chol_w': 8}):
        e.set_option('chol_merge', 0); e.set_option('chol_w', 0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_option('x_skip', 6)
        ts = []
        for r in range(5):
            e.timers(reset=True)
            try:
                e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
            except Exception:
                pass
            e.sync()
            ts.append(e.timers(reset=True)['cholesky'])
        print('N=%d far only %-34s median %.3f ms' % (N, opts, np.median(ts[1:])), flush=True)
    e.close()
