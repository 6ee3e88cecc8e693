"""What the factorisation's parts cost ALONE (diagnostic option x_skip: 1 = no far updates, 2 = no chain kernels,
4 = no near updates).  Near-diagonal data (tiny length scale) so that a skipped update leaves the matrix
positive definite and the chain's timing valid."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine

for N in [int(a) for a in sys.argv[1:]] or [8192, 16384]:
    rng = np.random.RandomState(0)
    X = rng.rand(N, 8); y = rng.randn(N); ell = np.full(8, 1e-3)
    e = Engine(0)
    for name, skip in (('all', 0), ('chain + near (no far)', 1), ('chain only', 5), ('far + near only', 2), ('far only', 6)):
        e.set_option('x_skip', skip)
        ts = []
        for r in range(5):
            e.timers(reset=True)
            try:
                e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
            except Exception as ex:
                pass
            e.sync()
            ts.append(e.timers(reset=True)['cholesky'])
        print('N=%d  %-24s median %.3f ms  min %.3f' % (N, name, np.median(ts[1:]), min(ts[1:])), flush=True)
    e.close()
