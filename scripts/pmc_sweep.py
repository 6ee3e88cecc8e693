import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
order = int(sys.argv[1]); N = 8192; d = 8; M = 1 << 16
rng = np.random.RandomState(1)
X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0); e.set_option('tile_order', order)
e.fit(X, y, 'se', ell, rho, sn2, bias)
for _ in range(2):
    e.sweep('ei', 0.0, rng.rand(M, d), k=4, want_all=False)
print(e.timers())
