import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = 8
rng = np.random.RandomState(1)
X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
for it in range(3):
    t0 = time.time(); e.fit(X, y, 'se', ell, rho, sn2, bias); e.sync(); t1 = time.time()
    tm = e.timers(reset=True)
    print(f"N={N} fit wall {t1-t0:.4f}s  " + " ".join(f"{k}={v:.2f}" for k, v in tm.items() if v))
