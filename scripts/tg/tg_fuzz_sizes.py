"""Random sizes through the task-graph Cholesky: every factor compared bit for bit with the stream schedule's, fallbacks counted.
python scripts/tg/tg_fuzz_sizes.py [sizes [seed]]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine
nsz = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
sizes = sorted(set([129, 255, 256, 257, 384, 385, 511, 513, 640, 1023, 1025] + list(rng.randint(130, 5200, size=nsz))))
a, b = Engine(0), Engine(0)
b.set_option('chol_tg', 0)
bad = 0
for N in sizes:
    d = int(rng.randint(1, 9))
    X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
    for rep in range(3):
        a.fit(X, y, 'se', ell, rho, sn2, bias, stage=2)
    b.fit(X, y, 'se', ell, rho, sn2, bias, stage=2)
    ok = np.array_equal(a.get_matrix('L'), b.get_matrix('L'))
    bad += not ok
    if not ok: print('MISMATCH at N = %d' % N, flush=True)
tm = a.timers(reset=True)
print('%d sizes from %d to %d, 3 factorisations each: mismatching factors %d, fallbacks %d' % (len(sizes), sizes[0], sizes[-1], bad, int(tm.get('chol_fallbacks', 0))), flush=True)
