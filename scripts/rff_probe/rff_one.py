"""One shape of the Thompson sweep for counter runs: python scripts/rff_probe/rff_one.py d S n log2M x_rff"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pybo_amd._lib import Engine
d, S, n, lm, v = (int(a) for a in sys.argv[1:6])
M = 1 << lm
rng = np.random.RandomState(0)
e = Engine(0)
e.fit(rng.rand(256, d), rng.randn(256), 'se', 0.5 * np.ones(d), 1.0, 1e-2, 0.0, stage=2)
e.set_option('x_rff', v)
W = rng.randn(S, n, d) * 2.0; b = rng.rand(S, n) * 2 * np.pi; th = rng.randn(S, n) * 0.1
Xc = rng.rand(M, d)
for r in range(2):
    e.rff_sweep(W, b, th, 0.3, Xc, k=1, want_all=False)
print(e.timers()['rff_sweep'] / 2)
