"""How much do co-resident fp64-MFMA waves slow the factorisation's CHAIN down when it never has to wait for a CU slot?
(diagnostic options x_bg / x_bg_lds / x_bg_iters: a synthetic register-only MFMA kernel of G workgroups on its own stream
next to the chain + near kernels, far updates left out: x_skip = 1.)  G = 256 with 100 KB of LDS: one background
workgroup per CU, half of every CU free; G = 384 / 72 KB: half of the CUs carry two.  The factorisation time reported
is the chain's (HIP events on the main stream); the background kernel runs longer than it.
Run with GPU_MAX_HW_QUEUES=8: with the default of 4 hardware queues the fifth stream shares a queue with one of the
factorisation's and its kernels wait behind the background kernel (12 ms instead of 5)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.RandomState(0)
X = rng.rand(N, 8); y = rng.randn(N); ell = np.full(8, 1e-3)
e = Engine(0)
e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
for skip, sname in ((1, 'chain + near'), (5, 'chain only')):
    for G, lds in ((0, 72), (64, 100), (128, 100), (256, 100), (256, 72), (384, 72), (448, 72)):
        e.set_option('x_skip', skip)
        e.set_option('x_bg', G)
        e.set_option('x_bg_lds', lds)
        e.set_option('x_bg_iters', 16000 if lds >= 100 or G <= 256 else 9000)     # ~7 ms of background
        ts = []
        for r in range(4):
            e.timers(reset=True)
            try:
                e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
            except Exception:
                pass
            e.sync()
            ts.append(e.timers(reset=True)['cholesky'])
        print('N=%d  %-13s background G=%3d x %3d KB LDS: median %.3f ms  min %.3f' % (N, sname, G, lds, np.median(ts[1:]), min(ts[1:])), flush=True)
e.close()
