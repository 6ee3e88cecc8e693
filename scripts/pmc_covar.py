"""One cold sweep + one warm step at N = 8192, d = 8, SE-ARD over 2^18 candidates, for counter passes over the two
covariance-evaluation kernels (k_cross_gram, k_sweep_rankq<1>):
    scripts/pmc_cmd.sh cov1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" scripts/pmc_covar.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
N, d, M = 8192, 8, 1 << 18
rng = np.random.RandomState(3)
e = Engine(0)
e.set_option('sweep_cache', 1)
X = rng.rand(N, d)
e.fit(X, np.sin(X.sum(axis=1)), 'se', np.full(d, 1.0), 1.0, 1e-3, 0.0)
Z = rng.rand(M, d)
e.sweep('ei', 0.5, Z, k=8, want_all=False)
e.timers(reset=True)
e.append(rng.rand(d), 0.1)
e.sweep_update('ei', 0.5, k=8, want_all=False)
print('rank-1 pass ms: %.3f' % e.timers()['rank1'])
