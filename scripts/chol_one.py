"""One configuration of the factorisation, 3 fits (for rocprofv3 --kernel-trace).  usage: chol_one.py N name=value ..."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine
N = int(sys.argv[1])
rng = np.random.RandomState(0)
X = rng.rand(N, 8); y = rng.randn(N); ell = np.full(8, 1e-3)
e = Engine(0)
for a in sys.argv[2:]:
    k, v = a.split('=')
    e.set_option(k, int(v))
for r in range(3):
    e.timers(reset=True)
    try:
        e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
    except Exception as ex:
        print('fit:', ex)
    e.sync()
    print('cholesky %.3f ms' % e.timers(reset=True)['cholesky'])
e.close()
