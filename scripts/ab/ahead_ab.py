"""The inversion's leading part riding behind the factorisation (option trtri_ahead) against the serial order:
Cholesky + triangular inverse (+ alpha) per fit, HIP-event spans and the host's wall clock, and the results bit for bit.
python scripts/ab/ahead_ab.py N [reps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine
N = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.RandomState(N)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
Xq = rng.rand(256, 8)
out = {}
for ahead in (0, 1, 0, 1):
    e = Engine(0)
    e.set_option('trtri_ahead', ahead); e.set_option('eager_inverse', 1)
    ch, tr, wall = [], [], []
    for r in range(reps):
        e.timers(reset=True)
        e.sync(); t0 = time.perf_counter()
        e.fit(X, y, 'se', ell, rho, sn2, bias); e.sync()
        wall.append((time.perf_counter() - t0) * 1e3)
        tm = e.timers(reset=True)
        ch.append(tm['cholesky']); tr.append(tm['trtri'])
    mu, s2 = e.predict(Xq)
    key = (mu.tobytes(), s2.tobytes())
    out.setdefault('ref', key)
    print('N=%d trtri_ahead=%d  cholesky %.3f  trtri %.3f  sum %.3f ms (medians; min sum %.3f)  host wall of the fit %.3f   posterior %s' % (
        N, ahead, np.median(ch[1:]), np.median(tr[1:]), np.median(np.array(ch[1:]) + np.array(tr[1:])),
        (np.array(ch[1:]) + np.array(tr[1:])).min(), np.median(wall[1:]), 'bit-identical' if key == out['ref'] else 'DIFFERS'), flush=True)
    e.close()
