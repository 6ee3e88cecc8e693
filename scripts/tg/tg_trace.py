"""Critical-path stamps of the task-graph Cholesky (its own clock): python scripts/tg/tg_trace.py N [opt=v ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine

N = int(sys.argv[1])
rng = np.random.RandomState(N)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
e.set_option('chol_tg', 1); e.set_option('chol_tg_trace', 1); e.set_option('chol_tg_tmo_ms', 500)
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, int(v))
for r in range(3):
    e.timers(reset=True)
    e.fit(X, y, 'se', ell, rho, sn2, bias, stage=2); e.sync()
    tm = e.timers(reset=True)['cholesky']
nP = (N + 127) // 128
diag, crit = e.chol_trace(nP)
print('N = %d: cholesky %.3f ms (HIP events); kernel-side span %.1f us' % (N, tm, diag[-1, 2]))
print('per diagonal block, microseconds (kernel clock): wait = start - ready-to-wait, potrf = end - start;')
per = np.diff(diag[:, 1])
print('block period: median %.1f  mean %.1f  min %.1f  max %.1f us' % (np.median(per), per.mean(), per.min(), per.max()))
print('potrf: median %.1f  max %.1f;  wait before potrf: median %.1f  max %.1f' % (
    np.median(diag[:, 2] - diag[:, 1]), (diag[:, 2] - diag[:, 1]).max(), np.median(diag[1:, 1] - diag[1:, 0]), (diag[1:, 1] - diag[1:, 0]).max()))
rows = list(range(min(nP - 1, 6))) + list(range(max(6, nP // 2 - 2), min(nP - 1, nP // 2 + 2))) + list(range(max(nP - 5, 6), nP - 1))
print('the workgroups that follow block row p, relative to the end of potrf(p): S1 / S2 = [waiting for the right-hand sides from .. '
      'loaded at .. last rows stored at], U = [waiting from .. earlier chunks in at .. tile loaded at .. stored at], '
      'V (tile (p+1, p+2), S1\'s right-hand sides of the NEXT block row) = [waiting from .. earlier chunks in at .. stored at]')
for p in rows:
    e0 = diag[p, 2]
    ts = crit[p]
    print('p=%3d potrf %6.1f..%6.1f (%.1f)  S1 %+6.1f %+6.1f %+6.1f   S2 %+6.1f %+6.1f %+6.1f   U %+6.1f %+6.1f %+6.1f %+6.1f   V %+6.1f %+6.1f %+6.1f   next potrf start +%.1f' % (
        p, diag[p, 1], diag[p, 2], diag[p, 2] - diag[p, 1], ts[0, 0] - e0, ts[0, 1] - e0, ts[1, 1] - e0,
        ts[2, 0] - e0, ts[2, 1] - e0, ts[3, 1] - e0, ts[4, 0] - e0, ts[4, 1] - e0, ts[5, 0] - e0, ts[5, 1] - e0,
        ts[6, 0] - e0, ts[6, 1] - e0, ts[7, 1] - e0, diag[p + 1, 1] - e0))
prof = e.last_chol_profile
w = prof[prof[:, 6] == 2]
if len(w):
    tot = diag[-1, 2]
    print('workers: %d  tasks/worker median %d  (us per worker, median): take %.0f  update tasks %.0f  solve tasks %.0f  publish %.0f  of %.0f total;  '
          'blocks applied/worker %d -> %.1f us per block-update incl. overheads' % (
        len(w), np.median(w[:, 0]), np.median(w[:, 1]) / 100, np.median(w[:, 2]) / 100, np.median(w[:, 3]) / 100, np.median(w[:, 4]) / 100, tot,
        np.median(w[:, 5]), w[:, 2].sum() / max(w[:, 5].sum(), 1) / 100))
e.close()
