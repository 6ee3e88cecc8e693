"""Where do the tasks of the task-graph Cholesky wait?  Per-task log (option chol_tg_trace = 2) -> for every task the time
its dependencies completed (from the log itself), the time it started, its duration; summarised per phase of the
factorisation.   python scripts/tg/tg_tasklog.py N [opt=v ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pybo_amd._lib import Engine

N = int(sys.argv[1])
rng = np.random.RandomState(N)
X = rng.rand(N, 8); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(8); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
e = Engine(0)
e.set_option('chol_tg', 1); e.set_option('chol_tg_trace', 2); e.set_option('chol_tg_tmo_ms', 500)
for kv in sys.argv[2:]:
    k, v = kv.split('='); e.set_option(k, int(v))
for r in range(3):
    e.timers(reset=True)
    e.fit(X, y, 'se', ell, rho, sn2, bias, stage=2); e.sync()
    tm = e.timers(reset=True)['cholesky']
nP = (N + 127) // 128
diag, crit = e.chol_trace(nP, full_log=True)
log = e.last_chol_tasklog
prof = e.last_chol_profile
t0 = None
recs = []
for wg in range(1024):
    n = int(prof[wg, 0])
    if prof[wg, 6] == 0 or n == 0:
        continue
    r = log[wg, :min(n, 1024)]
    task = r[:, :2].copy().view(np.int16).reshape(-1, 8)
    for i in range(len(r)):
        recs.append((int(task[i, 0]), int(task[i, 1]), int(task[i, 2]), int(task[i, 3]), int(task[i, 4]), int(task[i, 6]), int(r[i, 2]), int(r[i, 3]), wg))
recs = np.array(recs, dtype=np.int64)
# stamps relative to the diagonal trace's origin: chol_trace() subtracted out[0]; recover it from the critical records
e2 = e
raw0 = None
# (crit[p, i] are relative microseconds; the log is raw ticks: align through the first panel-solve record)
raw0 = float(e.last_chol_trace_origin)
ts = (recs[:, 6] - raw0) / 100.0
te = (recs[:, 7] - raw0) / 100.0
typ, I, J, k0, k1, aux = (recs[:, i] for i in range(6))
print('N = %d: cholesky %.3f ms; %d tasks logged' % (N, tm, len(recs)))
# completion tables
upd_end = {}     # (I, J, k1) -> end
trsm_end = {}    # (p, J, h) -> end
for i in range(len(recs)):
    if typ[i] == 2: upd_end[(I[i], J[i], k1[i])] = te[i]
    elif typ[i] == 1: trsm_end[(I[i], J[i], aux[i])] = te[i]
quad_end = {}
for i in range(len(recs)):
    if typ[i] == 3: quad_end[I[i]] = max(quad_end.get(I[i], 0), te[i])
potrf_end = diag[:, 2]
def col_solved(Jc, k):       # time block row k-1 of column Jc was solved
    if k == 0: return 0.0
    return max(trsm_end.get((k - 1, Jc, 0), 0.0), trsm_end.get((k - 1, Jc, 1), 0.0))
ready = np.zeros(len(recs))
for i in range(len(recs)):
    if typ[i] == 1:
        p = I[i]
        r = potrf_end[p]
        if p > 0: r = max(r, upd_end.get((p, J[i], p), 0.0))
    else:
        r = max(col_solved(I[i], k1[i]), col_solved(J[i], k1[i]))
        if k0[i] > 0: r = max(r, upd_end.get((I[i], J[i], k0[i]), 0.0))
    ready[i] = r
wait = ts - ready
dur = te - ts
K = np.maximum(k1 - k0, 1)
print('phase (by pivot block of the task = k1-1 or p) | tasks | wait after ready: median / p90 / max | duration median (per block)')
piv = np.where(typ == 1, I, k1 - 1)
for lo in range(0, nP, max(nP // 8, 1)):
    hi = lo + max(nP // 8, 1)
    for name, m in (('solve', typ == 1), ('update K=1', (typ == 2) & (K == 1)), ('update K>1', (typ == 2) & (K > 1)), ('critical', typ == 3)):
        mm = m & (piv >= lo) & (piv < hi)
        if mm.sum() == 0: continue
        print('  blocks %3d-%3d %-11s %6d   wait %7.1f %7.1f %7.1f   dur %6.1f (%.1f per block)' % (
            lo, hi - 1, name, mm.sum(), np.median(wait[mm]), np.percentile(wait[mm], 90), wait[mm].max(), np.median(dur[mm]), np.median(dur[mm] / K[mm])))
# utilisation over time
edges = np.linspace(0, diag[-1, 2], 17)
busy = np.zeros(16)
for i in range(len(recs)):
    a, b = ts[i], te[i]
    lo = np.searchsorted(edges, a) - 1
    hi = np.searchsorted(edges, b) - 1
    for s in range(max(lo, 0), min(hi, 15) + 1):
        busy[s] += max(0.0, min(b, edges[s + 1]) - max(a, edges[s]))
nw = int((prof[:, 6] != 0).sum())
print('busy fraction of %d workgroups per 1/16 of the run:' % nw, ' '.join('%.2f' % (busy[s] / (edges[s + 1] - edges[s]) / nw) for s in range(16)))
print('diagonal block reached at (us):', ' '.join('%d:%.0f' % (p, diag[p, 1]) for p in range(0, nP, max(nP // 16, 1))))
e.close()

# the lag-2 chain of the shadows: per block row p, relative to the end of potrf(p): the workers' solve of tile (p, p+3) and the
# last update of tile (p+1, p+3) (role S2's right-hand side of the NEXT block row), and of tile (p+1, p+1)'s chunk that ends at p
print()
print('per block row p, relative to the end of potrf(p) [us]: solve (p,p+3) halves start/end | final update of tile (p+1,p+3) start/end | '
      'update of (p+2,p+2) ending at block row p+1: start/end')
rows = {}
for i in range(len(recs)):
    rows.setdefault((int(typ[i]), int(I[i]), int(J[i]), int(k1[i])), []).append((ts[i], te[i]))
for p in range(2, nP - 4):
    e0 = diag[p, 2]
    a = rows.get((1, p, p + 3, 0), [])
    b = rows.get((2, p + 1, p + 3, p + 1), [])
    c = rows.get((2, p + 2, p + 2, p + 1), [])
    fmt = lambda L: ' '.join('%+.1f/%+.1f' % (x - e0, y - e0) for x, y in L) if L else '-'
    print('p=%2d  %s | %s | %s' % (p, fmt(a), fmt(b), fmt(c)))
