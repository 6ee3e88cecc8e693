"""Stage-by-stage GPU-vs-oracle check + first timings (development aid; the real tests are tests/)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
from oracle import gp_ref

def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

def check(N, d, M, kernel, sn2, seed=0, k=10):
    rng = np.random.RandomState(seed)
    X = rng.rand(N, d); y = np.sin(3 * X.sum(1)) + 0.1 * rng.randn(N)
    ell = 0.3 + 0.2 * rng.rand(d); rho = 1.3; bias = 0.2
    Xc = rng.rand(M, d)
    ref = gp_ref.GPRef(sn2, rho, ell, bias, kernel); ref.add_data(X, y)
    e = Engine(0)
    e.fit(X, y, kernel, ell, rho, sn2, bias, stage=1)
    K = e.get_matrix('K'); Kr = np.triu(ref.gram())
    print(f"[{kernel} N={N} d={d}] gram rel err {rel(K, Kr):.2e}")
    e.fit(X, y, kernel, ell, rho, sn2, bias, stage=2)
    L = e.get_matrix('L')
    print(f"   L rel err {rel(L, ref.L):.2e}  recon {np.linalg.norm(L@L.T-ref.gram())/np.linalg.norm(ref.gram()):.2e}")
    e.fit(X, y, kernel, ell, rho, sn2, bias)
    T = e.get_matrix('T')
    print(f"   T*L-I max {np.max(np.abs(T@ref.L-np.eye(N))):.2e}")
    a, alpha = e.get_vectors()
    print(f"   a rel {rel(a, ref.a):.2e} alpha rel {rel(alpha, ref.alpha()):.2e}")
    mo, mx = e.mean_at_obs()
    print(f"   mean_at_obs rel {rel(mo, ref.predict(X)[0]):.2e}")
    target = mx
    r = e.sweep('ei', target, Xc, k=k, want_all=True, want_moments=True)
    mu, s2 = ref.predict(Xc); ei = ref.get_improvement(target, Xc)
    print(f"   mu rel {rel(r['mu'], mu):.2e}  s2 max|d|/(1e-6 s2+1e-10 rho) {np.max(np.abs(r['s2']-s2)/(1e-6*s2+1e-10*rho)):.2e}  ei rel {rel(r['acq'], ei):.2e}")
    ti = gp_ref.topk_desc(r['acq'], k)
    print(f"   topk idx match(dev acq) {np.array_equal(ti, r['top_idx'])}  vs oracle acq {np.array_equal(gp_ref.topk_desc(ei,k), r['top_idx'])}")
    for acq, p in [('pi', target), ('ucb', 3.3), ('mean', None)]:
        rr = e.sweep(acq, p, Xc, k=3)
        refv = {'pi': ref.get_tail(target, Xc), 'ucb': mu + np.sqrt(3.3 * s2), 'mean': mu}[acq]
        print(f"   {acq} rel {rel(rr['acq'], refv):.2e}")
    e.close()

def timing(N, d, M, kernel='se', chunk=65536, order=0, reps=2):
    rng = np.random.RandomState(1)
    X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
    Xc = rng.rand(M, d)
    e = Engine(0); e.set_option('chunk', chunk); e.set_option('tile_order', order)
    for it in range(reps):
        t0 = time.time(); e.fit(X, y, kernel, ell, rho, sn2, bias); e.sync(); t1 = time.time()
        mo, mx = e.mean_at_obs()
        r = e.sweep('ei', mx, Xc, k=10, want_all=False); t2 = time.time()
        tm = e.timers(reset=True)
        print(f"[time N={N} d={d} M={M} chunk={chunk} order={order}] fit {t1-t0:.3f}s sweep {t2-t1:.3f}s  timers(ms): " +
              " ".join(f"{k}={v:.2f}" for k, v in tm.items() if v))
        if tm['sweep_trmm'] > 0:
            print(f"      sweep_trmm {tm['sweep_trmm_flop']/tm['sweep_trmm']/1e9:.2f} TFLOP/s (algorithmic)")
    e.close()

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'check'):
        check(300, 3, 1000, 'se', 1e-3)
        check(129, 1, 500, 'matern5', 1e-4)
        check(1000, 6, 5000, 'matern5', 1e-3)
        check(640, 8, 3000, 'matern3', 1e-3)
        check(2048, 2, 4096, 'se', 1e-4)
    if which in ('all', 'time'):
        timing(2048, 2, 1 << 17)
        timing(8192, 8, 1 << 17)
        timing(8192, 8, 1 << 17, order=1)
