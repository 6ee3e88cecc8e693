"""Host-side phases of bench.py's Thompson step (configs D / E): where the wall time beyond the device stages goes."""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from pybo_amd._lib import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else 'e'
w = bench.make_workload(wl, 1 << 20)
N, d, M = w['N'], w['d'], w['M']
S = 8 if wl == 'e' else 64
dev = torch.device('cuda', 0)
dX = torch.from_numpy(w['X']).to(dev); dy = torch.from_numpy(w['y']).to(dev); dXc = torch.from_numpy(w['Xc']).to(dev)
eng = Engine(0, None)
acc = {}
def lap(name, t0):
    t1 = time.perf_counter(); acc.setdefault(name, []).append(t1 - t0); return t1
for it in range(8):
    torch.cuda.synchronize(dev)
    t = t00 = time.perf_counter()
    th = threading.Thread(target=lambda: eng.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias']))
    th.start(); t = lap('thread start', t)
    Ws, bs, zs = [], [], []
    for s in range(S):
        Wd, bd, zd = bench.thompson_draw(w, s); Ws.append(Wd); bs.append(bd); zs.append(zd)
    t = lap('host draws', t)
    th.join(); t = lap('join (fit)', t)
    Wa, ba, za = np.array(Ws), np.array(bs), np.array(zs); t = lap('np.array', t)
    ths = eng.rff_posterior(Wa, ba, za, np.sqrt(2.0 * w['rho'] / 100)); t = lap('rff_posterior', t)
    tv, ti = eng.rff_sweep_dev(Wa, ba, ths, w['bias'], dXc.data_ptr(), M, 1); t = lap('rff_sweep_dev', t)
    acc.setdefault('total', []).append(t - t00)
tm = eng.timers(reset=True)
for k, v in acc.items():
    print('%-16s %8.3f ms' % (k, 1e3 * np.median(v[2:])))
print({k: round(v / 8, 3) for k, v in tm.items() if k in ('gram', 'cholesky', 'rff', 'rff_sweep') and v > 0})
