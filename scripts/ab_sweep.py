"""Interleaved A/B of sweep-kernel variants (tile_order option: bit0 = tile map, bits1.. = k-loop variant)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 17)
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [10, 22]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rng = np.random.RandomState(1)
X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
ell = 0.25 * np.ones(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-4 * rho
Xc = rng.rand(M, d)
kern = sys.argv[5] if len(sys.argv) > 5 else 'se'
e = Engine(0)
e.fit(X, y, kern, ell, rho, sn2, bias)
ref = None
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        e.set_option('tile_order', v % 100)
        e.set_option('super_m', (v // 100) or 8)
        e.timers(reset=True)
        out = e.sweep('ei', 0.0, Xc, k=4, want_all=True)
        tm = e.timers(reset=True)
        if ref is None: ref = out['acq']
        same = np.array_equal(ref, out['acq'])
        if r > 0: res[v].append(tm['sweep_trmm_flop'] / tm['sweep_trmm'] / 1e9)
        if not same: print("variant", v, "DIFFERS from variant", variants[0], np.max(np.abs(ref-out['acq'])))
for v in variants:
    a = np.array(res[v]); print(f"N={N} M={M} variant {v}: TF/s median {np.median(a):.2f} min {a.min():.2f} max {a.max():.2f}")
