"""A/B of the Thompson sweep kernels alone (option x_rff) at the shapes of configs D and E: python scripts/rff_probe/rff_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pybo_amd._lib import Engine

def run(d, S, n, M, variants, reps=4):
    rng = np.random.RandomState(0)
    X = rng.rand(256, d); y = rng.randn(256)
    e = Engine(0)
    e.fit(X, y, 'se', 0.5 * np.ones(d), 1.0, 1e-2, 0.0, stage=2)
    W = rng.randn(S, n, d) * 2.0; b = rng.rand(S, n) * 2 * np.pi; th = rng.randn(S, n) * 0.1
    Xc = rng.rand(M, d)
    ref = None
    for v in variants:
        e.set_option('x_rff', v)
        ts = []
        for r in range(reps):
            e.timers(reset=True)
            out = e.rff_sweep(W, b, th, 0.3, Xc, k=1, want_all=(r == 0))
            ts.append(e.timers(reset=True)['rff_sweep'])
            if r == 0:
                vals = out['vals']
        if ref is None:
            ref = vals
        # exact values on a few candidates, in long double
        pick = np.arange(0, M, M // 64)
        z = np.einsum('snd,md->smn', W.astype(np.longdouble), Xc[pick].astype(np.longdouble)) + b[:, None, :]
        want = (0.3 + (np.cos(z) * th[:, None, :]).sum(-1)).astype(float)
        err = np.max(np.abs(vals[:, pick] - want))
        ops = S * n * (d + 20.0) * M
        t = np.median(ts[1:])
        print('d=%2d S=%2d n=%3d M=2^%d x_rff=%d: %.3f ms  frac %.3f  max |err| vs long double %.2e  max |diff| vs first variant %.2e'
              % (d, S, n, int(np.log2(M)), v, t, ops / (t * 1e-3) / (256 * 4 * 16 * 2.4e9), err, np.max(np.abs(vals - ref))), flush=True)
    e.close()

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 0]
run(32, 64, 100, 1 << 20, variants)
run(6, 8, 100, 1 << 20, variants)

run(8, 16, 128, 1 << 18, variants)
run(3, 4, 37, 1 << 18, variants)
run(6, 64, 100, 1 << 20, variants)
run(6, 8, 96, 1 << 20, variants)
run(32, 64, 96, 1 << 20, variants)
run(32, 64, 112, 1 << 20, variants)
