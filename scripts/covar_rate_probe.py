"""Covariance-evaluation rate of the two kernels that are bound by it (k_cross_gram in a cold sweep, k_sweep_rankq in a
warm step) against the input dimension and the covariance family: slope = cost of a dimension of the squared distance,
intercept = cost of the covariance function itself.
    python scripts/covar_rate_probe.py [--n 8192] [--m 262144]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8192)
    ap.add_argument('--m', type=int, default=1 << 18)
    a = ap.parse_args()
    from pybo_amd._lib import Engine
    rng = np.random.RandomState(0)
    e = Engine(0)
    e.set_option('sweep_cache', 1)
    print('N = %d, M = %d: ns per 1000 covariance evaluations (whole chip), cross-Gram / warm correction' % (a.n, a.m))
    for kern in ('se', 'matern5'):
        for d in (1, 2, 4, 8, 16, 32):
            X = rng.rand(a.n, d)
            y = np.sin(X.sum(axis=1))
            Z = rng.rand(a.m, d)
            e.fit(X, y, kern, np.full(d, 0.4 * np.sqrt(d)), 1.0, 1e-3, 0.0)
            e.sweep('ei', 0.5, Z, k=8, want_all=False)            # warm-up + fills the sweep cache
            e.timers(reset=True)
            e.sweep('ei', 0.5, Z, k=8, want_all=False)
            t = e.timers()
            cg = t['cross_gram']
            e.timers(reset=True)
            e.append(rng.rand(d), 0.1)
            e.sweep_update("ei", 0.5, k=8, want_all=False)
            t = e.timers()
            ev = a.n * a.m
            print('  %-8s d = %2d   cross-Gram %7.3f ms = %6.2f   rank-1 %7.3f ms = %6.2f' %
                  (kern, d, cg, cg * 1e6 / ev * 1e3, t['rank1'], t['rank1'] * 1e6 / ev * 1e3))
    e.close()


if __name__ == '__main__':
    main()
