"""Full-grid oracle fixtures: the CPU restatement oracle/gp_ref.py evaluated over ALL 2^20
Sobol candidates of BASELINE config B (bench.make_workload('b')) and over the first 2^16 candidates of config C, run
once in the build container (a few minutes on 8 vCPU) and committed as data:

    tests/golden/grid_b.npz   ei (2^20,) float64 [92 % exact zeros: compresses to ~1 MB], top (256,) indices in the
                              deterministic order (value desc, index asc), mu / s2 at every 16th grid point, target,
                              sha256 of the inputs
    tests/golden/grid_c.npz   ucb (2^16,), mu, s2 (2^16,), top (256,), beta, sha256 of the inputs

tests/test_gpu_fullsize.py compares the WHOLE device EI array and the device's top-64 ORDER against these, instead
of a stride-512 sub-sample.  The oracle is test infrastructure (oracle/gp_ref.py header); nothing here is imported by
the product.  Usage:  python tests/golden/make_grid_fixtures.py [b] [c]
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bench                      # noqa: E402
from oracle import gp_ref         # noqa: E402


def digest(w, M):
    h = hashlib.sha256()
    for a in (w['X'], w['y'], w['Xc'][:M], w['ell'], np.array([w['rho'], w['sn2'], w['bias']])):
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def make_b():
    M = 1 << 20
    w = bench.make_workload('b', M)
    ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    target = float(ref.mean_at_obs().max())
    t0 = time.time()
    mu, s2 = ref.predict(w['Xc'])
    s = np.sqrt(s2)
    z = (mu - target) / s
    ei = (mu - target) * gp_ref.norm_cdf(z) + s * gp_ref.norm_pdf(z)      # = GPRef.get_improvement(target, Xc)
    print('config B: oracle over %d candidates in %.1f s; EI max %.6g, zeros %.1f %%'
          % (M, time.time() - t0, ei.max(), 100.0 * np.mean(ei == 0)))
    np.savez_compressed(os.path.join(HERE, 'grid_b.npz'), ei=ei, top=gp_ref.topk_desc(ei, 256),
                        mu16=mu[::16], s216=s2[::16], target=target, sha=digest(w, M))


def make_c():
    M = 1 << 16
    w = bench.make_workload('c', M)
    ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    beta = float(bench.ucb_beta(w['N']))
    t0 = time.time()
    mu, s2 = ref.predict(w['Xc'])
    ucb = mu + np.sqrt(beta * s2)
    print('config C: oracle over %d candidates in %.1f s' % (M, time.time() - t0))
    np.savez_compressed(os.path.join(HERE, 'grid_c.npz'), ucb=ucb, mu=mu, s2=s2, top=gp_ref.topk_desc(ucb, 256),
                        beta=beta, sha=digest(w, M))


def make_full(name):
    """North-star workload / config C over ALL 2^20 candidates (~10-20 min on 8 vCPU each): the oracle's ranking of the
    whole grid (top 256) and every 8th value + moments, so that the device's selected candidate is checked against the
    oracle's argmax over the WHOLE grid, not over a sub-sample.  -> tests/golden/grid_<name>_full.npz (~3 MB)"""
    M = 1 << 20
    w = bench.make_workload(name, M)
    ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    t0 = time.time()
    mu, s2 = ref.predict(w['Xc'])
    if w['acq'] == 'ei':
        param = float(ref.mean_at_obs().max())
        s = np.sqrt(s2)
        z = (mu - param) / s
        val = (mu - param) * gp_ref.norm_cdf(z) + s * gp_ref.norm_pdf(z)
    else:
        param = float(bench.ucb_beta(w['N']))
        val = mu + np.sqrt(param * s2)
    print('%s: oracle over %d candidates in %.1f s; max %.9g at %d' % (name, M, time.time() - t0, val.max(), int(np.argmax(val))))
    top = gp_ref.topk_desc(val, 256)
    np.savez_compressed(os.path.join(HERE, 'grid_%s_full.npz' % name), val8=val[::8], mu8=mu[::8], s28=s2[::8], top=top,
                        top_val=val[top], param=param, vmax=float(val.max()), sha=digest(w, M))


if __name__ == '__main__':
    which = sys.argv[1:] or ['b', 'c']
    for name in ('ns', 'c'):
        if name + '_full' in which:
            make_full(name)
    if 'b' in which:
        make_b()
    if 'c' in which:
        make_c()
