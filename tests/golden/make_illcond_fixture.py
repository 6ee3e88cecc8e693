"""Long-double truth for the ILL-CONDITIONED regime: the reference initialises its GP with
sn2 = 1e-6 (pybo/bayesopt.py:98), i.e. cond(K + sn2 I) ~ N rho / sn2 ~ 1e9..1e10, where an explicit triangular inverse
(what the HIP sweep uses) and substitution (what oracle/gp_ref.py uses) may part ways.  For the inputs of BASELINE
config B (N = 2048) and of the north-star workload (N = 8192), each with sn2 = 1e-6 * rho and with the literal
sn2 = 1e-6, this script evaluates the exact-GP posterior moments at 256 candidates in 80-bit long double
(eps = 1.1e-19):

    K (long double, exp in long double)  ->  W = K^-1 [k*_1 .. k*_256, y - bias]  by iterative refinement: fp64 Cholesky
    as the preconditioner, residuals B - K W in long double, until the correction is below 1e-19 relative
    (contraction ~ cond * eps64 ~ 1e-6 per sweep)  ->  mu = bias + k*.alpha,  s2 = rho - k*.w   in long double.

At N = 512 the same routine is checked against a direct long-double Cholesky (below) before anything is written.
Candidates: 192 points of the workload's Sobol grid (every 5461st) + 64 observed points displaced by 1e-3 length-scales
(where s2 ~ sn2 and the cancellation rho - q is worst).  Output: tests/golden/illcond_ld.npz (float64 roundings of the
long-double results; ~30 KB).  Run once in the build container (~10 min on 8 vCPU):  python tests/golden/make_illcond_fixture.py
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402

ld = np.longdouble
_K = None


def kern_ld(A, B, rho):
    r2 = np.zeros((len(A), len(B)), dtype=ld)
    for k in range(A.shape[1]):
        df = A[:, k][:, None] - B[:, k][None, :]
        r2 += df * df
    return ld(rho) * np.exp(-r2 / ld(2))


def _matmul_cols(cols):
    return _K @ cols


def matmul_ld(K, W, pool):
    if pool is None:
        return K @ W
    parts = np.array_split(np.arange(W.shape[1]), pool._processes)
    outs = pool.map(_matmul_cols, [np.ascontiguousarray(W[:, p]) for p in parts if len(p)])
    return np.concatenate(outs, axis=1)


def solve_refined(K, B, pool, tag=''):
    """K^-1 B to long-double accuracy: fp64 Cholesky preconditioner + long-double residuals."""
    cf = sla.cho_factor(K.astype(np.float64), lower=True)
    W = sla.cho_solve(cf, B.astype(np.float64)).astype(ld)
    # The corrections contract by ~cond * eps64 per sweep until they reach the noise floor of the long-double residual
    # (eps_ld * |K| |W| seen through K^-1: the kriging weights themselves are only determined to ~cond * eps_ld; the
    # quadratic forms k*.w and k*.alpha built from them are far better determined, which the N = 512 self-check
    # against a direct long-double factorisation confirms).  Stop one sweep after the corrections stop shrinking.
    prev = np.inf
    for it in range(12):
        R = B - matmul_ld(K, W, pool)
        dW = sla.cho_solve(cf, R.astype(np.float64)).astype(ld)
        W = W + dW
        rel = float(np.max(np.abs(dW)) / np.max(np.abs(W)))
        print('   %s refinement %d: max |dW| / max |W| = %.2e' % (tag, it, rel), flush=True)
        if rel > 0.1 * prev or rel < 1e-19:
            return W
        prev = rel
    raise RuntimeError('refinement did not converge')


def posterior_ld(X, y, ell, rho, sn2, bias, Z, pool, tag=''):
    global _K
    Xs, Zs = X.astype(ld) / ell.astype(ld), Z.astype(ld) / ell.astype(ld)
    K = kern_ld(Xs, Xs, rho)
    K[np.diag_indices_from(K)] += ld(sn2)
    Ks = kern_ld(Xs, Zs, rho)
    _K = K
    B = np.concatenate([Ks, (y.astype(ld) - ld(bias))[:, None]], axis=1)
    W = solve_refined(K, B, pool, tag)
    mu = ld(bias) + Ks.T @ W[:, -1]
    s2 = ld(rho) - np.sum(Ks * W[:, :-1], axis=0)
    return mu, s2


def direct_ld(X, y, ell, rho, sn2, bias, Z):
    Xs, Zs = X.astype(ld) / ell.astype(ld), Z.astype(ld) / ell.astype(ld)
    K = kern_ld(Xs, Xs, rho)
    N = len(K)
    K[np.diag_indices_from(K)] += ld(sn2)
    L = np.zeros_like(K)
    for j in range(N):
        L[j, j] = np.sqrt(K[j, j] - (L[j, :j] ** 2).sum())
        L[j + 1:, j] = (K[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    Ks = kern_ld(Xs, Zs, rho)
    V = np.zeros_like(Ks)
    a = np.zeros(N, dtype=ld)
    r = y.astype(ld) - ld(bias)
    for i in range(N):
        V[i] = (Ks[i] - L[i, :i] @ V[:i]) / L[i, i]
        a[i] = (r[i] - L[i, :i] @ a[:i]) / L[i, i]
    return ld(bias) + V.T @ a, ld(rho) - (V * V).sum(0)


def candidates(w):
    rng = np.random.RandomState(2026)
    grid = w['Xc'][::5461][:192]
    near = w['X'][rng.choice(w['N'], 64, replace=False)] + 1e-3 * w['ell'] * rng.randn(64, w['d'])
    return np.clip(np.concatenate([grid, near]), w['lo'], w['hi'])


def main():
    out = {}
    # self-check of the refinement against a direct long-double factorisation (N = 512, the worst noise level)
    w = bench.make_workload('b', 1 << 20)
    Z = candidates(w)
    sub = slice(0, 512)
    mu_r, s2_r = posterior_ld(w['X'][sub], w['y'][sub], w['ell'], w['rho'], 1e-6 * w['rho'], w['bias'], Z[:32], None, 'check')
    mu_d, s2_d = direct_ld(w['X'][sub], w['y'][sub], w['ell'], w['rho'], 1e-6 * w['rho'], w['bias'], Z[:32])
    err_mu = float(np.max(np.abs(mu_r - mu_d)) / np.sqrt(w['rho']))
    err_s2 = float(np.max(np.abs(s2_r - s2_d)) / w['rho'])
    print('self-check N=512: |mu_refined - mu_direct| / sqrt(rho) = %.1e, |s2 ...| / rho = %.1e' % (err_mu, err_s2))
    assert err_mu < 1e-13 and err_s2 < 1e-15, 'refinement and direct long-double factorisation disagree'
    out['selfcheck'] = np.array([err_mu, err_s2])
    for name in ('b', 'ns'):
        w = bench.make_workload(name, 1 << 20)
        Z = candidates(w)
        out['Z_' + name] = Z
        for label, sn2 in (('rel', 1e-6 * w['rho']), ('lit', 1e-6)):
            t0 = time.time()
            # the kernel matrix must exist before the workers fork (they read the module global): build it, then open the pool
            Xs = w['X'].astype(ld) / w['ell'].astype(ld)
            global _K
            _K = kern_ld(Xs, Xs, w['rho'])
            _K[np.diag_indices_from(_K)] += ld(sn2)
            with mp.get_context('fork').Pool(8) as pool:
                Zs = Z.astype(ld) / w['ell'].astype(ld)
                Ks = kern_ld(Xs, Zs, w['rho'])
                B = np.concatenate([Ks, (w['y'].astype(ld) - ld(w['bias']))[:, None]], axis=1)
                W = solve_refined(_K, B, pool, '%s/%s' % (name, label))
            mu = ld(w['bias']) + Ks.T @ W[:, -1]
            s2 = ld(w['rho']) - np.sum(Ks * W[:, :-1], axis=0)
            out['mu_%s_%s' % (name, label)] = mu.astype(np.float64)
            out['s2_%s_%s' % (name, label)] = s2.astype(np.float64)
            out['sn2_%s_%s' % (name, label)] = float(sn2)
            print('%s sn2=%s (%.3g = %.2g rho): s2/rho in [%.2e, %.2e]   %.0f s'
                  % (name, label, sn2, sn2 / w['rho'], float(s2.min() / w['rho']), float(s2.max() / w['rho']),
                     time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, 'illcond_ld.npz'), **out)


if __name__ == '__main__':
    main()
