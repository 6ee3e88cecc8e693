"""BASELINE.json's other configurations as parity cases at FULL problem size (the bench times them; here
their results are checked): C = Hartmann-6 / N 8192 / Matern-5/2 / UCB, D = d 32 / N 16384 / SE-ARD /
Thompson with 100-feature RFF draws (all 64), E = batch-BO, 8 Thompson draws on C's model.  Inputs are the
bench's own (`bench.make_workload`); the DEVICE sweeps the full 2^20 candidates of every configuration, the
oracle a sub-sample of 2048 grid points plus the device's top-k."""
import numpy as np
import pytest

import bench
from oracle import gp_ref
from helpers import s2_tol, mu_tol

pytestmark = pytest.mark.gpu

M = 1 << 20


def _fit_pair(w):
    from pybo_amd._lib import Engine
    e = Engine(0)
    e.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
    ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    return e, ref


def test_config_c_matern_ucb_full_size():
    w = bench.make_workload('c', M)
    e, ref = _fit_pair(w)
    beta = bench.ucb_beta(w['N'])
    r = e.sweep('ucb', beta, w['Xc'], k=10, want_moments=True)
    rho = w['rho']
    assert np.all(r['s2'] > 0) and np.all(r['s2'] <= rho * (1 + 1e-12))
    np.testing.assert_array_equal(r['top_idx'], gp_ref.topk_desc(r['acq'], 10))
    # UCB is formed from the device's own moments exactly as the policy does (pybo/policies/simple.py:62-66)
    np.testing.assert_allclose(r['acq'], r['mu'] + np.sqrt(beta * r['s2']), rtol=1e-14, atol=0)
    pick = np.unique(np.concatenate([np.arange(0, M, 512), r['top_idx']]))
    mr, sr = ref.predict(w['Xc'][pick])
    assert np.all(np.abs(r['mu'][pick] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(r['s2'][pick] - sr) <= s2_tol(sr, rho))
    want = mr + np.sqrt(beta * sr)
    np.testing.assert_allclose(r['acq'][pick], want, rtol=1e-6, atol=1e-9 * np.sqrt(rho))
    # the candidate the device selects is the oracle's best among the sub-sample
    assert r['top_idx'][0] == pick[int(np.argmax(want))]
    e.close()


def test_config_d_fit_and_thompson_full_size():
    """N = 16384, d = 32: the fit through its defining equations (K + sn2 I) alpha = y - bias and
    L L^T = K + sn2 I (probed with random vectors), then the Thompson path against the oracle's draws."""
    from pybo_amd._lib import Engine
    w = bench.make_workload('d', M)
    N, d, rho, sn2, bias = w['N'], w['d'], w['rho'], w['sn2'], w['bias']
    e = Engine(0)
    e.fit(w['X'], w['y'], 'se', w['ell'], rho, sn2, bias)
    K = gp_ref.kernel(gp_ref.SE_ARD, w['X'], w['X'], w['ell'], rho)
    K[np.diag_indices(N)] += sn2
    a, alpha = e.get_vectors()
    r = w['y'] - bias
    res = K @ alpha - r
    assert np.linalg.norm(res) <= 1e-11 * (np.linalg.norm(K, 'fro') * np.linalg.norm(alpha) + np.linalg.norm(r))
    L = e.get_matrix('L')
    P = np.random.RandomState(0).randn(N, 4)
    assert np.linalg.norm(L @ (L.T @ P) - K @ P) <= 1e-13 * np.linalg.norm(K, 'fro') * np.linalg.norm(P)
    np.testing.assert_allclose(L @ a, r, rtol=0, atol=1e-11 * np.linalg.norm(r))
    del K, L
    # Thompson: the bench's draw order (seeds 100 + s), ALL 64 draws over the full grid
    S, n = 64, 100
    Ws, bs, zs = [], [], []
    for s in range(S):
        rng = np.random.RandomState(100 + s)
        Wd, bd = gp_ref.rff_draw_spectral(gp_ref.SE_ARD, n, d, w['ell'], rng)
        Ws.append(Wd); bs.append(bd); zs.append(rng.randn(n))
    Ws, bs = np.array(Ws), np.array(bs)
    As, vs = e.rff_gram_batch(Ws, bs)
    ths = []
    for q in range(S):
        C = np.cos(w['X'] @ Ws[q].T + bs[q])
        np.testing.assert_allclose(As[q], C.T @ C, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(vs[q], C.T @ r, rtol=1e-11, atol=1e-9)
        ths.append(gp_ref.rff_posterior_theta(As[q], vs[q], n, rho, sn2, zs[q]))
    out = e.rff_sweep(Ws, bs, np.array(ths), bias, w['Xc'], k=3)
    assert out['vals'].shape == (S, M)
    for q in range(S):
        # oracle on 2048 grid points + the three candidates the device ranked first for this draw
        pick = np.unique(np.concatenate([np.arange(0, M, 512), out['top_idx'][q]]))
        want = gp_ref.RFFSample(Ws[q], bs[q], ths[q], bias).get(w['Xc'][pick])
        np.testing.assert_allclose(out['vals'][q][pick], want, rtol=1e-9, atol=1e-9 * np.sqrt(rho))
        # top-k of the full grid = ranking of the returned values (value descending, index ascending)
        v = out['vals'][q]
        best = np.argpartition(-v, 8)[:8]
        best = best[np.lexsort((best, -v[best]))][:3]
        np.testing.assert_array_equal(out['top_idx'][q], best)
        assert out['top_idx'][q][0] == pick[int(np.argmax(want))]
    assert len(set(out['top_idx'][:, 0].tolist())) > 8          # 64 draws do not all agree
    e.close()


def test_stream_schedule_above_the_task_graph_limit_full_size():
    """N = 24576 = 192 blocks: above "chol_tg_max" (160) the factorisation runs on the STREAM schedule (kernels_fit.hip: ~250
    launches over four streams) -- the path every fit took before round 4 and still the fallback of an aborted task-graph
    launch.  The fit through its defining equations (probed with random vectors, as for config D), then 2048 candidates'
    moments and the EI ranking against the oracle."""
    from pybo_amd._lib import Engine
    N, d = 24576, 8
    rng = np.random.RandomState(24576)
    X = rng.rand(N, d)
    y = -((X - 0.4) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell, rho, bias = 0.25 * np.ones(d), float(np.var(y)), float(np.mean(y))
    sn2 = 1e-4 * rho
    e = Engine(0)
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    assert e.timers()['chol_fallbacks'] == 0                   # not a fallback: the schedule chosen by size
    K = gp_ref.kernel(gp_ref.SE_ARD, X, X, ell, rho)
    K[np.diag_indices(N)] += sn2
    a, alpha = e.get_vectors()
    r = y - bias
    res = K @ alpha - r
    assert np.linalg.norm(res) <= 1e-11 * (np.linalg.norm(K, 'fro') * np.linalg.norm(alpha) + np.linalg.norm(r))
    L = e.get_matrix('L')
    P = np.random.RandomState(0).randn(N, 4)
    assert np.linalg.norm(L @ (L.T @ P) - K @ P) <= 1e-13 * np.linalg.norm(K, 'fro') * np.linalg.norm(P)
    np.testing.assert_allclose(L @ a, r, rtol=0, atol=1e-11 * np.linalg.norm(r))
    del K, L
    ref = gp_ref.make_gp(sn2, rho, ell, bias, 'se')
    ref.add_data(X, y)
    Z = np.random.RandomState(1).rand(2048, d)
    target = float(e.mean_at_obs()[1])
    assert abs(target - ref.mean_at_obs().max()) <= 1e-9
    out = e.sweep('ei', target, Z, k=10, want_moments=True)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(out['mu'] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(out['s2'] - sr) <= s2_tol(sr, rho))
    want = ref.get_improvement(target, Z)
    big = np.abs(want) > 1e-12 * np.abs(want).max()
    np.testing.assert_allclose(out['acq'][big], want[big], rtol=1e-6)
    assert out['top_idx'][0] == int(np.argmax(want))
    e.close()


def test_config_e_batch_of_thompson_recommendations():
    """q = 8 draws on the Matern-5/2 model of config C: eight distinct recommendations, each the argmax of
    its own posterior sample over the grid (checked against the host evaluation of the same sample)."""
    w = bench.make_workload('e', M)
    e, ref = _fit_pair(w)
    q, n = 8, 100
    samples = [ref.sample_f(n, rng=100 + s) for s in range(q)]
    Ws = np.array([s.W for s in samples]); bs = np.array([s.b for s in samples])
    As, vs = e.rff_gram_batch(Ws, bs)
    r = w['y'] - w['bias']
    ths = []
    for s in range(q):
        C = np.cos(w['X'] @ Ws[s].T + bs[s])
        np.testing.assert_allclose(As[s], C.T @ C, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(vs[s], C.T @ r, rtol=1e-11, atol=1e-9)
        # the weight posterior from the DEVICE Gram reproduces the oracle's draw (same rng order)
        z = np.random.RandomState(100 + s)
        gp_ref.rff_draw_spectral(ref.kid, n, w['d'], w['ell'], z)
        th = gp_ref.rff_posterior_theta(As[s], vs[s], n, w['rho'], w['sn2'], z.randn(n))
        np.testing.assert_allclose(th, samples[s].theta, rtol=1e-6, atol=1e-8)
        ths.append(samples[s].theta)
    out = e.rff_sweep(Ws, bs, np.array(ths), w['bias'], w['Xc'], k=1, want_all=False)
    picks = out['top_idx'][:, 0]
    for s in range(q):
        host = np.concatenate([samples[s].get(w['Xc'][m0:m0 + (1 << 17)]) for m0 in range(0, M, 1 << 17)])
        assert picks[s] == int(np.argmax(host))
        assert abs(out['top_val'][s, 0] - host[picks[s]]) <= 1e-9 * max(1.0, abs(host[picks[s]]))
    assert len(set(picks.tolist())) > 1
    e.close()
