"""Seeded random shapes through the engine's main entry points against the oracle: odd N, d and M (nothing a multiple of
the 16 / 64 / 128 tile edges), every covariance family, cold sweep, warm step (appends + cached re-score), mean-only and
gradient calls.  The fixed-shape parity tests cover the BASELINE configurations; this one looks for indexing mistakes in
the staging / tail code of the kernels."""
import numpy as np
import pytest

from oracle import gp_ref
from helpers import synth_problem, s2_tol, mu_tol

pytestmark = pytest.mark.gpu

KERNELS = ['se', 'matern5', 'matern3', 'matern1']


@pytest.mark.parametrize('seed', range(64))
def test_random_shapes_against_the_oracle(seed):
    from pybo_amd._lib import Engine
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([1, 2, 3, 17, 63, 64, 65, 127, 128, 129, 200, 257, 300, 511, 640]))
    d = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 40]))
    M = int(rng.choice([1, 2, 63, 127, 128, 129, 255, 1000, 1025, 4097]))
    kernel = KERNELS[seed % 4]
    nadd = int(rng.choice([0, 1, 3, 9]))
    X, y, ell = synth_problem(N + nadd, d, seed=seed)
    rho, sn2, bias = 1.0 + rng.rand(), 10.0 ** rng.uniform(-4, -2), rng.randn() * 0.3
    Z = rng.rand(M, d)
    ref = gp_ref.make_gp(sn2, rho, ell, bias, kernel)
    ref.add_data(X[:N], y[:N])
    e = Engine(0)
    e.set_option('sweep_cache', 1)
    e.fit(X[:N], y[:N], kernel, ell, rho, sn2, bias)
    k = min(8, M)
    out = e.sweep('ucb', 2.0, Z, k=k, want_moments=True)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(out['mu'] - mr) <= mu_tol(mr, rho)), (N, d, M, kernel)
    assert np.all(np.abs(out['s2'] - sr) <= s2_tol(sr, rho)), (N, d, M, kernel)
    ucb = mr + np.sqrt(2.0 * sr)
    assert abs(out['top_val'][0] - ucb.max()) <= 1e-6 * max(1.0, abs(ucb.max()))
    # warm step: nadd appended observations, cached sums corrected, grid re-scored
    for i in range(N, N + nadd):
        e.append(X[i], y[i])
        ref.add_data(X[i:i + 1], y[i:i + 1])
    if nadd:
        warm = e.sweep_update('ucb', 2.0, k=k, want_moments=True)
        mr, sr = ref.predict(Z)
        assert np.all(np.abs(warm['mu'] - mr) <= mu_tol(mr, rho)), (N, d, M, kernel, nadd)
        assert np.all(np.abs(warm['s2'] - sr) <= s2_tol(sr, rho)), (N, d, M, kernel, nadd)
    # point calls: moments + gradients, the mean alone
    P = Z[:min(M, 19)]
    got = e.predict(P, grad=True)
    want = ref.predict(P, grad=True)
    assert np.all(np.abs(got[0] - want[0]) <= mu_tol(want[0], rho))
    assert np.all(np.abs(got[1] - want[1]) <= s2_tol(want[1], rho))
    if kernel != 'matern1':          # (the exponential kernel's gradient has its kink at r = 0; covered in test_gpu_parity)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[3], want[3], rtol=1e-6, atol=1e-7)
        m1, dm1 = e.predict_mean(P, grad=True)
        np.testing.assert_allclose(dm1, want[2], rtol=1e-6, atol=1e-7)
        assert np.all(np.abs(m1 - want[0]) <= mu_tol(want[0], rho))
    e.close()


@pytest.mark.parametrize('seed', range(16))
def test_random_thompson_shapes_against_numpy(seed):
    """gpx_rff_sweep / gpx_rff_gram on odd shapes: draws S, features n (not a multiple of 16, beyond 128), d beyond the
    resident 64 (k-chunked projection), candidates not a multiple of 128 -- against the closed form in numpy."""
    from pybo_amd._lib import Engine
    rng = np.random.RandomState(5000 + seed)
    N = int(rng.choice([5, 64, 130, 300]))
    d = int(rng.choice([1, 3, 8, 31, 64, 65, 100]))
    n = int(rng.choice([1, 7, 16, 100, 127, 129, 200]))
    S = int(rng.choice([1, 2, 5]))
    M = int(rng.choice([1, 127, 129, 1000, 2049]))
    X = rng.rand(N, d)
    y = np.sin(X.sum(axis=1))
    W = rng.randn(S, n, d) * 2.0
    b = rng.rand(S, n) * 2.0 * np.pi
    th = rng.randn(S, n) * 0.1
    Z = rng.rand(M, d) * 2.0 - 1.0
    bias = 0.25
    e = Engine(0)
    e.fit(X, y, 'se', np.full(d, 0.7 * np.sqrt(d)), 1.0, 1e-3, bias)
    k = min(4, M)
    r = e.rff_sweep(W, b, th, bias, Z, k=k)
    for s in range(S):
        want = bias + np.cos(Z @ W[s].T + b[s]) @ th[s]
        scale = np.abs(th[s]).sum() + abs(bias)
        np.testing.assert_allclose(r['vals'][s], want, rtol=0, atol=1e-12 * max(scale, 1.0)), (N, d, n, S, M)
        np.testing.assert_array_equal(r['top_idx'][s], gp_ref.topk_desc(r['vals'][s], k))
    C = np.cos(X @ W[0].T + b[0])
    A, v = e.rff_gram(W[0], b[0])
    np.testing.assert_allclose(A, C.T @ C, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(v, C.T @ (y - bias), rtol=1e-11, atol=1e-10)
    e.close()


@pytest.mark.parametrize('N', [641, 1153, 2500, 4099])
def test_odd_sizes_through_the_multi_panel_factorisation(N):
    """Sizes that are no multiple of the 128-block (identity padding inside the last block) and span several outer panels
    with all four streams active: residual of the factor, the explicit inverse, and the posterior against numpy."""
    from pybo_amd._lib import Engine
    X, y, ell = synth_problem(N, 5, seed=N)
    rho, sn2, bias = 1.2, 1e-3, 0.1
    ref = gp_ref.make_gp(sn2, rho, ell, bias, 'matern5')
    ref.add_data(X, y)
    K = ref.gram()
    e = Engine(0)
    e.fit(X, y, 'matern5', ell, rho, sn2, bias)
    L = e.get_matrix('L')
    assert np.linalg.norm(L @ L.T - K) / np.linalg.norm(K) <= 1e-13
    T = e.get_matrix('T')
    assert np.max(np.abs(T @ L - np.eye(N))) < 1e-9
    Z = np.random.RandomState(N).rand(777, 5)
    mu, s2 = e.predict(Z)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(mu - mr) <= mu_tol(mr, rho)) and np.all(np.abs(s2 - sr) <= s2_tol(sr, rho))
    e.close()


def test_task_graph_factor_is_the_stream_schedules_at_random_sizes():
    """The persistent kernel's roles start and stop inside the first and last block rows in a different way at every size: the
    factor must be the stream schedule's bit for bit at all of them (scripts/tg/tg_fuzz_sizes.py runs 90 sizes)."""
    from pybo_amd._lib import Engine
    rng = np.random.RandomState(11)
    a, b = Engine(0), Engine(0)
    b.set_option('chol_tg', 0)
    for N in [129, 256, 257, 385, 513, 1025] + list(rng.randint(130, 3300, size=12)):
        N = int(N)
        d = int(rng.randint(1, 7))
        X = rng.rand(N, d); y = np.sin(X.sum(1)) + 1e-3 * rng.randn(N)
        ell = 0.3 * np.ones(d)
        for e in (a, a, b):
            e.fit(X, y, 'matern5', ell, 1.1, 1e-4, 0.0, stage=2)
        assert np.array_equal(a.get_matrix('L'), b.get_matrix('L')), N
    assert a.timers(reset=True)['chol_fallbacks'] == 0
    a.close(); b.close()
