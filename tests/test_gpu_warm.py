"""Warm BO step (SURVEY N3's sweep-side half): with fixed hyper-parameters and a resident
candidate set, one more observation adds ONE row to V = T K*, so the per-candidate sums q = colsum(V^2), p = V^T a
of the last full sweep are corrected in O(N M) by gpx_append and re-scored in O(M) by gpx_sweep_update -- instead of
the refit + full solve the reference repeats every iteration (pybo/bayesopt.py:262-269 with a fixed `xgrid=` in
pybo/solvers/lbfgs.py:42-50).  Checked against a from-scratch fit + full sweep on the device AND against the CPU
oracle after 1, 16 and 200 appends (the factor grows across two 128-block boundaries on the way)."""
import numpy as np
import pytest

from oracle import gp_ref
from helpers import synth_problem, s2_tol, mu_tol

pytestmark = pytest.mark.gpu


def _fresh(X, y, kernel, ell, rho, sn2, bias):
    from pybo_amd._lib import Engine
    e = Engine(0)
    e.fit(X, y, kernel, ell, rho, sn2, bias)
    return e


@pytest.mark.parametrize('kernel,N0,d', [('se', 300, 3), ('matern5', 256, 2), ('matern3', 200, 20), ('matern1', 130, 5),
                                         ('se', 140, 40)])
def test_sweep_update_tracks_a_full_resweep(kernel, N0, d):
    from pybo_amd._lib import GpxError
    X, y, ell = synth_problem(N0 + 200, d, seed=17)
    rho, sn2, bias = 1.3, 1e-3, 0.2
    Z = np.random.RandomState(5).rand(3000, d)
    e = _fresh(X[:N0], y[:N0], kernel, ell, rho, sn2, bias)
    with pytest.raises(GpxError):
        e.sweep_update('ei', 0.5, k=5)                      # nothing cached yet
    e.set_option('sweep_cache', 1)
    full0 = e.sweep('ei', 0.5, Z, k=10, want_moments=True)
    e.set_option('sweep_cache', 0)
    assert e.sweep_cache_size() == len(Z)
    e.sweep('mean', None, X[:50], k=0)                      # an unrelated sweep leaves the cache alone
    same = e.sweep_update('ei', 0.5, k=10, want_moments=True)
    for key in ('acq', 'mu', 's2', 'top_val', 'top_idx'):
        np.testing.assert_array_equal(same[key], full0[key])        # re-scoring the untouched sums: bitwise
    n = N0
    for upto in (N0 + 1, N0 + 16, N0 + 200):
        while n < upto:
            assert e.append(X[n], y[n])
            n += 1
        assert e.N == n
        _, target = e.mean_at_obs()
        warm = e.sweep_update('ei', target, k=10, want_moments=True)
        cold_e = _fresh(X[:n], y[:n], kernel, ell, rho, sn2, bias)
        cold = cold_e.sweep('ei', target, Z, k=10, want_moments=True)
        cold_e.close()
        ref = gp_ref.make_gp(sn2, rho, ell, bias, kernel)
        ref.add_data(X[:n], y[:n])
        mr, sr = ref.predict(Z)
        for got in (warm, cold):
            assert np.all(np.abs(got['mu'] - mr) <= mu_tol(mr, rho))
            assert np.all(np.abs(got['s2'] - sr) <= s2_tol(sr, rho))
        # warm and cold agree far inside the stated tolerance (same maths, different summation order)
        assert np.all(np.abs(warm['s2'] - cold['s2']) <= 0.01 * s2_tol(sr, rho))
        assert np.all(np.abs(warm['mu'] - cold['mu']) <= 0.01 * mu_tol(mr, rho))
        eir = ref.get_improvement(ref.mean_at_obs().max(), Z)
        big = eir > 1e-9 * eir.max()
        np.testing.assert_allclose(warm['acq'][big], eir[big], rtol=1e-6)
        assert warm['top_idx'][0] == int(np.argmax(eir)) == cold['top_idx'][0]
        # the target / acquisition may change between re-scorings at no cost
        ucb = e.sweep_update('ucb', 2.0, k=3)
        assert ucb['top_idx'][0] == int(np.argmax(mr + np.sqrt(2.0 * sr)))
    e.fit(X[:N0], y[:N0], kernel, ell, rho, sn2, bias)      # a refit invalidates the cache
    assert e.sweep_cache_size() == 0
    with pytest.raises(GpxError):
        e.sweep_update('ei', 0.5, k=5)
    e.close()


def test_append_grows_the_factor_across_block_boundaries():
    """Round 1 refused an append at N = 128 j and refitted in O(N^3); now the factors move into buffers one block
    larger.  N = 256 exactly, 130 appends (two boundaries), compared with a from-scratch fit."""
    X, y, ell = synth_problem(256 + 130, 2, seed=3)
    rho, sn2, bias = 1.1, 1e-3, 0.0
    e = _fresh(X[:256], y[:256], 'se', ell, rho, sn2, bias)
    for i in range(256, 386):
        assert e.append(X[i], y[i])
    ref = gp_ref.make_gp(sn2, rho, ell, bias)
    ref.add_data(X, y)
    L = e.get_matrix('L')
    K = ref.gram()
    assert np.linalg.norm(L @ L.T - K) <= 1e-13 * np.linalg.norm(K)
    a, alpha = e.get_vectors()
    np.testing.assert_allclose(alpha, ref.alpha(), rtol=1e-7, atol=1e-9)
    Z = np.random.RandomState(0).rand(500, 2)
    mu, s2 = e.predict(Z)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(mu - mr) <= mu_tol(mr, rho)) and np.all(np.abs(s2 - sr) <= s2_tol(sr, rho))
    assert abs(e.loglik() - ref.loglikelihood()) <= 1e-9 * abs(ref.loglikelihood())
    e.close()


def test_bo_loop_over_a_device_grid_runs_warm_and_selects_the_same_points():
    """solve_bayesopt with a user model (fixed hyper-parameters) and a Sobol grid resident in HBM: after the first
    iteration every acquisition sweep is a re-scoring of the cached sums.  Same queried points as the CPU oracle
    driving the same loop over the same grid."""
    from pybo_amd import solve_bayesopt, models, inits
    bounds = np.array([[0.0, 1.0], [0.0, 1.0]])
    f = lambda x: float(-np.sum((np.asarray(x) - 0.35) ** 2) + 0.05 * np.sin(9 * x[0]))     # noqa: E731
    X0 = np.random.RandomState(1).rand(20, 2)
    y0 = np.array([f(x) for x in X0])
    grid = inits.init_sobol_device(bounds, 4096)
    host_grid = np.asarray(grid)
    gp = models.make_gp(1e-4, 0.2, [0.3, 0.3], float(y0.mean()))
    ref = gp_ref.make_gp(1e-4, 0.2, [0.3, 0.3], float(y0.mean()))
    gp.add_data(X0, y0); ref.add_data(X0, y0)
    xa, ma, ia = solve_bayesopt(f, bounds, model=gp, niter=12, policy='ei', recommender='incumbent',
                                solver=('lbfgs', {'xgrid': grid, 'nbest': 3}), rng=0)
    xb, mb, ib = solve_bayesopt(f, bounds, model=ref, niter=12, policy='ei', recommender='incumbent',
                                solver=('lbfgs', {'xgrid': host_grid, 'nbest': 3}), rng=0)
    np.testing.assert_allclose(ia.x, ib.x, atol=2e-5)
    np.testing.assert_allclose(ia.y, ib.y, atol=1e-6)
    assert ma._state.cache_grid is grid and ma._state.engine.sweep_cache_size() == 4096
    tm = ma._state.engine.timers()
    # 12 iterations: ONE full sweep of the grid (the first), then rank-1 corrections
    assert tm['rank1'] > 0 and tm['append'] > 0


# ---- announced observations: the value-independent part of add_data runs during the objective (gpx_append_begin) ---
def test_announced_append_is_bit_identical_and_survives_every_way_of_not_using_it():
    from pybo_amd._lib import Engine
    from helpers import synth_problem
    X, y, ell = synth_problem(300, 3, seed=12)          # Np = 384: 84 rows of padding left
    rho, sn2, bias = 1.3, 1e-3, 0.2
    Z = np.random.RandomState(1).rand(20000, 3)
    rng = np.random.RandomState(5)
    new = rng.rand(6, 3)
    ynew = np.sin(3 * new.sum(1))

    def engine():
        e = Engine(0)
        e.fit(X, y, 'matern5', ell, rho, sn2, bias)
        e.set_option('sweep_cache', 1)
        e.sweep('ei', 0.4, Z, k=5, want_all=False)
        e.set_option('sweep_cache', 0)
        return e

    plain, ahead = engine(), engine()
    for i in range(6):
        assert plain.append(new[i], ynew[i])
        if i == 2:
            assert ahead.append_begin(new[5])           # announce ONE point, append ANOTHER: the announcement is ignored
        elif i == 4:
            assert ahead.append_begin(new[i])
            mu, s2 = ahead.predict(Z[:50])              # other work between the announcement and the value is fine
        else:
            assert ahead.append_begin(new[i])
        assert ahead.append(new[i], ynew[i])
        a = plain.sweep_update('ei', 0.4, k=5, want_moments=True)
        b = ahead.sweep_update('ei', 0.4, k=5, want_moments=True)
        for key in ('acq', 'mu', 's2', 'top_val', 'top_idx'):
            np.testing.assert_array_equal(a[key], b[key])      # the same FMAs in the same order: the same bits
    np.testing.assert_array_equal(plain.get_matrix('L'), ahead.get_matrix('L'))
    np.testing.assert_array_equal(plain.get_vectors()[1], ahead.get_vectors()[1])
    # an announcement followed by a refit, and one without a live cache
    assert ahead.append_begin(new[0])
    ahead.fit(X, y, 'matern5', ell, rho, sn2, bias)
    assert not ahead.append_begin(new[0])               # the refit dropped the cache: nothing to run ahead
    r = ahead.sweep('ei', 0.4, Z[:3000], k=3)
    ref = engine().sweep('ei', 0.4, Z[:3000], k=3)
    np.testing.assert_array_equal(r['acq'], ref['acq'])
    # at a block boundary (N = 256 = Np: the benchmark's N = 8192 is such a size) the announcement adds the block itself,
    # as the append would have; same bits as the unannounced run
    Xb, yb, ellb = synth_problem(256, 2, seed=3)
    pair = []
    for announce in (False, True):
        e = Engine(0)
        e.fit(Xb, yb, 'se', ellb, 1.0, 1e-3, 0.0)
        e.set_option('sweep_cache', 1)
        e.sweep('ucb', 2.0, Z[:1000, :2], k=1, want_all=False)
        e.set_option('sweep_cache', 0)
        rs = []
        for xn, yn in ((np.array([0.3, 0.4]), 0.1), (np.array([0.5, 0.1]), -0.2)):
            if announce:
                assert e.append_begin(xn)
            assert e.append(xn, yn)
            rs.append(e.sweep_update('ucb', 2.0, k=3, want_moments=True))      # (one correction per re-score, as in the loop)
        pair.append((rs, e.get_matrix('L'), e.get_matrix('T')))
    for ra, rb in zip(pair[0][0], pair[1][0]):
        for key in ('acq', 'mu', 's2', 'top_val', 'top_idx'):
            np.testing.assert_array_equal(ra[key], rb[key])
    np.testing.assert_array_equal(pair[0][1], pair[1][1])
    np.testing.assert_array_equal(pair[0][2], pair[1][2])


def test_the_loop_announces_its_query_point_and_results_do_not_change():
    """pybo_amd.bayesopt._bo_step calls model.anticipate(x) between the solver and the objective; the run is the same
    bit for bit as without it, and add_data finds the announcement (the correction pass is not repeated)."""
    import pybo_amd
    from pybo_amd import models, inits
    from helpers import branin
    bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
    rng = np.random.RandomState(0)
    X = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.rand(400, 2)
    y = -branin(X) / 10.0
    traces = []
    for announce in (True, False):
        gp = models.make_gp(1e-4 * np.var(y), np.var(y), 0.25 * (bounds[:, 1] - bounds[:, 0]), np.mean(y))
        gp.add_data(X, y)
        seen = []
        real = models.GP.anticipate

        def spy(self, x, seen=seen):
            seen.append(real(self, x))
            return seen[-1]
        grid = inits.init_sobol_device(bounds, 30000, rng=3)
        models.GP.anticipate = spy if announce else (lambda self, x: False)
        try:
            xb, m, info = pybo_amd.solve_bayesopt(lambda x: float(-branin(x)[0] / 10.0), bounds, model=gp, niter=5,
                                                  solver=('lbfgs', {'xgrid': grid}), recommender='latent', rng=1)
        finally:
            models.GP.anticipate = real
        if announce:
            assert seen[1:] == [True] * 4               # from the second iteration on the cache is live
        traces.append((info.x, info.xbest, m._engine().get_matrix('L')))
    for a, b in zip(*traces):
        np.testing.assert_array_equal(a, b)


def test_long_warm_run_across_block_boundaries_ends_where_a_fresh_fit_does():
    """300 iterations of solve_bayesopt over a resident grid of 2^17 Sobol points, started at N = 900: the factor grows
    across three 128-block boundaries, every acquisition sweep after the first is a correction of the cached sums
    (announced query points, noisy objective, checkpoint on).  At the end the warm model must agree with the oracle
    fitted from scratch on the final data, and its cached re-score with a cold sweep of a fresh handle."""
    from pybo_amd import solve_bayesopt, models, inits
    from pybo_amd._lib import Engine
    from pybo_amd.models import gp as gpmod
    import bench
    for pooled in gpmod._ENGINE_POOL:              # fresh handles: the launch counter below is a handle's lifetime total
        pooled.close()
    del gpmod._ENGINE_POOL[:]
    d = 6
    bounds = np.stack([np.zeros(d), np.ones(d)], axis=1)
    noise = np.random.RandomState(3)
    calls = []

    def f(x):
        calls.append(np.array(x, dtype=float))
        return float(bench.hartmann6(np.array(x, ndmin=2))[0] + 1e-3 * noise.randn())

    X0 = np.random.RandomState(2).rand(900, d)
    y0 = np.array([float(bench.hartmann6(x[None])[0]) for x in X0])
    sn2, rho, ell, bias = 1e-5, float(np.var(y0)), np.full(d, 0.35), float(y0.mean())
    gp = models.make_gp(sn2, rho, ell, bias, kernel='matern5')
    gp.add_data(X0, y0)
    grid = inits.init_sobol_device(bounds, 1 << 17)
    _, model, info = solve_bayesopt(f, bounds, model=gp, niter=300, policy='ei', recommender='incumbent',
                                    solver=('lbfgs', {'xgrid': grid, 'nbest': 2}), rng=0)
    eng = model._state.engine
    # (solve_bayesopt feeds a user model the box centre first: 900 + 1 + 300 observations)
    assert eng.N == 1201 and len(calls) == 301
    tm = eng.timers()
    assert tm['sweep_trmm_launches'] <= 4          # ONE full sweep (2 chunks of 65536), everything after it warm
    Xall, yall = model.data
    ref = gp_ref.make_gp(sn2, rho, ell, bias, 'matern5')
    ref.add_data(Xall, yall)
    Z = np.random.RandomState(9).rand(4000, d)
    mu, s2 = model.predict(Z)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(mu - mr) <= mu_tol(mr, rho)) and np.all(np.abs(s2 - sr) <= s2_tol(sr, rho))
    target = float(eng.mean_at_obs()[1])
    warm = eng.sweep_update('ei', target, k=8, want_moments=True)
    cold_e = Engine(0)
    cold_e.fit(Xall, yall, 'matern5', ell, rho, sn2, bias)
    cold = cold_e.sweep('ei', target, np.asarray(grid), k=8, want_moments=True)
    cold_e.close()
    mg, sg = cold['mu'], cold['s2']
    assert np.all(np.abs(warm['mu'] - mg) <= mu_tol(mg, rho)) and np.all(np.abs(warm['s2'] - sg) <= s2_tol(sg, rho))
    assert warm['top_idx'][0] == cold['top_idx'][0]
