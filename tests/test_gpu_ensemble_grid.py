"""the hyper-parameter ENSEMBLE sweep (pybo's default model is an MCMC ensemble of GPs,
reference bayesopt.py:115) and candidate grids generated in HBM, through the C-ABI, against the oracle."""
import numpy as np
import pytest

from oracle import gp_ref
from helpers import synth_problem

pytestmark = pytest.mark.gpu

HYPERS = [(1e-3, 1.0, 0.30, 0.0), (5e-3, 1.6, 0.45, 0.2), (2e-4, 0.7, 0.22, -0.1), (1e-2, 1.2, 0.60, 0.05)]


def _members(N, d, kernel, seed):
    from pybo_amd import models
    X, y, _ = synth_problem(N, d, seed=seed)
    dev, ref = [], []
    for sn2, rho, ell, bias in HYPERS:
        g = models.make_gp(sn2, rho, [ell] * d, bias, kernel=kernel)
        r = gp_ref.make_gp(sn2, rho, [ell] * d, bias, kernel)
        g.add_data(X, y); r.add_data(X, y)
        dev.append(g); ref.append(r)
    return dev, ref


def _ref_ensemble(ref, kind, param, Xc):
    mus = np.array([r.predict(Xc)[0] for r in ref])
    s2s = np.array([r.predict(Xc)[1] for r in ref])
    mu = mus.mean(0)
    s2 = np.maximum((s2s + mus ** 2).mean(0) - mu ** 2, 0.0)
    if kind == 'ei':
        return np.mean([r.get_improvement(param, Xc) for r in ref], axis=0), mu, s2
    if kind == 'pi':
        return np.mean([r.get_tail(param, Xc) for r in ref], axis=0), mu, s2
    if kind == 'ucb':
        return mu + np.sqrt(param * s2), mu, s2
    return mu, mu, s2


@pytest.mark.parametrize('kind,param', [('ei', 0.4), ('pi', 0.3), ('ucb', 3.0), ('mean', None)])
@pytest.mark.parametrize('N,d,kernel', [(150, 2, 'se'), (700, 3, 'matern5')])
def test_ensemble_sweep_matches_oracle_members(kind, param, N, d, kernel):
    from pybo_amd._lib import Engine
    dev, ref = _members(N, d, kernel, seed=N + d)
    Xc = np.random.RandomState(5).rand(3000, d)
    engines = [g._engine() for g in dev]
    moments = kind in ('ucb', 'mean')
    out = Engine.ensemble_sweep(engines, kind, param, Xc, k=8, want_all=True, want_moments=moments)
    want, mu, s2 = _ref_ensemble(ref, kind, param, Xc)
    scale = np.max(np.abs(want))
    np.testing.assert_allclose(out['acq'], want, rtol=1e-6, atol=1e-9 * scale)
    if moments:
        np.testing.assert_allclose(out['mu'], mu, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(out['s2'], s2, rtol=1e-6, atol=1e-10)
    # the top-k is the exact ranking of the device's own averaged values (value desc, index asc)
    idx = gp_ref.topk_desc(out['acq'], 8)
    np.testing.assert_array_equal(out['top_idx'], idx)
    np.testing.assert_array_equal(out['top_val'], out['acq'][idx])
    # and the average is the member-order sum of the members' own sweeps divided once by n
    if kind in ('ei', 'pi'):
        own = np.mean([e.sweep(kind, param, Xc, k=0)['acq'] for e in engines], axis=0)
        np.testing.assert_allclose(out['acq'], own, rtol=1e-14, atol=0)


def test_ensemble_errors():
    from pybo_amd._lib import Engine, GpxError
    dev, _ = _members(40, 2, 'se', seed=3)
    other = Engine(0)                                   # never fitted
    with pytest.raises(GpxError):
        Engine.ensemble_sweep([dev[0]._engine(), other], 'ei', 0.1, np.zeros((4, 2)), k=1)
    with pytest.raises(GpxError):                       # moments only exist for ucb / mean
        Engine.ensemble_sweep([dev[0]._engine()], 'ei', 0.1, np.zeros((4, 2)), k=0, want_moments=True)
    with pytest.raises(GpxError):
        Engine.ensemble_sweep([dev[0]._engine()], 'ei', 0.1, np.zeros((4, 2)), k=4097)


def test_mcmc_model_uses_the_device_ensemble():
    """MCMC over device GPs: predict / get_improvement / acq_topk run as ONE ensemble call and agree with
    the member-by-member host combination."""
    from pybo_amd import models
    X, y, _ = synth_problem(40, 2, seed=11)
    gp = models.make_gp(1e-3, 1.0, [0.4, 0.4], 0.0)
    gp.params['like.sn2'].set_prior('horseshoe', 0.1)
    gp.params['kern.rho'].set_prior('lognormal', 0.0, 1.0)
    gp.params['kern.ell'].set_prior('uniform', [0.02] * 2, [3.0] * 2)
    gp.params['mean.bias'].set_prior('normal', 0.0, 4.0)
    gp.add_data(X, y)
    ens = models.MCMC(gp, n=5, burn=10, rng=0)
    Xc = np.random.RandomState(1).rand(500, 2)
    mu, s2 = ens.predict(Xc)
    posts = [m.predict(Xc) for m in ens._members]
    mus = np.array([p[0] for p in posts]); s2s = np.array([p[1] for p in posts])
    np.testing.assert_allclose(mu, mus.mean(0), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(s2, (s2s + mus ** 2).mean(0) - mus.mean(0) ** 2, rtol=1e-9, atol=1e-14)
    ei = ens.get_improvement(0.2, Xc)
    np.testing.assert_allclose(ei, np.mean([m.get_improvement(0.2, Xc) for m in ens._members], axis=0),
                               rtol=1e-13, atol=0)
    tv, ti = ens.acq_topk('ei', 0.2, Xc, 5)
    np.testing.assert_array_equal(ti, gp_ref.topk_desc(ei, 5))
    np.testing.assert_array_equal(tv, ei[ti])
    tv, ti = ens.acq_topk('ucb', 2.0, Xc, 5)
    ucb = mu + np.sqrt(2.0 * s2)
    np.testing.assert_array_equal(ti, gp_ref.topk_desc(ucb, 5))


# ---- grids in HBM ------------------------------------------------------------------------------------
@pytest.mark.parametrize('d,n', [(1, 7), (2, 1000), (8, 4097), (21, 333)])
def test_device_sobol_equals_host_sobol_bit_for_bit(d, n):
    from pybo_amd import inits
    bounds = np.stack([-1.0 - np.arange(d), 2.0 + 0.5 * np.arange(d)], axis=1)
    host = inits.init_sobol(bounds, n, rng=4)
    grid = inits.init_sobol_device(bounds, n, rng=4)
    assert grid.shape == (n, d) and len(grid) == n
    np.testing.assert_array_equal(np.asarray(grid), host)
    pick = np.array([n - 1, 0, n // 2])
    np.testing.assert_array_equal(grid[pick], host[pick])
    np.testing.assert_array_equal(grid[n - 1], host[n - 1])


@pytest.mark.parametrize('d,n', [(1, 5), (3, 1001), (8, 5000)])
def test_device_uniform_equals_philox_restatement(d, n):
    from pybo_amd._lib import DeviceGrid
    bounds = np.stack([-2.0 + np.arange(d), 1.0 + 2.0 * np.arange(d)], axis=1)
    seed = 0x1234567890ABCDE
    grid = DeviceGrid('uniform', bounds, n, seed=seed)
    got = np.asarray(grid)
    np.testing.assert_array_equal(got, gp_ref.grid_uniform(seed, bounds, n))
    assert np.all(got >= bounds[:, 0]) and np.all(got <= bounds[:, 1])
    # crude uniformity: every coordinate's mean within 5 sigma of the centre for the larger grids
    if n >= 1000:
        u = (got - bounds[:, 0]) / (bounds[:, 1] - bounds[:, 0])
        assert np.all(np.abs(u.mean(0) - 0.5) < 5.0 / np.sqrt(12.0 * n))


def test_grid_errors():
    from pybo_amd._lib import DeviceGrid, GpxError
    with pytest.raises(ValueError):
        DeviceGrid('halton', [[0, 1]], 4)
    g = DeviceGrid('uniform', [[0, 1], [0, 1]], 16, seed=1)
    with pytest.raises(GpxError):
        g.rows([16])
    with pytest.raises(GpxError):
        DeviceGrid('uniform', np.zeros((1025, 2)), 4)       # d > 1024


def test_sweeps_over_a_device_grid_equal_sweeps_over_the_host_copy():
    from pybo_amd import models, inits, policies
    X, y, ell = synth_problem(300, 3, seed=2)
    gp = models.make_gp(1e-3, 1.2, ell, 0.1)
    gp.add_data(X, y)
    bounds = np.array([[0.0, 1.0]] * 3)
    grid = inits.init_sobol_device(bounds, 20000, rng=0)
    host = np.asarray(grid)
    for kind, param in (('ei', 0.3), ('ucb', 2.5)):
        tv_d, ti_d = gp.acq_topk(kind, param, grid, 10)
        tv_h, ti_h = gp.acq_topk(kind, param, host, 10)
        np.testing.assert_array_equal(ti_d, ti_h)
        np.testing.assert_array_equal(tv_d, tv_h)
    sample = gp.sample_f(40, rng=3)
    tv_d, ti_d = sample.topk(grid, 6)
    tv_h, ti_h = sample.topk(host, 6)
    np.testing.assert_array_equal(ti_d, ti_h)
    np.testing.assert_array_equal(tv_d, tv_h)
    index = policies.Thompson(gp, bounds, None, n=30, rng=1)
    assert hasattr(index, 'topk')


def test_solver_with_a_device_grid_gives_the_host_grid_answer():
    from pybo_amd import models, inits, policies, solvers
    X, y, ell = synth_problem(120, 2, seed=6)
    gp = models.make_gp(1e-3, 1.0, ell, 0.0)
    gp.add_data(X, y)
    bounds = np.array([[0.0, 1.0]] * 2)
    grid = inits.init_sobol_device(bounds, 4096, rng=2)
    index = policies.EI(gp, bounds, X)
    xd, fd = solvers.solve_lbfgs(index, bounds, xgrid=grid)
    xh, fh = solvers.solve_lbfgs(index, bounds, xgrid=np.asarray(grid))
    np.testing.assert_array_equal(xd, xh)
    assert fd == fh
    # a host-side index (no .topk) still works: the grid is materialised
    f = lambda Z, grad=False: ((-np.sum((Z - 0.3) ** 2, axis=1), -2 * (Z - 0.3)) if grad
                               else -np.sum((Z - 0.3) ** 2, axis=1))
    xs, _ = solvers.solve_lbfgs(f, bounds, xgrid=grid)
    np.testing.assert_allclose(xs, [0.3, 0.3], atol=1e-6)


# ---- batched multi-start refinement ------------------------------------------------------
def test_lockstep_refinement_on_the_device_equals_sequential():
    """Rows of a batched predict-with-gradient call (two or more rows) are independent of the batch they travel in, so the
    lock-step refinement of all seeds reproduces the one-after-the-other refinement exactly (a lone seed travels twice:
    solvers.lbfgs._batch_form).  A SINGLE-row call takes the one-pass form (ds2 = -2 (T k).(T dk)): same value to
    rounding, not the same bits."""
    from pybo_amd import models, policies, solvers
    X, y, ell = synth_problem(200, 3, seed=8)
    gp = models.make_gp(1e-3, 1.0, ell, 0.0, kernel='matern5')
    gp.add_data(X, y)
    bounds = np.array([[0.0, 1.0]] * 3)
    index = policies.EI(gp, bounds, X)
    P = np.random.RandomState(0).rand(11, 3)
    fb, gb = index(P, grad=True)
    for i in range(len(P)):
        fi, gi = index(P[[i, (i + 3) % len(P)]], grad=True)
        assert fi[0] == fb[i]
        np.testing.assert_array_equal(gi[0], gb[i])
        f1, g1 = index(P[i:i + 1], grad=True)
        np.testing.assert_allclose(f1[0], fb[i], rtol=1e-11, atol=1e-300)
        np.testing.assert_allclose(g1[0], gb[i], rtol=1e-9, atol=1e-12 * np.abs(gb).max())
    grid = np.random.RandomState(1).rand(3000, 3)
    xa, fa = solvers.solve_lbfgs(index, bounds, nbest=5, xgrid=grid, select='best', batched=False)
    xb, fb_ = solvers.solve_lbfgs(index, bounds, nbest=5, xgrid=grid, select='best', batched=True)
    np.testing.assert_array_equal(xa, xb)
    assert fa == fb_


@pytest.mark.parametrize('N,d,kernel,M', [(150, 2, 'se', 5), (300, 3, 'matern5', 37)])
def test_ensemble_predict_equals_the_members_own_gradients(N, d, kernel, M):
    """gpx_ensemble_predict: every member's moments and gradients in one call (the members' kernels overlap on their
    own streams) are BITWISE what gpx_predict gives member by member, and match the oracle; the ensemble's
    get_improvement / get_tail / predict with grad=True built on it equal the member loop."""
    from pybo_amd._lib import Engine, GpxError
    from pybo_amd.models.mcmc import MCMC
    dev, ref = _members(N, d, kernel, seed=7 * N + d)
    Z = np.random.RandomState(9).rand(M, d)
    engines = [g._engine() for g in dev]
    mu, s2, dmu, ds2 = Engine.ensemble_predict(engines, Z)
    assert mu.shape == (len(dev), M) and ds2.shape == (len(dev), M, d)
    for i, (g, r) in enumerate(zip(dev, ref)):
        own = g._engine().predict(Z, grad=True)
        for got, o in zip((mu[i], s2[i], dmu[i], ds2[i]), own):
            np.testing.assert_array_equal(got, o)
        want = r.predict(Z, grad=True)
        np.testing.assert_allclose(dmu[i], want[2], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(ds2[i], want[3], rtol=1e-6, atol=1e-8)
    ens = MCMC.__new__(MCMC)                       # an ensemble over exactly these members (no sampling)
    ens._proto, ens._members = dev[0], dev
    for name, target in (('get_improvement', 0.4), ('get_tail', 0.3)):
        f, gr = getattr(ens, name)(target, Z, grad=True)
        loop = [getattr(m, name)(target, Z, True) for m in dev]
        np.testing.assert_allclose(f, np.mean([o[0] for o in loop], axis=0), rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(gr, np.mean([o[1] for o in loop], axis=0), rtol=1e-12, atol=1e-15)
    got = ens.predict(Z, grad=True)
    posts = [m.predict(Z, True) for m in dev]
    mus, dmus = np.array([p[0] for p in posts]), np.array([p[2] for p in posts])
    np.testing.assert_allclose(got[0], mus.mean(0), rtol=1e-14)
    np.testing.assert_allclose(got[2], dmus.mean(0), rtol=1e-13, atol=1e-16)
    with pytest.raises(GpxError):
        Engine.ensemble_predict(engines, np.zeros((0, d)))


# ---- the reference's DEFAULT model at BASELINE scale -----------------------------------------------
def test_ten_member_ensemble_sweep_at_the_north_star_size():
    """pybo's default model is MCMC(gp, n=10) (pybo/bayesopt.py:115): ten member GPs at N = 8192, d = 8, EI averaged over the
    members on the full 2^20 Sobol grid, ONE gpx_ensemble_sweep call.  The oracle (ten CPU fits) on every 512th candidate
    plus the device's top-k: values at the EI tolerance, the winner, the order of the top-k."""
    import bench
    from pybo_amd._lib import Engine
    from helpers import ei_tol
    M, n, k = 1 << 20, 10, 32
    w = bench.make_workload('ns', M)
    X, y, Z = w['X'], w['y'], w['Xc']
    hyp = bench.ensemble_hypers(w, n)              # the members bench.py --ensemble times
    engines = []
    for sn2, rho, ell, bias in hyp:
        e = Engine(0)
        e.fit(X, y, w['kernel'], ell, rho, sn2, bias)
        engines.append(e)
    target = float(np.max(y))                      # one incumbent for all members (the policy's target is a model-level scalar)
    out = Engine.ensemble_sweep(engines, 'ei', target, Z, k=k, want_all=True)
    ei = out['acq']
    assert np.all(np.isfinite(ei)) and np.all(ei >= -1e-300)
    np.testing.assert_array_equal(out['top_idx'], gp_ref.topk_desc(ei, k))          # top-k = ranking of the returned average
    np.testing.assert_array_equal(out['top_val'], ei[out['top_idx']])
    # the average is the member-order sum of the members' own sweeps, divided once by n: bitwise on a slice of the grid
    sl = slice(0, 1 << 16)
    own = np.zeros(1 << 16)
    for e in engines:
        own += e.sweep('ei', target, Z[sl], k=0)['acq']
    np.testing.assert_array_equal(ei[sl], own / n)
    # oracle members on a sub-sample + the device's top-k
    pick = np.unique(np.concatenate([np.arange(0, M, 512), out['top_idx']]))
    assert len(pick) >= 2048
    want = np.zeros(len(pick))
    tol = np.zeros(len(pick))
    for sn2, rho, ell, bias in hyp:
        ref = gp_ref.make_gp(sn2, rho, ell, bias, w['kernel'])
        ref.add_data(X, y)
        mr, sr = ref.predict(Z[pick])
        want += ref.get_improvement(target, Z[pick])
        tol += ei_tol(mr, sr, target, rho)
    want /= n
    tol /= n
    big = want > 1e-9 * want.max()
    assert big.sum() > 500
    np.testing.assert_allclose(ei[pick][big], want[big], rtol=1e-6, atol=0)
    assert np.all(np.abs(ei[pick] - want) <= tol)
    assert out['top_idx'][0] == pick[int(np.argmax(want))]
    pos = np.searchsorted(pick, out['top_idx'])
    vals = want[pos]
    gaps = vals[:-1] - vals[1:]
    assert np.all(gaps >= -(tol[pos][:-1] + tol[pos][1:]))
    for e in engines:
        e.close()
