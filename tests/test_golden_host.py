"""Host-side plugin logic against golden vectors captured from the reference's own modules
(tests/golden/make_golden.py imports /root/reference/pybo/{policies/simple,solvers/lbfgs,inits/methods,
recommenders}.py in the build container; only the vectors travel)."""
import os

import numpy as np
import pytest

from pybo_amd import bayesopt, inits, policies, recommenders, solvers
from helpers import StubModel, analytic_index

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_inits_match_reference_bit_for_bit():
    g = np.load(os.path.join(G, 'inits.npz'))
    b2 = [[0.0, 1.0], [2.0, 4.0]]
    b3 = [[-5.0, 10.0], [0.0, 15.0], [1.0, 3.0]]
    assert np.array_equal(inits.init_middle(b2), g['middle_b2'])
    assert np.array_equal(inits.init_middle(b3), g['middle_b3'])
    for name, bounds in (('b2', b2), ('b3', b3)):
        for seed in (0, 7):
            for n in (None, 5, 64):
                key = '%s_s%d_n%s' % (name, seed, n)
                assert np.array_equal(inits.init_uniform(bounds, n, seed), g['uniform_' + key]), key
                assert np.array_equal(inits.init_latin(bounds, n, seed), g['latin_' + key]), key


def test_init_sobol_is_a_valid_low_discrepancy_design():
    # by design NOT the reference's LGPL generator (candidates are an input of the hot path)
    b3 = np.array([[-5.0, 10.0], [0.0, 15.0], [1.0, 3.0]])
    X = inits.init_sobol(b3, 64, 0)
    assert X.shape == (64, 3)
    assert np.all(X >= b3[:, 0]) and np.all(X <= b3[:, 1])
    u = (X - b3[:, 0]) / (b3[:, 1] - b3[:, 0])
    # 64 consecutive Sobol points put exactly 16 in each quarter of every axis... up to the skip
    # offset; check balance loosely
    for k in range(3):
        h, _ = np.histogram(u[:, k], bins=4, range=(0, 1))
        assert h.min() >= 12 and h.max() <= 20
    assert inits.init_sobol(b3, None, 1).shape == (9, 3)


@pytest.mark.parametrize('pname,kw', [('EI', {}), ('EI', {'xi': 0.25}), ('PI', {}), ('PI', {'xi': 0.3}),
                                      ('UCB', {}), ('UCB', {'delta': 0.05, 'xi': 0.7})])
def test_policies_match_reference(pname, kw):
    g = np.load(os.path.join(G, 'policies.npz'))
    Xobs = [x for x in g['Xobs']]
    Xq = g['Xq']
    stub = StubModel()
    index = getattr(policies, pname)(stub, None, Xobs, **kw)
    tag = pname + ''.join('_%s%g' % kv for kv in sorted(kw.items()))
    v = index(Xq)
    v2, gr = index(Xq, grad=True)
    np.testing.assert_allclose(v, g[tag + '_val'], rtol=1e-15, atol=0)
    np.testing.assert_allclose(v2, g[tag + '_val_g'], rtol=1e-15, atol=0)
    np.testing.assert_allclose(gr, g[tag + '_grad'], rtol=1e-15, atol=0)
    # same protocol calls in the same order (copy -> predict(X_obs) -> index calls)
    assert '|'.join(stub.log) == str(g[tag + '_log'])


def test_ucb_beta_uses_number_of_observations():
    g = np.load(os.path.join(G, 'policies.npz'))
    beta = 0.2 * 2 * np.log(np.pi ** 2 / 0.3) + 0.2 * 7 * np.log(4.0)      # SURVEY F7, N = 3
    assert abs(beta - 3.3381851359777412) < 1e-14
    assert abs(float(g['UCB_beta_N3']) - beta) < 1e-12
    stub = StubModel()
    Xq = g['Xq'][:, :2]
    idx = policies.UCB(stub, None, [np.zeros(2)] * 3)
    mu, s2 = StubModel.moments(Xq)[:2]
    np.testing.assert_allclose((idx(Xq) - mu) ** 2 / s2, beta, rtol=1e-12)


@pytest.mark.parametrize('kind', ['bimodal2', 'tilted1', 'quad5'])
def test_solver_selection_matches_reference(kind):
    g = np.load(os.path.join(G, 'solver.npz'))
    f, bounds = analytic_index(kind)
    d = len(bounds)
    for nbest in (1, 3, 10):
        grid = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * np.random.RandomState(11).rand(200, d)
        x, fx = solvers.solve_lbfgs(f, bounds, nbest=nbest, xgrid=grid)
        np.testing.assert_allclose(x, g['%s_nb%d_x' % (kind, nbest)], rtol=0, atol=1e-12)
        np.testing.assert_allclose(fx, g['%s_nb%d_f' % (kind, nbest)], rtol=1e-12)
    x, fx = solvers.solve_lbfgs(f, bounds, nbest=4, ngrid=500, rng=5)
    np.testing.assert_allclose(x, g['%s_rng5_x' % kind], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fx, g['%s_rng5_f' % kind], rtol=1e-12)


def test_solver_reference_selection_quirk_and_fix():
    # F6: the reference returns the refinement of the single best grid point; with a grid whose best
    # point sits in the basin of the LOWER peak, 'first' and 'best' differ.
    f, bounds = analytic_index('bimodal2')
    grid = np.array([[0.25, 0.3], [0.6, 0.6], [0.1, 0.9]])       # best grid value is at the minor peak
    x1, f1 = solvers.solve_lbfgs(f, bounds, nbest=3, xgrid=grid)
    x2, f2 = solvers.solve_lbfgs(f, bounds, nbest=3, xgrid=grid, select='best')
    assert f2 >= f1
    np.testing.assert_allclose(x2, [0.8, 0.8], atol=2e-2)
    np.testing.assert_allclose(x1, [0.25, 0.3], atol=5e-2)


def test_solver_uses_device_topk_hook_when_present():
    f, bounds = analytic_index('bimodal2')
    calls = []

    def g(X, grad=False):
        return f(X, grad)

    def topk(xgrid, k):
        calls.append((len(xgrid), k))
        v = f(xgrid)
        order = np.lexsort((np.arange(len(v)), -v))[:k]
        return v[order], order
    g.topk = topk
    grid = np.random.RandomState(0).rand(300, 2)
    xa, fa = solvers.solve_lbfgs(g, bounds, nbest=5, xgrid=grid)
    xb, fb = solvers.solve_lbfgs(f, bounds, nbest=5, xgrid=grid)
    assert calls == [(300, 5)]
    np.testing.assert_allclose(xa, xb)
    np.testing.assert_allclose(fa, fb)


def test_recommenders_match_reference():
    g = np.load(os.path.join(G, 'recommenders.npz'))
    Xobs = g['Xobs']
    bounds = np.array([[0.0, 1.0]] * 3)
    np.testing.assert_array_equal(recommenders.best_incumbent(StubModel(), bounds, Xobs), g['incumbent'])
    np.testing.assert_allclose(recommenders.best_latent(StubModel(), bounds, Xobs), g['latent'], atol=1e-12)


def test_get_component_resolution_table():
    rng = np.random.RandomState(0)
    gc = bayesopt.get_component
    assert gc('ei', policies, rng) is policies.EI
    assert gc('ucb', policies, rng) is policies.UCB
    th = gc('thompson', policies, rng)                 # has an rng argument -> partial with rng injected
    assert th.func is policies.Thompson and th.keywords == {'rng': rng}
    s = gc(('lbfgs', {'nbest': 3}), solvers, rng, lstrip='solve_')
    assert s.func is solvers.solve_lbfgs and s.keywords == {'nbest': 3, 'rng': rng}
    assert gc('latent', recommenders, rng, lstrip='best_') is recommenders.best_latent
    assert gc('incumbent', recommenders, rng, lstrip='best_') is recommenders.best_incumbent
    f = lambda model, bounds, X: None                   # noqa: E731  any callable passes through
    assert gc(f, policies, rng) is f
    with pytest.raises(ValueError):
        gc('nope', policies, rng)
    with pytest.raises(ValueError):
        gc(('ei', {'bogus': 1}), policies, rng)
    with pytest.raises(ValueError):
        gc(('thompson', {'rng': 1}), policies, rng)     # rng is never a user kwarg
    with pytest.raises(ValueError):
        gc(('ei', 1, 2), policies, rng)
    with pytest.raises(ValueError):
        gc('EI', policies, rng)                         # names are lower-case, as in the reference


# ---- round 2: fixtures captured from the reference's OWN bayesopt.py (loaded against a recording `reggie`) ----
def _spec(text):
    import ast
    return ast.literal_eval(text)


def test_get_component_matches_the_reference_function():
    """Every row = one call of the reference's get_component (pybo/bayesopt.py:125-176): which function it
    resolved to, which kwargs were bound, whether the shared rng was injected -- or that it raised."""
    rng = np.random.RandomState(0)
    mods = {'policies': policies, 'solvers': solvers, 'recommenders': recommenders}
    rows = np.load(os.path.join(G, 'components.npz'))['table'].tolist()
    assert len(rows) == 16
    for row in rows:
        spec, modname, strip, outcome = row.split('|')[:4]
        spec = _spec(spec)
        if outcome != 'ok':
            # (the reference's malformed-tuple branch dies inside its own error message -- '{:r}' is not a
            # format code -- as ValueError under Python 2 and TypeError under the Python 3 re-load; here: ValueError)
            assert outcome in ('ValueError', 'TypeError')
            with pytest.raises(ValueError):
                bayesopt.get_component(spec, mods[modname], rng, lstrip=strip)
            continue
        name, kwargs, has_rng = row.split('|')[4:]
        got = bayesopt.get_component(spec, mods[modname], rng, lstrip=strip)
        func = getattr(got, 'func', got)
        kw = dict(getattr(got, 'keywords', {}) or {})
        assert (kw.pop('rng', None) is rng) == bool(int(has_rng)), row
        assert func.__name__ == name and sorted(kw.items()) == _spec(kwargs), row


@pytest.mark.parametrize('tag,bounds,kw', [
    ('ei_latent_2d', [[0.0, 1.0], [-0.5, 1.0]], dict(policy='ei', recommender='latent')),
    ('pi_incumbent_2d', [[0.0, 1.0], [-0.5, 1.0]], dict(policy=('pi', {'xi': 0.02}), recommender='incumbent')),
    ('ucb_latent_3d', [[0.0, 1.0]] * 3, dict(policy='ucb', recommender='latent',
                                               solver=('lbfgs', {'nbest': 4, 'ngrid': 400})))])
def test_whole_bo_loop_matches_the_reference_loop(tag, bounds, kw):
    """solve_bayesopt end to end (pybo/bayesopt.py:234-287: copy of the caller's model, centre point, then
    niter x [policy -> solver -> objective -> add_data -> recommender]) over a data-dependent stub model: the
    trace of queried points, values and recommendations equals the one the reference's own loop produced."""
    from helpers import SmootherModel, loop_objective
    g = np.load(os.path.join(G, 'loop.npz'))
    kw = dict(kw)
    kw.setdefault('solver', ('lbfgs', {'ngrid': 300}))
    mine = SmootherModel()
    xbest, model, info = bayesopt.solve_bayesopt(loop_objective, bounds, model=mine, niter=6, rng=4, **kw)
    assert len(mine.Y) == 0                               # the caller's model is never mutated
    assert len(model.Y) == int(g[tag + '_ndata']) == 7
    np.testing.assert_allclose(info.x, g[tag + '_x'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(info.y, g[tag + '_y'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(info.xbest, g[tag + '_xbest'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(xbest, g[tag + '_final'], rtol=0, atol=1e-12)


def _replay_loop_gp(case, make_model, solve):
    """One case of loop_gp.npz through pybo_amd.solve_bayesopt: returns (trace, per-iteration grid selections, fixture)."""
    from helpers import loop_gp_objective, recording_solver
    tag, bounds, okind, hyp, kern, kw, niter = case
    g = np.load(os.path.join(G, 'loop_gp.npz'))
    kw = dict(kw)
    name, skw = kw.pop('solver')
    log = []
    model = make_model(hyp[0], hyp[1], np.array(hyp[2]), hyp[3], kern)
    xbest, final, info = bayesopt.solve_bayesopt(loop_gp_objective(okind), bounds, model=model, niter=niter, rng=11,
                                                 solver=(recording_solver(solve, log), skw), **kw)
    per = len(log) // niter
    assert per * niter == len(log) and per == int(g[tag + '_stages_per_iter'])
    return (xbest, final, info), log[::per], {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + '_')}


@pytest.mark.parametrize('case', __import__('helpers').LOOP_GP_CASES, ids=lambda c: c[0])
def test_reference_loop_over_a_real_gp_is_reproduced(case):
    """loop_gp.npz = the REFERENCE's solve_bayesopt + solve_lbfgs + policies + recommenders (run in the build container by
    tests/golden/make_loop_gp.py) driving oracle.GPRef.  The same model under pybo_amd's loop, solver, policies and
    recommenders gives the same trace and the same grid selections: pybo's own glue, end to end over a real GP."""
    from oracle import gp_ref
    from pybo_amd import solvers
    (xbest, final, info), grid, g = _replay_loop_gp(case, gp_ref.make_gp, solvers.solve_lbfgs)
    assert final.ndata == case[6] + 1
    np.testing.assert_allclose(info.x, g['x'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(info.y, g['y'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(info.xbest, g['xbest'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(xbest, g['final'], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(np.array([t[0] for t in grid]), g['grid_top'])
    np.testing.assert_allclose([t[1] for t in grid], g['grid_best'], rtol=1e-12)


@pytest.mark.parametrize('tag,bounds,kw', [('b2', [[0.0, 1.0], [-0.5, 1.0]], {}),
                                           ('b3_n5', [[-5.0, 10.0], [0.0, 15.0], [1.0, 3.0]], {'ninit': 5}),
                                           ('flat', [[0.0, 1.0]], {})])
def test_init_model_matches_the_reference(tag, bounds, kw, monkeypatch):
    """init_model (pybo/bayesopt.py:60-118): the latin design drawn from the rng, the heuristic hyper-parameters
    (sn2 = 1e-6, rho = range of y with the 0.1 floor, ell = width / 4, bias = mean y), the four priors and
    MCMC(n=10, burn=100, rng=<the shared state>).  The model constructor and the sampler are replaced by
    recorders on both sides (reggie on the reference's, pybo_amd.models here): what is compared is what pybo
    itself decides."""
    from helpers import loop_objective
    from pybo_amd import models
    made = {}

    class Param(object):
        prior = None

        def set_prior(self, kind, *args):
            self.prior = (kind,) + tuple(np.array(a, dtype=float) for a in args)

    class Recorder(object):
        def __init__(self, sn2, rho, ell, bias, **_):
            self.args = (sn2, rho, np.array(ell, dtype=float), bias)
            self.params = {k: Param() for k in ('like.sn2', 'kern.rho', 'kern.ell', 'mean.bias')}

        def add_data(self, X, Y):
            self.data = (np.array(X, dtype=float), np.array(Y, dtype=float))

    def fake_gp(*a, **k):
        made['gp'] = Recorder(*a, **k)
        return made['gp']

    def fake_mcmc(model, n=None, burn=None, rng=None):
        made['mcmc'] = [n, burn, int(isinstance(rng, np.random.RandomState))]
        return model

    monkeypatch.setattr(models, 'make_gp', fake_gp)
    monkeypatch.setattr(models, 'MCMC', fake_mcmc)
    f = (lambda x: 0.5) if tag == 'flat' else loop_objective
    bayesopt.init_model(f, bounds, rng=9, **kw)
    g = np.load(os.path.join(G, 'init_model.npz'))
    gp = made['gp']
    hyp = np.concatenate([[gp.args[0], gp.args[1]], gp.args[2], [gp.args[3]]])
    np.testing.assert_allclose(hyp, g[tag + '_hypers'], rtol=1e-15, atol=0)
    np.testing.assert_array_equal(gp.data[0], g[tag + '_X'])
    np.testing.assert_array_equal(gp.data[1], g[tag + '_Y'])
    for name, prm in gp.params.items():
        assert prm.prior[0] == str(g['%s_prior_%s_kind' % (tag, name)])
        for j, a in enumerate(prm.prior[1:]):
            np.testing.assert_allclose(np.atleast_1d(a), g['%s_prior_%s_%d' % (tag, name, j)], rtol=1e-15)
    assert made['mcmc'] == g[tag + '_mcmc'].tolist()
