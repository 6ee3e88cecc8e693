"""The MCMC meta-model (pybo_amd.models.MCMC) is host logic over any member model with hyper-parameter access;
on CPU it is driven with the oracle model."""
import pickle

import numpy as np
import pytest

from oracle import gp_ref
from pybo_amd.models.mcmc import MCMC
from pybo_amd.models.priors import log_prior
from pybo_amd import solve_bayesopt


def _problem(n=40, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.rand(n, 2)
    K = gp_ref.kernel(0, X, X, np.array([0.2, 0.6]), 2.0) + 1e-2 * np.eye(n)
    y = np.linalg.cholesky(K) @ rng.randn(n) + 0.5
    m = gp_ref.make_gp(1e-3, 1.0, [0.4, 0.4], 0.0)
    m.params['like.sn2'].set_prior('horseshoe', 0.1)
    m.params['kern.rho'].set_prior('lognormal', 0.0, 1.0)
    m.params['kern.ell'].set_prior('uniform', [0.01, 0.01], [3.0, 3.0])
    m.params['mean.bias'].set_prior('normal', 0.0, 4.0)
    m.add_data(X, y)
    return m, X, y


def test_prior_log_densities():
    assert log_prior(None, 3.0) == 0.0
    assert log_prior(('uniform', 0.0, 1.0), [0.2, 0.9]) == 0.0
    assert log_prior(('uniform', [0.0, 0.5], [1.0, 1.0]), [0.2, 0.4]) == -np.inf
    assert abs(log_prior(('normal', 1.0, 4.0), 3.0) - (-0.5)) < 1e-15
    assert abs(log_prior(('lognormal', 0.0, 1.0), np.e) - (-0.5 - 1.0)) < 1e-15
    assert log_prior(('lognormal', 0.0, 1.0), -1.0) == -np.inf
    hs = lambda x: log_prior(('horseshoe', 0.1), x)          # noqa: E731
    assert hs(1e-3) > hs(1e-1) > hs(10.0) and hs(-1.0) == -np.inf
    assert abs(hs(0.1) - np.log(np.log(4.0))) < 1e-15
    with pytest.raises(ValueError):
        log_prior(('cauchy', 1.0), 1.0)


def test_loglikelihood_matches_the_dense_formula():
    m, X, y = _problem()
    K = m.gram()
    r = y - m.bias
    want = -0.5 * r @ np.linalg.solve(K, r) - 0.5 * np.linalg.slogdet(K)[1] - 0.5 * len(y) * np.log(2 * np.pi)
    assert abs(m.loglikelihood() - want) < 1e-9 * abs(want)


def test_chain_is_seed_deterministic_and_finds_the_generating_scales():
    m, X, y = _problem()
    a = MCMC(m, n=10, burn=100, rng=1)
    b = MCMC(m, n=10, burn=100, rng=1)
    np.testing.assert_array_equal(a.samples, b.samples)
    assert a.samples.shape == (10, 5) and m.sn2 == 1e-3            # the caller's model is untouched
    s = MCMC(m, n=60, burn=200, rng=2).samples
    ell = np.exp(s[:, 2:4]).mean(0)
    assert 0.1 < ell[0] < 0.4 and 0.3 < ell[1] < 1.5               # generated with (0.2, 0.6)
    assert ell[0] < ell[1]
    assert 1e-3 < np.exp(s[:, 0]).mean() < 1e-1                    # generated with 1e-2


def test_ensemble_moments_and_acquisitions_are_member_averages():
    m, X, y = _problem()
    mc = MCMC(m, n=6, burn=30, rng=3)
    Z = np.random.RandomState(4).rand(7, 2)
    mus = np.array([g.predict(Z)[0] for g in mc._members])
    s2s = np.array([g.predict(Z)[1] for g in mc._members])
    mu, s2 = mc.predict(Z)
    np.testing.assert_allclose(mu, mus.mean(0), rtol=1e-14)
    np.testing.assert_allclose(s2, (s2s + mus ** 2).mean(0) - mu ** 2, rtol=1e-12)
    np.testing.assert_allclose(mc.get_improvement(0.7, Z),
                               np.mean([g.get_improvement(0.7, Z) for g in mc._members], axis=0), rtol=1e-14)
    np.testing.assert_allclose(mc.get_tail(0.7, Z),
                               np.mean([g.get_tail(0.7, Z) for g in mc._members], axis=0), rtol=1e-14)
    # gradients of the mixture moments by finite differences
    mu, s2, dmu, ds2 = mc.predict(Z, grad=True)
    h = 1e-6
    for j in range(2):
        E = np.zeros(2); E[j] = h
        mp, sp = mc.predict(Z + E)
        mm, sm = mc.predict(Z - E)
        np.testing.assert_allclose(dmu[:, j], (mp - mm) / (2 * h), rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(ds2[:, j], (sp - sm) / (2 * h), rtol=1e-4, atol=1e-8)
    assert not hasattr(mc, 'acq_topk')              # host members: policies fall back to index(xgrid)
    f = mc.sample_f(50, rng=0).get(Z)
    assert f.shape == (7,)


def test_add_data_continues_the_chain_and_copy_is_independent_in_data():
    m, X, y = _problem()
    mc = MCMC(m, n=5, burn=20, rng=5)
    c = mc.copy()
    before = mc.samples.copy()
    c.add_data(np.array([0.5, 0.5]), 0.3)
    assert c.ndata == 41 and mc.ndata == 40
    np.testing.assert_array_equal(mc.samples, before)
    assert not np.array_equal(c.samples, before)
    back = pickle.loads(pickle.dumps(c))
    # samples are read back as log(exp(theta)): one ulp of round trip per pickle generation
    np.testing.assert_allclose(back.samples, c.samples, rtol=4e-16, atol=0)
    Z = np.random.RandomState(0).rand(3, 2)
    np.testing.assert_allclose(back.predict(Z)[0], c.predict(Z)[0], rtol=1e-12)


def test_bo_loop_runs_on_the_ensemble():
    bounds = [[0.0, 1.0], [0.0, 1.0]]
    f = lambda x: float(-np.sum((np.asarray(x) - 0.3) ** 2))        # noqa: E731
    m = gp_ref.make_gp(1e-4, 1.0, [0.3, 0.3], 0.0)
    m.params['kern.ell'].set_prior('uniform', [0.05, 0.05], [2.0, 2.0])
    m.params['kern.rho'].set_prior('lognormal', 0.0, 1.0)
    m.params['like.sn2'].set_prior('horseshoe', 0.1)
    X0 = np.random.RandomState(1).rand(6, 2)
    m.add_data(X0, [f(x) for x in X0])
    mc = MCMC(m, n=4, burn=20, rng=1)          # (a 4-member ensemble after 20 updates: outcome is seed-dependent)
    xb, mm, info = solve_bayesopt(f, bounds, model=mc, niter=5, solver=('lbfgs', {'ngrid': 300, 'nbest': 2}),
                                  rng=2)
    assert mm.ndata == 6 + 1 + 5 and info.y.max() > -0.05


def test_slice_update_leaves_an_analytic_target_invariant():
    """The update is checked on its own, on a target with known moments: a correlated 3-d normal whose third
    coordinate has a very different scale (the role the bias plays next to the log-parameters).  A direction
    scale that depended on the current state would bias these moments (+0.49 in the mean of
    log rho on a prior-only target)."""
    from pybo_amd.models.mcmc import _slice_update
    mean = np.array([0.0, 1.0, -2.0])
    A = np.array([[1.5, 0.0, 0.0], [0.6, 0.8, 0.0], [0.0, 1.0, 3.0]])
    cov = A @ A.T
    P = np.linalg.inv(cov)
    logp = lambda th: float(-0.5 * (th - mean) @ P @ (th - mean))      # noqa: E731
    rng = np.random.RandomState(11)
    scale = np.array([1.0, 1.0, 3.0])
    th, lp = mean.copy(), 0.0
    draws = np.empty((12000, 3))
    for i in range(len(draws)):
        th, lp = _slice_update(logp, th, lp, rng, scale)
        assert abs(lp - logp(th)) < 1e-12
        draws[i] = th
    draws = draws[500:]
    # effective sample size of this chain is ~ a third of its length: 4-sigma bands
    se = np.sqrt(np.diag(cov) / (len(draws) / 3.0))
    assert np.all(np.abs(draws.mean(0) - mean) < 4.0 * se), (draws.mean(0), mean)
    np.testing.assert_allclose(np.cov(draws.T), cov, rtol=0.12, atol=0.12)


def test_direction_scale_is_frozen_at_construction():
    m, X, y = _problem()
    s = MCMC(m, n=4, burn=10, rng=0)
    want = np.array([1.0, 1.0, 1.0, 1.0, np.sqrt(m.rho)])
    np.testing.assert_array_equal(s._scale, want)
    s.add_data(np.array([[0.3, 0.3]]), [0.1])
    np.testing.assert_array_equal(s._scale, want)          # the chain's own rho moved, the scale did not
    np.testing.assert_array_equal(pickle.loads(pickle.dumps(s))._scale, want)
    np.testing.assert_array_equal(s.copy()._scale, want)


@pytest.mark.parametrize('spec', [1, 2, 4, 7])
def test_speculative_batched_slice_update_is_the_sequential_one(spec):
    """With a batched density (`logp_many`, the device path) the stepping-out ends and the next `spec` shrinkage
    candidates are evaluated together; the chain, the returned densities and the random stream must be EXACTLY
    those of the one-at-a-time procedure."""
    from pybo_amd.models.mcmc import _slice_update
    mean = np.array([0.0, 1.0, -2.0])
    A = np.array([[1.5, 0.0, 0.0], [0.6, 0.8, 0.0], [0.0, 1.0, 3.0]])
    P = np.linalg.inv(A @ A.T)
    calls = []
    logp = lambda th: float(-0.5 * (th - mean) @ P @ (th - mean))      # noqa: E731

    def many(ths):
        calls.append(len(ths))
        return np.array([logp(t) for t in ths])
    r1, r2 = np.random.RandomState(3), np.random.RandomState(3)
    scale = np.array([1.0, 1.0, 3.0])
    a, la = mean.copy(), 0.0
    b, lb = mean.copy(), 0.0
    for _ in range(800):
        a, la = _slice_update(logp, a, la, r1, scale)
        b, lb = _slice_update(logp, b, lb, r2, scale, logp_many=many, spec=spec)
        assert np.array_equal(a, b) and la == lb
    assert r1.rand() == r2.rand()
    if spec >= 4:
        assert np.mean(calls) > 1.5          # the batches really carry several states


def test_ensemble_objects_are_freed_by_reference_counting_alone():
    """Every BO step copies the ensemble several times (pybo/bayesopt.py:249, pybo/policies/simple.py:20); a copy
    that sat in a reference cycle kept its member models -- and their device handles -- until the cyclic collector
    ran, which emptied the handle pool (80 % of a default run was spent creating and destroying handles)."""
    import gc
    import weakref
    m, X, y = _problem(n=25)
    gc.collect()
    gc.disable()
    try:
        mc = MCMC(m, n=3, burn=5, rng=0)
        cp = mc.copy()
        refs = [weakref.ref(mc), weakref.ref(cp), weakref.ref(cp._members[0])]
        assert getattr(mc, 'acq_topk', None) is None          # host members: no whole-grid hook, as before
        del mc, cp
        assert all(r() is None for r in refs)
    finally:
        gc.enable()
