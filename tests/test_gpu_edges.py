"""Edge cases of the hot path through the C-ABI: the smallest models (N = 1, 2, 3), sizes around the 128-block
boundary, a single candidate, k larger than the candidate count, the widest supported inputs, exact ties, NaN
candidates, candidates that coincide with observations, a duplicated observation.  Moments against the oracle with
the tolerances of SURVEY 8(d); the selected candidate identical."""
import numpy as np
import pytest

from oracle import gp_ref
from helpers import s2_tol, mu_tol

pytestmark = pytest.mark.gpu


def _case(N, d, M, kernel, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.rand(N, d)
    y = np.sin(X.sum(1)) + 0.1 * rng.randn(N)
    ell = np.full(d, 0.4)
    Z = rng.rand(M, d)
    ref = gp_ref.make_gp(1e-3, 1.3, ell, 0.2, kernel)
    ref.add_data(X, y)
    return X, y, ell, Z, ref


@pytest.mark.parametrize('N,d,M,k,kernel', [
    (1, 1, 5, 3, 'se'), (2, 1, 5, 3, 'se'), (3, 1, 5, 3, 'matern5'), (127, 1, 5, 3, 'se'), (128, 1, 5, 3, 'se'),
    (129, 1, 5, 3, 'matern3'), (5, 3, 1, 1, 'se'), (5, 3, 1, 0, 'se'), (200, 2, 7, 7, 'se'), (200, 2, 7, 10, 'se'),
    (130, 64, 300, 5, 'se'), (50, 2, 129, 64, 'matern1'), (1, 2, 1, 1, 'matern5')])
def test_small_and_boundary_sizes(N, d, M, k, kernel):
    from pybo_amd._lib import Engine
    X, y, ell, Z, ref = _case(N, d, M, kernel)
    e = Engine(0)
    e.fit(X, y, kernel, ell, 1.3, 1e-3, 0.2)
    _, target = e.mean_at_obs()
    assert abs(target - ref.mean_at_obs().max()) <= 1e-9
    r = e.sweep('ei', target, Z, k=k, want_moments=True)
    mr, sr = ref.predict(Z)
    assert np.all(np.abs(r['mu'] - mr) <= mu_tol(mr, 1.3)) and np.all(np.abs(r['s2'] - sr) <= s2_tol(sr, 1.3))
    eir = ref.get_improvement(ref.mean_at_obs().max(), Z)
    if k:
        kk = min(k, M)
        assert r['top_idx'][0] == int(np.argmax(eir))
        assert np.all(r['top_idx'][:kk] >= 0) and np.all(r['top_idx'][kk:] == -1)     # k > M: padded with -1
        assert np.all(np.diff(r['top_val'][:kk]) <= 0)
    g, gr = e.predict(Z[:min(M, 3)], grad=True), ref.predict(Z[:min(M, 3)], grad=True)
    for a, b in zip(g, gr):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-8)
    e.close()


def test_ties_nan_candidates_and_candidates_at_observations():
    from pybo_amd._lib import Engine, GpxError
    rng = np.random.RandomState(1)
    X, y = rng.rand(40, 2), rng.randn(40)
    e = Engine(0)
    e.fit(X, y, 'se', [0.3, 0.3], 1.0, 1e-2, 0.0)
    base = rng.rand(5, 2)
    r = e.sweep('ucb', 2.0, np.tile(base, (4, 1)), k=6)              # every candidate four times
    best = int(np.argmax(e.sweep('ucb', 2.0, base, k=0)['acq']))
    np.testing.assert_array_equal(r['top_idx'][:4], best + 5 * np.arange(4))      # exact ties: ascending index
    assert len(set(r['top_val'][:4])) == 1
    Z = rng.rand(6, 2)
    Z[2, 0] = np.nan
    r = e.sweep('ei', 0.1, Z, k=6)
    assert r['top_idx'][-1] == 2 and np.all(np.isfinite(r['top_val'][:5]))          # NaN ranks last
    s2 = e.sweep('ei', 0.1, X[:3], k=0, want_moments=True)['s2']                   # candidates = observations
    ref = gp_ref.make_gp(1e-2, 1.0, [0.3, 0.3], 0.0)
    ref.add_data(X, y)
    np.testing.assert_allclose(s2, ref.predict(X[:3])[1], rtol=1e-6, atol=1e-10)
    assert np.all(s2 > 0) and np.all(s2 < 1e-2)
    assert e.append(X[0], y[0]) and e.N == 41                                       # a duplicated observation
    ref.add_data(X[:1], y[:1])
    mu = e.sweep('mean', None, base, k=0, want_moments=True)['mu']
    np.testing.assert_allclose(mu, ref.predict(base)[0], rtol=1e-6, atol=1e-9)
    for call in (lambda: e.sweep('ei', 0.1, np.zeros((0, 2)), k=0), lambda: e.predict(np.zeros((0, 2)))):
        with pytest.raises(GpxError):                                               # the library refuses M = 0
            call()
    e.close()


def test_gpx_options_environment_applies_to_new_handles_and_fails_loudly(monkeypatch):
    """GPX_OPTIONS="name=value,..." (include/gpx.h): options every new handle starts with -- how an A/B run reaches the
    handles the plug-in layer creates.  A bad entry must fail the creation, never be ignored."""
    from pybo_amd._lib import Engine, GpxError
    from helpers import synth_problem
    X, y, ell = synth_problem(200, 3, seed=5)
    Z = np.random.RandomState(2).rand(1, 3)
    res = {}
    for env in ('', 'grad_form=1', 'grad_form=2,grad_rb_cs=1024'):
        monkeypatch.setenv('GPX_OPTIONS', env)
        e = Engine(0)
        e.fit(X, y, 'se', ell, 1.0, 1e-3, 0.0)
        res[env] = e.predict(Z, grad=True)
        e.close()
    # a single point: auto = the one-pass form (what grad_form=2 pins); grad_form=1 is the two-pass form, equal to rounding
    for a, b in zip(res[''], res['grad_form=2,grad_rb_cs=1024']):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
    for a, b in zip(res[''], res['grad_form=1']):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    for bad in ('no_such_option=1', 'grad_form=7', 'grad_form', 'grad_form=x'):
        monkeypatch.setenv('GPX_OPTIONS', bad)
        with pytest.raises(GpxError):
            Engine(0)
    monkeypatch.delenv('GPX_OPTIONS')
    Engine(0).close()
