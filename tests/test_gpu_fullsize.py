"""Full-size checks at BASELINE.json's configurations through size-independent properties, plus a
sub-sampled comparison against the oracle (the oracle cannot evaluate 1e6 candidates in seconds)."""
import numpy as np
import pytest

from oracle import gp_ref
from helpers import branin, s2_tol, mu_tol

pytestmark = pytest.mark.gpu


def _config_b(N=2048, seed=0):
    rng = np.random.RandomState(seed)
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    X = lo + (hi - lo) * rng.rand(N, 2)
    y = -branin(X) / 10.0 + 1e-3 * rng.randn(N)
    ell = 0.25 * (hi - lo)
    rho, bias = float(np.var(y)), float(np.mean(y))
    return X, y, ell, rho, 1e-4, bias, lo, hi


def test_config_b_full_grid_properties_and_subsample_parity():
    from pybo_amd._lib import Engine
    from scipy.stats import qmc
    X, y, ell, rho, sn2, bias, lo, hi = _config_b()
    M = 1 << 20
    Z = lo + (hi - lo) * qmc.Sobol(2, scramble=False).random(M)
    e = Engine(0)
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    mo, mx = e.mean_at_obs()
    r = e.sweep('ei', mx, Z, k=10, want_moments=True)
    mu, s2, ei = r['mu'], r['s2'], r['acq']
    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(ei))
    assert np.all(s2 > 0) and np.all(s2 <= rho * (1 + 1e-12))
    assert np.all(ei >= 0)
    # top-k of 1e6 = ranking of the returned values
    np.testing.assert_array_equal(r['top_idx'], gp_ref.topk_desc(ei, 10))
    # chunk-size independence, bitwise
    e.set_option('chunk', 16384)
    r2 = e.sweep('ei', mx, Z[: 1 << 17], k=3, want_moments=True)
    assert np.array_equal(r2['mu'], mu[: 1 << 17]) and np.array_equal(r2['s2'], s2[: 1 << 17])
    # sweeping the observed points reproduces the closed-form mean y - sn2*alpha
    r3 = e.sweep('mean', None, X, k=1)
    np.testing.assert_allclose(r3['acq'], mo, rtol=0, atol=1e-7 * np.sqrt(rho))
    # linearity of the posterior mean in y (bias = 0): mu[y1 + y2] = mu[y1] + mu[y2]
    sub = Z[:: 4096]
    y1, y2 = y - bias, np.sin(X.sum(1))
    mus = []
    for yy in (y1, y2, y1 + y2):
        e.fit(X, yy, 'se', ell, rho, sn2, 0.0)
        mus.append(e.predict(sub)[0])
    np.testing.assert_allclose(mus[0] + mus[1], mus[2], rtol=0, atol=1e-8)
    # oracle parity on a sub-sample of the grid + the selected candidate
    ref = gp_ref.make_gp(sn2, rho, ell, bias)
    ref.add_data(X, y)
    pick = np.unique(np.concatenate([np.arange(0, M, 1024), r['top_idx']]))
    mr, sr = ref.predict(Z[pick])
    assert np.all(np.abs(mu[pick] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(s2[pick] - sr) <= s2_tol(sr, rho))
    eir = ref.get_improvement(ref.mean_at_obs().max(), Z[pick])
    big = eir > 1e-9 * eir.max()
    np.testing.assert_allclose(ei[pick][big], eir[big], rtol=2e-5)   # EI amplifies ds2 by ~z^2
    e.close()


def test_north_star_size_subsample_parity():
    """N = 8192, d = 8 (the north-star target shape): full fit, 2^17 candidates on the device, oracle on a
    256-candidate sub-sample (the oracle's fit alone is ~4 s)."""
    from pybo_amd._lib import Engine
    rng = np.random.RandomState(2)
    N, d, M = 8192, 8, 1 << 17
    X = rng.rand(N, d)
    y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.25 * np.ones(d)
    rho, bias = float(np.var(y)), float(np.mean(y))
    sn2 = 1e-4 * rho
    Z = rng.rand(M, d)
    e = Engine(0)
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    mo, mx = e.mean_at_obs()
    r = e.sweep('ei', mx, Z, k=10, want_moments=True)
    assert np.all(r['s2'] > 0) and np.all(r['s2'] <= rho * (1 + 1e-12))
    np.testing.assert_array_equal(r['top_idx'], gp_ref.topk_desc(r['acq'], 10))
    ref = gp_ref.make_gp(sn2, rho, ell, bias)
    ref.add_data(X, y)
    pick = np.unique(np.concatenate([np.arange(0, M, 512), r['top_idx']]))
    mr, sr = ref.predict(Z[pick])
    assert np.all(np.abs(r['mu'][pick] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(r['s2'][pick] - sr) <= s2_tol(sr, rho))
    np.testing.assert_allclose(mo, ref.mean_at_obs(), rtol=0, atol=1e-7 * np.sqrt(rho))
    e.close()
