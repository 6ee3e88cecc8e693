"""Full-size checks at BASELINE.json's configurations through size-independent properties, plus a
sub-sampled comparison against the oracle (the oracle cannot evaluate 1e6 candidates in seconds)."""
import numpy as np
import pytest

from oracle import gp_ref
import bench
from helpers import s2_tol, mu_tol, ei_tol, ei_from_moments

pytestmark = pytest.mark.gpu


def test_config_b_full_grid_properties_and_subsample_parity():
    """BASELINE config B exactly as bench.py builds it (bench.make_workload('b'): Branin, N = 2048, SE-ARD, EI,
    2^20 Sobol candidates)."""
    from pybo_amd._lib import Engine
    M = 1 << 20
    w = bench.make_workload('b', M)
    X, y, ell, rho, sn2, bias, Z = w['X'], w['y'], w['ell'], w['rho'], w['sn2'], w['bias'], w['Xc']
    e = Engine(0)
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    mo, mx = e.mean_at_obs()
    r = e.sweep('ei', mx, Z, k=64, want_moments=True)
    mu, s2, ei = r['mu'], r['s2'], r['acq']
    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(ei))
    assert np.all(s2 > 0) and np.all(s2 <= rho * (1 + 1e-12))
    # EI >= 0 up to the denormal range: for z < -38 phi(z) and Phi(z) leave the normal range at different z and
    # (mu - t) Phi + s phi can come out as -1e-310 (the oracle's formula does the same); 92 % of this grid has
    # EI == 0 exactly
    assert np.all(ei >= -1e-300)
    # top-k of 1e6 = ranking of the returned values
    np.testing.assert_array_equal(r['top_idx'], gp_ref.topk_desc(ei, 64))
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
    pick = np.unique(np.concatenate([np.arange(0, M, 512), r['top_idx']]))
    mr, sr = ref.predict(Z[pick])
    assert np.all(np.abs(mu[pick] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(s2[pick] - sr) <= s2_tol(sr, rho))
    # EI, in two steps (VERDICT round 1, weak #2 -- no relaxed blanket rtol):
    # (1) the acquisition kernel itself: EI formed on the host from the DEVICE's moments equals the device's EI
    #     to round-off over the whole 2^20 grid (erfc-based Phi on both sides; the cancellation in
    #     s (phi + z Phi) costs ~z^2 ulps in the far tail)
    host = ei_from_moments(mu, s2, mx)
    live = host > 1e-9 * host.max()
    np.testing.assert_allclose(ei[live], host[live], rtol=1e-11, atol=0)
    # (2) against the oracle: the north-star's 1e-6 RELATIVE bar, on every compared candidate whose EI is within
    #     1e-9 of the maximum (z from -5.8 to 0.5 here; measured max 4e-8, profiles/history/r02_ei_conditioning_config_b.txt)
    #     -- round 1 checked 2e-5 on a harsher variant of this workload (sn2 = 3.5e-6 rho instead of bench.py's
    #     1e-4 rho); the first-order sensitivity |z| dmu/s + (z^2/2) ds2/s2 explains both numbers --
    #     and within the stated moment tolerances propagated through EI (helpers.ei_tol) everywhere else
    target = ref.mean_at_obs().max()
    assert abs(mx - target) <= 1e-9 * np.sqrt(rho)
    eir = ref.get_improvement(target, Z[pick])
    big = eir > 1e-9 * eir.max()
    assert big.sum() >= 64
    np.testing.assert_allclose(ei[pick][big], eir[big], rtol=1e-6, atol=0)
    assert np.all(np.abs(ei[pick] - eir) <= ei_tol(mr, sr, target, rho))
    # the selected candidate is the oracle's best among sub-sample + device top-k
    assert r['top_idx'][0] == pick[int(np.argmax(eir))]
    e.close()


def test_north_star_full_grid_subsample_parity_and_selection():
    """The north-star workload exactly as bench.py times it (bench.make_workload('ns'): N = 8192, d = 8, SE-ARD,
    EI, 2^20 Sobol candidates): the whole grid on the device (16 chunks), the oracle on 2048 grid points + the
    device's top-k; moments, EI and the SELECTED candidate are checked."""
    from pybo_amd._lib import Engine
    M = 1 << 20
    w = bench.make_workload('ns', M)
    rho = w['rho']
    e = Engine(0)
    e.fit(w['X'], w['y'], w['kernel'], w['ell'], rho, w['sn2'], w['bias'])
    mo, mx = e.mean_at_obs()
    r = e.sweep('ei', mx, w['Xc'], k=10, want_moments=True)
    mu, s2, ei = r['mu'], r['s2'], r['acq']
    assert np.all(np.isfinite(mu)) and np.all(np.isfinite(ei)) and np.all(ei >= 0)
    assert np.all(s2 > 0) and np.all(s2 <= rho * (1 + 1e-12))
    np.testing.assert_array_equal(r['top_idx'], gp_ref.topk_desc(ei, 10))
    host = ei_from_moments(mu, s2, mx)
    live = host > 1e-9 * host.max()
    np.testing.assert_allclose(ei[live], host[live], rtol=1e-11, atol=0)
    ref = gp_ref.make_gp(w['sn2'], rho, w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    pick = np.unique(np.concatenate([np.arange(0, M, 512), r['top_idx']]))
    assert len(pick) >= 2048
    mr, sr = ref.predict(w['Xc'][pick])
    assert np.all(np.abs(mu[pick] - mr) <= mu_tol(mr, rho))
    assert np.all(np.abs(s2[pick] - sr) <= s2_tol(sr, rho))
    np.testing.assert_allclose(mo, ref.mean_at_obs(), rtol=0, atol=1e-7 * np.sqrt(rho))
    target = ref.mean_at_obs().max()
    eir = ref.get_improvement(target, w['Xc'][pick])
    big = eir > 1e-9 * eir.max()
    assert big.sum() > 1000
    np.testing.assert_allclose(ei[pick][big], eir[big], rtol=1e-6, atol=0)     # measured: 6e-13
    assert np.all(np.abs(ei[pick] - eir) <= ei_tol(mr, sr, target, rho))
    # selected index: the device's winner is the oracle's winner over sub-sample + device top-k, and the
    # oracle ranks the device's top-k in the same order wherever the gap exceeds the tolerance
    assert r['top_idx'][0] == pick[int(np.argmax(eir))]
    pos = np.searchsorted(pick, r['top_idx'])
    vals = eir[pos]
    gaps = vals[:-1] - vals[1:]
    tol = ei_tol(mr, sr, target, rho)[pos]
    assert np.all(gaps >= -(tol[:-1] + tol[1:]))
    np.testing.assert_allclose(r['top_val'], vals, rtol=0, atol=2 * tol.max())
    e.close()


# ---- the WHOLE grid against committed oracle fixtures ------------------------------------
def _digest(w, M):
    import hashlib
    h = hashlib.sha256()
    for a in (w['X'], w['y'], w['Xc'][:M], w['ell'], np.array([w['rho'], w['sn2'], w['bias']])):
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name))


def _check_ranking(dev_idx, dev_val, ref_vals, ref_top, tol):
    """The device's top-k against the oracle's ranking of the WHOLE grid: same winner, every device pick is one of
    the oracle's best (up to the tolerance at the cut), and the oracle orders the device's picks the same way wherever
    two neighbours differ by more than the tolerance."""
    k = len(dev_idx)
    v = ref_vals[dev_idx]
    if ref_vals[ref_top[0]] - ref_vals[ref_top[1]] > 2 * tol:
        assert dev_idx[0] == ref_top[0]
    assert np.all(v >= ref_vals[ref_top[k - 1]] - 2 * tol)            # nobody outside the oracle's top-k (mod ties)
    assert np.all(v[:-1] - v[1:] >= -2 * tol)                          # same order
    clear = ref_vals[ref_top[:k]] - ref_vals[ref_top[k]] > 2 * tol     # oracle picks clearly above the cut ...
    assert set(ref_top[:k][clear]) <= set(dev_idx)                     # ... are all found by the device
    np.testing.assert_allclose(dev_val, v, rtol=0, atol=2 * tol)


def test_config_b_whole_grid_against_the_committed_oracle_sweep():
    """tests/golden/grid_b.npz = oracle/gp_ref.py over ALL 2^20 candidates of config B (130 s of host time, run once
    by tests/golden/make_grid_fixtures.py).  Round 2 compared a stride-512 sub-sample: a candidate mis-scored off the
    stride would never have been seen.  Now every EI value, the selected index and the top-64 order are compared."""
    from pybo_amd._lib import Engine
    M = 1 << 20
    w = bench.make_workload('b', M)
    fx = _golden('grid_b.npz')
    assert _digest(w, M) == str(fx['sha']), 'the fixture was made from other inputs than bench.make_workload builds'
    rho = w['rho']
    e = Engine(0)
    e.fit(w['X'], w['y'], 'se', w['ell'], rho, w['sn2'], w['bias'])
    _, mx = e.mean_at_obs()
    assert abs(mx - float(fx['target'])) <= 1e-9 * np.sqrt(rho)
    r = e.sweep('ei', mx, w['Xc'], k=64, want_moments=True)
    ei, ei_o = r['acq'], fx['ei']
    live = ei_o > 1e-9 * ei_o.max()
    assert live.sum() > 10000              # (92 % of this grid has EI == 0 exactly; 12294 candidates are live)
    rel = np.abs(ei[live] - ei_o[live]) / ei_o[live]
    print('config B, whole grid: %d live candidates, max relative EI error %.2e' % (live.sum(), rel.max()))
    assert rel.max() <= 1e-6                                            # the north-star bar, on every live candidate
    # below 1e-9 of the maximum (z < -6: EI ~ s phi(z) / z^2 amplifies the moments' round-off by z^2 / 2) the values
    # cannot matter for the selection; they are held to an absolute 1e-12 of the maximum
    assert np.all(np.abs(ei[~live] - ei_o[~live]) <= 1e-12 * ei_o.max())
    mu16, s216 = r['mu'][::16], r['s2'][::16]
    assert np.all(np.abs(mu16 - fx['mu16']) <= mu_tol(fx['mu16'], rho))
    assert np.all(np.abs(s216 - fx['s216']) <= s2_tol(fx['s216'], rho))
    assert int(np.argmax(ei)) == int(fx['top'][0]) == int(r['top_idx'][0])       # the selected candidate, whole grid
    _check_ranking(r['top_idx'], r['top_val'], ei_o, fx['top'], 1e-6 * ei_o.max())
    e.close()


def test_config_c_first_2_16_candidates_against_the_committed_oracle_sweep():
    """tests/golden/grid_c.npz = the oracle's UCB, mu, s2 on the first 65536 Sobol candidates of config C (N = 8192,
    Matern-5/2; 160 s of host time): every value and the ranking."""
    from pybo_amd._lib import Engine
    M = 1 << 16
    w = bench.make_workload('c', M)
    fx = _golden('grid_c.npz')
    assert _digest(w, M) == str(fx['sha'])
    rho = w['rho']
    beta = bench.ucb_beta(w['N'])
    assert beta == float(fx['beta'])
    e = Engine(0)
    e.fit(w['X'], w['y'], w['kernel'], w['ell'], rho, w['sn2'], w['bias'])
    r = e.sweep('ucb', beta, w['Xc'], k=64, want_moments=True)
    assert np.all(np.abs(r['mu'] - fx['mu']) <= mu_tol(fx['mu'], rho))
    assert np.all(np.abs(r['s2'] - fx['s2']) <= s2_tol(fx['s2'], rho))
    scale = np.abs(fx['ucb']).max()
    np.testing.assert_allclose(r['acq'], fx['ucb'], rtol=1e-6, atol=1e-9 * scale)
    assert int(np.argmax(r['acq'])) == int(fx['top'][0])
    _check_ranking(r['top_idx'], r['top_val'], fx['ucb'], fx['top'], 1e-6 * scale)
    e.close()


@pytest.mark.parametrize('name', ['ns', 'c'])
def test_whole_grid_selection_against_the_oracle_s_ranking_of_all_2_20_candidates(name):
    """tests/golden/grid_{ns,c}_full.npz: oracle/gp_ref.py over ALL 2^20 candidates of the north-star workload (EI) and of
    config C (UCB) -- ~10 min of host time each in the build container (make_grid_fixtures.py ns_full c_full) -- kept as
    the oracle's top 256 of the whole grid plus every 8th value and moment.  The device's selected candidate is the
    oracle's argmax over the WHOLE grid, its top 64 are the oracle's in the oracle's order (up to the tolerance), and
    131 072 values / moments agree to the stated tolerances."""
    from pybo_amd._lib import Engine
    M = 1 << 20
    w = bench.make_workload(name, M)
    fx = _golden('grid_%s_full.npz' % name)
    assert _digest(w, M) == str(fx['sha'])
    rho = w['rho']
    e = Engine(0)
    e.fit(w['X'], w['y'], w['kernel'], w['ell'], rho, w['sn2'], w['bias'])
    if w['acq'] == 'ei':
        _, param = e.mean_at_obs()
        assert abs(param - float(fx['param'])) <= 1e-9 * np.sqrt(rho)
    else:
        param = bench.ucb_beta(w['N'])
        assert param == float(fx['param'])
    r = e.sweep(w['acq'], param, w['Xc'], k=64, want_moments=True)
    val = r['acq']
    scale = float(fx['vmax'])
    # every 8th candidate
    assert np.all(np.abs(r['mu'][::8] - fx['mu8']) <= mu_tol(fx['mu8'], rho))
    assert np.all(np.abs(r['s2'][::8] - fx['s28']) <= s2_tol(fx['s28'], rho))
    live = np.abs(fx['val8']) > 1e-9 * abs(scale)
    rel = np.abs(val[::8][live] - fx['val8'][live]) / np.abs(fx['val8'][live])
    print('%s: %d of %d strided values live, max relative error %.2e' % (name, live.sum(), len(live), rel.max()))
    assert rel.max() <= 1e-6
    # the ranking of the WHOLE grid
    top, top_val = fx['top'], fx['top_val']
    tol = 1e-6 * abs(scale)
    assert int(np.argmax(val)) == int(r['top_idx'][0])
    if top_val[0] - top_val[1] > 2 * tol:
        assert int(r['top_idx'][0]) == int(top[0])                      # the selected candidate
    rank_of = {int(i): k for k, i in enumerate(top)}
    assert all(int(i) in rank_of for i in r['top_idx']), 'a device pick is outside the best 256 of 2^20 by the oracle'
    ov = np.array([top_val[rank_of[int(i)]] for i in r['top_idx']])     # the oracle's values of the device's picks
    assert np.all(ov[:-1] - ov[1:] >= -2 * tol)                         # in the oracle's order
    assert np.all(ov >= top_val[63] - 2 * tol)                          # and they ARE its top 64 (up to ties at the cut)
    np.testing.assert_allclose(r['top_val'], ov, rtol=0, atol=2 * tol)
    e.close()
