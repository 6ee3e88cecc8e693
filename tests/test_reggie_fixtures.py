"""
Parity against `reggie` itself -- the package that holds the arithmetic behind every `model.*` call of pybo
(pybo/bayesopt.py:105-115, pybo/policies/simple.py:20-64, pybo/recommenders.py:22-34).

The fixtures tests/golden/reggie_*.npz are written by tests/golden/make_reggie_fixtures.py where `reggie` can be imported.
It cannot in the build container (requirements.txt:8 is a bare git URL, there is no network), so today every case here is an
EXPECTED FAILURE with that reason -- the visible marker of "parity unpinned" -- and turns into a real comparison the day the
files exist: the CPU cases hold the oracle (oracle/gp_ref.py) to reggie's numbers, the -m gpu cases the device path.
Tolerances: SURVEY.md section 8(d).
"""
import os

import numpy as np
import pytest

from helpers import mu_tol, s2_tol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
WHY = ('tests/golden/reggie_%s.npz is absent: `reggie` (requirements.txt:8, unpinned git dependency) is not importable in the '
       'build container; run tests/golden/make_reggie_fixtures.py where it is -- parity against reggie is UNPINNED until then')
TAGS = ('p1d', 'p2d', 'p6d')


def _fixture(kind):
    path = os.path.join(G, 'reggie_%s.npz' % kind)
    if not os.path.exists(path):
        pytest.xfail(WHY % kind)
    return np.load(path)


def _models(g, tag, device):
    hyp = g[tag + '_hyp']
    d = g[tag + '_X'].shape[1]
    if device:
        from pybo_amd import models
        m = models.make_gp(hyp[0], hyp[1], hyp[2:2 + d], hyp[2 + d])
    else:
        from oracle import gp_ref
        m = gp_ref.make_gp(hyp[0], hyp[1], hyp[2:2 + d], hyp[2 + d])
    m.add_data(g[tag + '_X'], g[tag + '_y'])
    return m, hyp


def _check_predict(device):
    g = _fixture('predict')
    for tag in TAGS:
        m, hyp = _models(g, tag, device)
        mu, s2, dmu, ds2 = m.predict(g[tag + '_Z'], grad=True)
        assert np.all(np.abs(mu - g[tag + '_mu']) <= mu_tol(g[tag + '_mu'], hyp[1])), tag
        assert np.all(np.abs(s2 - g[tag + '_s2']) <= s2_tol(g[tag + '_s2'], hyp[1])), \
            '%s: predictive variance differs -- latent vs noisy (s2 + sn2)? max diff %g, sn2 %g' % (
                tag, np.abs(s2 - g[tag + '_s2']).max(), hyp[0])
        np.testing.assert_allclose(dmu, g[tag + '_dmu'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(ds2, g[tag + '_ds2'], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(m.predict(g[tag + '_X'])[0], g[tag + '_mu_obs'], rtol=1e-6, atol=1e-8)


def _check_acq(kind, device):
    g = _fixture(kind)
    for tag in TAGS:
        m, hyp = _models(g, tag, device)
        f = m.get_improvement if kind == 'ei' else m.get_tail
        target = float(g[tag + '_target'])
        want = g[tag + '_val']
        big = np.abs(want) > 1e-12 * np.abs(want).max()
        np.testing.assert_allclose(f(target, g[tag + '_Z'])[big], want[big], rtol=1e-6)
        v, dv = f(target, g[tag + '_Z'], True)
        np.testing.assert_allclose(v[big], g[tag + '_val_g'][big], rtol=1e-6)
        np.testing.assert_allclose(dv, g[tag + '_grad'], rtol=1e-5, atol=1e-9)


def _check_sample_f(device):
    g = _fixture('sample_f')
    for tag in TAGS:
        m, hyp = _models(g, tag, device)
        for s in (0, 7):
            f = m.sample_f(100, np.random.RandomState(s))
            want = g['%s_seed%d_val' % (tag, s)]
            got = f.get(g[tag + '_Z'])
            assert np.allclose(got, want, rtol=1e-6, atol=1e-8), (
                '%s seed %d: the drawn function differs from reggie\'s for the same RandomState -- draw order / scaling of '
                'W, b or of the weight posterior differs; Thompson parity is then distributional, not point-wise' % (tag, s))


def _check_loglik(device):
    g = _fixture('loglik')
    keys = [k for k in g.files if k.endswith('_loglikelihood') or k.endswith('_get_loglike')]
    if not keys:
        pytest.skip('the reggie build that wrote the fixtures exposes no log-likelihood accessor')
    for k in keys:
        tag = k.split('_')[0]
        m, hyp = _models(g, tag, device)
        ll = m.loglikelihood() if hasattr(m, 'loglikelihood') else m.loglik_at(m.hyper_vector())[0]
        np.testing.assert_allclose(ll, float(g[k]), rtol=1e-8)


def test_oracle_predict_matches_reggie():
    _check_predict(False)


@pytest.mark.parametrize('kind', ['ei', 'pi'])
def test_oracle_acquisitions_match_reggie(kind):
    _check_acq(kind, False)


def test_oracle_sample_f_matches_reggie():
    _check_sample_f(False)


def test_oracle_loglik_matches_reggie():
    _check_loglik(False)


@pytest.mark.gpu
def test_device_predict_matches_reggie():
    _check_predict(True)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['ei', 'pi'])
def test_device_acquisitions_match_reggie(kind):
    _check_acq(kind, True)


@pytest.mark.gpu
def test_device_sample_f_matches_reggie():
    _check_sample_f(True)


@pytest.mark.gpu
def test_device_loglik_matches_reggie():
    _check_loglik(True)
