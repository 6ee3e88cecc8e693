"""GPU: the triangular inverse's leading part riding behind the task-graph factorisation (option trtri_ahead) -- same
kernels on another stream, so every result must be bit-identical to the serial order, in every way the library gets there:
eagerly, lazily after the previous model's inverse was used, discarded by the next fit, and next to a failing factor."""
import numpy as np
import pytest

from helpers import synth_problem

pytestmark = pytest.mark.gpu


def _engine(**opts):
    from pybo_amd._lib import Engine
    e = Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    return e


def _problem(N, d=5, seed=3):
    X, y, ell = synth_problem(N, d, seed=seed)
    return X, y, ell, 1.3, 1e-3, 0.2


@pytest.mark.parametrize('N', [1024, 1100, 3072, 3200, 4224, 8192])      # 8 (the default minimum), 9, 24, 25, 33 and 64 blocks
def test_eager_inverse_ahead_is_bit_identical(N):
    X, y, ell, rho, sn2, bias = _problem(N)
    Xq = np.random.RandomState(1).rand(300, X.shape[1])
    res = {}
    for ahead in (0, 1):
        e = _engine(trtri_ahead=ahead, eager_inverse=1)
        for _ in range(2):
            e.fit(X, y, 'se', ell, rho, sn2, bias)
        tm = e.timers(reset=True)
        assert tm['trtri_ahead'] == (2 if ahead else 0)
        assert tm['chol_fallbacks'] == 0
        res[ahead] = e.predict(Xq) + tuple(e.get_vectors())
        if N <= 4224:
            res[ahead] += (e.get_matrix('T'),)
        e.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_lazy_inverse_speculates_only_after_the_previous_inverse_was_used():
    N = 4096
    X, y, ell, rho, sn2, bias = _problem(N)
    Xq = np.random.RandomState(2).rand(64, X.shape[1])
    ref = _engine(trtri_ahead=0)
    ref.fit(X, y, 'se', ell, rho, sn2, bias)
    want = ref.predict(Xq)
    e = _engine()                                              # trtri_ahead is on by default
    e.fit(X, y, 'se', ell, rho, sn2, bias)                     # first model of the handle: nothing to go by
    got = e.predict(Xq)                                        # forms the inverse
    assert e.timers(reset=True)['trtri_ahead'] == 0
    e.fit(X, y, 'se', ell, rho, sn2, bias)                     # the previous inverse was used: this one runs ahead
    got2 = e.predict(Xq)
    assert e.timers(reset=True)['trtri_ahead'] == 1
    # a fit whose inverse is never asked for, then another fit: the side stream's work is waited for and dropped
    e.fit(0.5 * X, y, 'se', ell, rho, sn2, bias)              # runs ahead (the previous inverse was used) ... for nothing
    e.fit(X, y, 'se', ell, rho, sn2, bias)                     # previous inverse NOT used: no speculation here
    got3 = e.predict(Xq)
    assert e.timers(reset=True)['trtri_ahead'] == 0
    for g in (got, got2, got3):
        assert np.array_equal(g[0], want[0]) and np.array_equal(g[1], want[1])
    e.close(); ref.close()


def test_ahead_next_to_a_factor_that_fails():
    N = 3200
    X, y, ell, rho, sn2, bias = _problem(N)
    Xbad = X.copy()
    Xbad[N - 7] = Xbad[N - 9]                                  # two identical points late in the matrix, no noise: singular
    e = _engine(eager_inverse=1)
    with pytest.raises(np.linalg.LinAlgError):
        e.fit(Xbad, y, 'se', ell, rho, 0.0, bias)
    e.fit(X, y, 'se', ell, rho, sn2, bias)                     # the handle is intact and the next model is right
    ref = _engine(trtri_ahead=0, eager_inverse=1)
    ref.fit(X, y, 'se', ell, rho, sn2, bias)
    Xq = np.random.RandomState(4).rand(50, X.shape[1])
    a, b = e.predict(Xq), ref.predict(Xq)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    e.close(); ref.close()


def test_batched_likelihoods_between_a_fit_and_its_inverse_leave_the_riding_inversion_alone():
    """gpx_loglik_batch at this size runs task-graph factorisations of its own on the handle (lent buffers): the handle's model
    may still have the leading part of its inversion on the side stream -- that state must survive."""
    N = 3000
    X, y, ell, rho, sn2, bias = _problem(N)
    Xq = np.random.RandomState(5).rand(40, X.shape[1])
    ref = _engine(trtri_ahead=0)
    ref.fit(X, y, 'se', ell, rho, sn2, bias)
    want = ref.predict(Xq)
    e = _engine()
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    e.predict(Xq)                                              # the inverse was used: the next fit speculates
    e.fit(X, y, 'se', ell, rho, sn2, bias)
    hyp = np.array([[sn2, rho] + list(ell) + [bias], [2 * sn2, 0.9 * rho] + list(1.1 * ell) + [bias]])
    ll = e.loglik_batch(hyp)
    got = e.predict(Xq)
    tm = e.timers(reset=True)
    assert tm['trtri_ahead'] == 1 and tm['chol_fallbacks'] == 0
    assert np.all(np.isfinite(ll)) and abs(ll[0] - ref.loglik()) <= 1e-8 * abs(ll[0])
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    e.close(); ref.close()
