"""Shared test helpers: the deterministic stub model the golden vectors were captured with
(tests/golden/make_golden.py defines the same formulas), analytic indices, synthetic problems."""
import numpy as np


class StubModel(object):
    def __init__(self, log=None):
        self.log = [] if log is None else log

    def copy(self):
        self.log.append('copy')
        return StubModel(self.log)

    @staticmethod
    def moments(X):
        X = np.array(X, ndmin=2, dtype=float)
        t = X.sum(axis=1)
        mu = np.sin(1.7 * t) + 0.3 * t
        s2 = 0.2 + 0.1 * np.cos(0.9 * t) ** 2
        dmu = np.repeat((1.7 * np.cos(1.7 * t) + 0.3)[:, None], X.shape[1], axis=1)
        ds2 = np.repeat((-0.18 * np.cos(0.9 * t) * np.sin(0.9 * t))[:, None], X.shape[1], axis=1)
        return mu, s2, dmu, ds2

    def predict(self, X, grad=False):
        self.log.append('predict:%d' % int(bool(grad)))
        m = self.moments(X)
        return m if grad else m[:2]

    def get_improvement(self, target, X, grad=False):
        self.log.append('get_improvement:%d' % int(bool(grad)))
        mu = self.moments(X)[0]
        out = mu - target
        return (out, np.ones_like(np.array(X, ndmin=2, dtype=float))) if grad else out

    def get_tail(self, target, X, grad=False):
        self.log.append('get_tail:%d' % int(bool(grad)))
        mu = self.moments(X)[0]
        out = 1.0 / (1.0 + np.exp(-(mu - target)))
        return (out, np.ones_like(np.array(X, ndmin=2, dtype=float))) if grad else out


class SmootherModel(object):
    """A deterministic, DATA-DEPENDENT stand-in for a reggie model (so that the BO loop is not trivial): a
    kernel smoother with closed-form moments and gradients.  (tests/golden/make_golden.py drove the reference's loop with the same class.)
        w_i(x) = exp(-|x - x_i|^2 / (2 ell^2))     mu = sum w_i y_i / (c + sum w_i)     s2 = 1 / (1 + sum w_i)"""

    def __init__(self, ell=0.25, c=1e-2, X=None, Y=None):
        self.ell, self.c = float(ell), float(c)
        self.X = np.empty((0, 0)) if X is None else X
        self.Y = np.empty(0) if Y is None else Y

    def copy(self):
        return SmootherModel(self.ell, self.c, self.X.copy(), self.Y.copy())

    def add_data(self, X, Y):
        X = np.array(X, ndmin=2, dtype=float)
        Y = np.array(Y, ndmin=1, dtype=float)
        self.X = X if self.X.size == 0 else np.vstack([self.X, X])
        self.Y = np.hstack([self.Y, Y])

    def predict(self, X, grad=False):
        X = np.array(X, ndmin=2, dtype=float)
        D = X[:, None, :] - self.X[None, :, :]
        W = np.exp(-0.5 * (D ** 2).sum(-1) / self.ell ** 2)
        sw = self.c + W.sum(1)
        mu = (W @ self.Y) / sw
        s2 = 1.0 / (1.0 + W.sum(1))
        if not grad:
            return mu, s2
        dW = -D / self.ell ** 2 * W[:, :, None]
        dsw = dW.sum(1)
        dmu = (np.einsum('mnd,n->md', dW, self.Y) - mu[:, None] * dsw) / sw[:, None]
        ds2 = -(s2 ** 2)[:, None] * dsw
        return mu, s2, dmu, ds2

    def _z(self, target, X, grad):
        post = self.predict(X, grad)
        mu, s2 = post[:2]
        s = np.sqrt(s2)
        z = (mu - target) / s
        cdf = 0.5 * (1.0 + np.vectorize(__import__('math').erf)(z / np.sqrt(2.0)))
        pdf = np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)
        return post, s, z, cdf, pdf

    def get_improvement(self, target, X, grad=False):
        post, s, z, cdf, pdf = self._z(target, X, grad)
        ei = (post[0] - target) * cdf + s * pdf
        if not grad:
            return ei
        return ei, cdf[:, None] * post[2] + (0.5 * pdf / s)[:, None] * post[3]

    def get_tail(self, target, X, grad=False):
        post, s, z, cdf, pdf = self._z(target, X, grad)
        if not grad:
            return cdf
        dz = post[2] / s[:, None] - (0.5 * z / post[1])[:, None] * post[3]
        return cdf, pdf[:, None] * dz


def loop_objective(x):
    x = np.ravel(x)
    return float(-np.sum((x - 0.3) ** 2) + 0.1 * np.sin(5.0 * x[0]))


class StubModel(object):
    """Deterministic stand-in for a reggie model: closed-form 'posterior' of the query points, and a log
    of the protocol calls the policy makes."""

    def __init__(self, log=None):
        self.log = [] if log is None else log

    def copy(self):
        self.log.append('copy')
        return StubModel(self.log)

    @staticmethod
    def moments(X):
        X = np.array(X, ndmin=2, dtype=float)
        t = X.sum(axis=1)
        mu = np.sin(1.7 * t) + 0.3 * t
        s2 = 0.2 + 0.1 * np.cos(0.9 * t) ** 2
        dmu = np.repeat((1.7 * np.cos(1.7 * t) + 0.3)[:, None], X.shape[1], axis=1)
        ds2 = np.repeat((-0.18 * np.cos(0.9 * t) * np.sin(0.9 * t))[:, None], X.shape[1], axis=1)
        return mu, s2, dmu, ds2

    def predict(self, X, grad=False):
        self.log.append('predict:%d' % int(bool(grad)))
        m = self.moments(X)
        return m if grad else m[:2]

    def get_improvement(self, target, X, grad=False):
        self.log.append('get_improvement:%d' % int(bool(grad)))
        mu = self.moments(X)[0]
        out = mu - target            # any deterministic function of (target, X) pins `target`
        return (out, np.ones_like(np.array(X, ndmin=2, dtype=float))) if grad else out

    def get_tail(self, target, X, grad=False):
        self.log.append('get_tail:%d' % int(bool(grad)))
        mu = self.moments(X)[0]
        out = 1.0 / (1.0 + np.exp(-(mu - target)))
        return (out, np.ones_like(np.array(X, ndmin=2, dtype=float))) if grad else out


def analytic_index(kind):
    if kind == 'bimodal2':
        c1, c2 = np.array([0.8, 0.8]), np.array([0.25, 0.3])

        def f(X, grad=False):
            X = np.array(X, ndmin=2, dtype=float)
            e1 = 2.0 * np.exp(-8.0 * ((X - c1) ** 2).sum(1))
            e2 = 1.5 * np.exp(-6.0 * ((X - c2) ** 2).sum(1))
            v = e1 + e2
            if not grad:
                return v
            g = e1[:, None] * (-16.0 * (X - c1)) + e2[:, None] * (-12.0 * (X - c2))
            return v, g
        return f, np.array([[0.0, 1.0], [0.0, 1.0]])
    if kind == 'tilted1':
        def f(X, grad=False):
            X = np.array(X, ndmin=2, dtype=float)
            x = X[:, 0]
            v = np.sin(3.0 * x) + 0.5 * x
            if not grad:
                return v
            return v, (3.0 * np.cos(3.0 * x) + 0.5)[:, None]
        return f, np.array([[0.0, 4.0]])
    if kind == 'quad5':
        c = np.array([0.3, -0.2, 0.6, 0.1, -0.5])

        def f(X, grad=False):
            X = np.array(X, ndmin=2, dtype=float)
            v = -((X - c) ** 2 * np.arange(1, 6)).sum(1)
            if not grad:
                return v
            return v, -2.0 * (X - c) * np.arange(1, 6)
        return f, np.array([[-1.0, 1.0]] * 5)
    raise KeyError(kind)


def branin(X):
    X = np.array(X, ndmin=2, dtype=float)
    a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5.0 / np.pi, 6.0, 10.0, 1.0 / (8 * np.pi)
    return a * (X[:, 1] - b * X[:, 0] ** 2 + c * X[:, 0] - r) ** 2 + s * (1 - t) * np.cos(X[:, 0]) + s


def synth_problem(N, d, seed=0, noise=1e-2):
    """Smooth synthetic regression problem in the unit box with moderate conditioning."""
    rng = np.random.RandomState(seed)
    X = rng.rand(N, d)
    y = np.sin(3.0 * X.sum(1)) + 0.5 * np.cos(5.0 * X[:, 0]) + noise * rng.randn(N)
    ell = 0.3 + 0.2 * rng.rand(d)
    return X, y, ell


def s2_tol(s2_ref, rho):
    """Stated tolerance for latent variances: |d| <= 1e-6*s2 + 1e-10*rho."""
    return 1e-6 * np.abs(s2_ref) + 1e-10 * rho


def mu_tol(mu_ref, rho):
    return 1e-6 * np.abs(mu_ref) + 1e-9 * np.sqrt(rho)


def ei_tol(mu_ref, s2_ref, target, rho):
    """Tolerance for expected improvement at a candidate = the STATED moment tolerances propagated to first
    order through EI = (mu - t) Phi(z) + s phi(z), z = (mu - t)/s  (dEI/dmu = Phi(z), dEI/ds2 = phi(z)/(2 s)),
    plus the direct 1e-6 relative term.  For z << 0, EI ~ s phi(z)/z^2, so the RELATIVE sensitivity is
    |z| dmu/s + (z^2/2) ds2/s2: at config B (s2/rho down to 1e-7, z down to -5.5 among the candidates within
    1e-9 of the best EI) moments that are right to 1e-6 leave EI right to ~1.5e-5 only -- for ANY
    implementation, the oracle included.  DESIGN.md section 6 (tolerance ladder)."""
    from scipy.special import erfc
    s = np.sqrt(s2_ref)
    z = (mu_ref - target) / s
    cdf = 0.5 * erfc(-z * 0.70710678118654752440)
    pdf = 0.39894228040143267794 * np.exp(-0.5 * z * z)
    ei = (mu_ref - target) * cdf + s * pdf
    return 1e-6 * np.abs(ei) + 1.05 * (cdf * mu_tol(mu_ref, rho) + pdf * s2_tol(s2_ref, rho) / (2.0 * s))


def ei_from_moments(mu, s2, target):
    from scipy.special import erfc
    s = np.sqrt(s2)
    z = (mu - target) / s
    return (mu - target) * 0.5 * erfc(-z * 0.70710678118654752440) + s * 0.39894228040143267794 * np.exp(-0.5 * z * z)


# ---- the reference's loop over a real GP (tests/golden/make_loop_gp.py -> loop_gp.npz) ---------------------------------
LOOP_GP_CASES = (
    # tag, bounds, objective id, GP hyper-parameters (sn2, rho, ell, bias), kernel, solve_bayesopt kwargs, niter
    ('ei_se_2d', [[0.0, 1.0], [-0.5, 1.0]], 'bumps2', (1e-4, 1.0, [0.3, 0.35], 0.0), 'se',
     dict(policy='ei', recommender='latent', solver=('lbfgs', {'nbest': 5, 'ngrid': 2000})), 8),
    ('ucb_matern_3d', [[0.0, 1.0]] * 3, 'quad3', (1e-3, 0.8, [0.4, 0.5, 0.45], -0.1), 'matern5',
     dict(policy=('ucb', {'xi': 0.3}), recommender='incumbent', solver=('lbfgs', {'nbest': 4, 'ngrid': 3000})), 7),
    ('ei_xi_se_1d', [[0.0, 4.0]], 'tilted1', (1e-4, 1.5, [0.6], 0.2), 'se',
     dict(policy=('ei', {'xi': 0.05}), recommender='latent', solver=('lbfgs', {'nbest': 3, 'ngrid': 500})), 9),
)


def loop_gp_objective(kind):
    if kind == 'bumps2':
        c1, c2 = np.array([0.75, 0.6]), np.array([0.2, -0.1])
        return lambda x: float(1.2 * np.exp(-6.0 * np.sum((np.ravel(x) - c1) ** 2)) +
                               0.9 * np.exp(-9.0 * np.sum((np.ravel(x) - c2) ** 2)))
    if kind == 'quad3':
        c = np.array([0.3, 0.7, 0.45])
        return lambda x: float(-np.sum((np.ravel(x) - c) ** 2 * np.array([1.0, 2.0, 0.5])))
    if kind == 'tilted1':
        return lambda x: float(np.sin(3.0 * np.ravel(x)[0]) + 0.5 * np.ravel(x)[0])
    raise KeyError(kind)


class SeenIndex(object):
    """An acquisition index that notes what the solver's grid stage selects: the `nbest` best grid points in ranking order
    and the best value -- whether the solver ranks f(xgrid) on the host or asks the index for its device top-k."""

    def __init__(self, f, nbest, log):
        self._f, self._nbest, self._log = f, nbest, log
        if hasattr(f, 'topk'):
            self.topk = self._topk

    def __call__(self, X, grad=False):
        res = self._f(X, grad=grad)
        if not grad and np.array(X, ndmin=2).shape[0] > 1:
            v = np.asarray(res)
            order = np.lexsort((np.arange(len(v)), -v))[:self._nbest]
            self._log.append((order.copy(), float(v[order[0]])))
        return res

    def _topk(self, Z, k):
        vals, idx = self._f.topk(Z, k)
        self._log.append((np.asarray(idx, dtype=int).copy(), float(np.asarray(vals)[0])))
        return vals, idx

    def __getattr__(self, name):
        return getattr(self._f, name)


def recording_solver(solve, log):
    """`solve` (a solve_lbfgs) behind the solver-plugin signature, its index wrapped in SeenIndex."""
    def solver(f, bounds, nbest=10, ngrid=10000, xgrid=None, rng=None):
        return solve(SeenIndex(f, nbest, log), bounds, nbest=nbest, ngrid=ngrid, xgrid=xgrid, rng=rng)
    return solver
