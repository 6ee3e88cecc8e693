"""
oracle/gp_ref.py -- CPU fp64 restatement of the GP posterior + acquisition path.

*** TEST INFRASTRUCTURE ONLY ***  Nothing under pybo_amd/ imports this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may.  The product path is the HIP library and
fails loudly without it.

*** PARITY UNPINNED (against reggie) ***  pybo delegates every line of GP arithmetic to the third-party
package `reggie` (requirements.txt:8 `git+https://github.com/mwhoffman/reggie.git`, no tag / commit;
setup.py:32 unversioned).  It is absent from /root/reference, not installed and not fetchable, and the
reference ships no tests or golden vectors.  This file therefore restates the
published algorithm reggie implements -- exact GP regression, Rasmussen & Williams (2006) Alg. 2.1, with
the parametrisation of the constructor call `reggie.make_gp(sn2, rho, ell, bias)`
(/root/reference/pybo/bayesopt.py:98-105) -- and is anchored on pybo's own call sites:

    model.add_data(X, Y)                  pybo/bayesopt.py:114,258,269
    model.predict(X[, grad])   -> (mu, s2[, dmu, ds2])      pybo/policies/simple.py:21,35,64-70
                                                             pybo/recommenders.py:22-24,34
    model.get_improvement(target, X, grad)                   pybo/policies/simple.py:25
    model.get_tail(target, X, grad)                          pybo/policies/simple.py:39
    model.sample_f(n, rng).get(X, grad)                      pybo/policies/simple.py:48
    model.copy()                                             pybo/policies/simple.py:20,34,57

What pins it instead: analytic known answers, scikit-learn's independent GP and a long-double
re-evaluation (tests/test_oracle.py), plus golden vectors captured from the importable reference
modules for everything that IS pybo's own code (tests/golden/).
"""
import numpy as np
import scipy.linalg as sla
import scipy.special as sps

SE_ARD, MATERN52, MATERN32, MATERN12 = 0, 1, 2, 3
KERNEL_IDS = {'se': SE_ARD, 'matern5': MATERN52, 'matern3': MATERN32, 'matern1': MATERN12}

_SQRT5 = 2.23606797749978969641
_SQRT3 = 1.73205080756887729353
S2_FLOOR = 1e-100  # latent variances are clipped here before sqrt (round-off guard)


def sqdist(A, B):
    """Squared euclidean distance between rows of A (n,d) and B (m,d) by direct differences
    (never the |a|^2+|b|^2-2ab expansion: that loses ~1e-16*|a|^2 absolute, fatal next to data)."""
    r2 = np.zeros((A.shape[0], B.shape[0]))
    for k in range(A.shape[1]):
        df = A[:, k][:, None] - B[:, k][None, :]
        r2 += df * df
    return r2


def kern_from_r2(kid, r2, rho):
    """Covariance as a function of the squared length-scaled distance."""
    if kid == SE_ARD:
        return rho * np.exp(-0.5 * r2)
    if kid == MATERN52:
        s = _SQRT5 * np.sqrt(r2)
        return rho * (1.0 + s + (5.0 / 3.0) * r2) * np.exp(-s)
    if kid == MATERN32:
        s = _SQRT3 * np.sqrt(r2)
        return rho * (1.0 + s) * np.exp(-s)
    if kid == MATERN12:
        return rho * np.exp(-np.sqrt(r2))
    raise ValueError('unknown kernel id')


def dkern_dr2(kid, r2, rho):
    """d k / d r2 (Matern-1/2 is not differentiable at r2 = 0: see below)."""
    if kid == SE_ARD:
        return -0.5 * rho * np.exp(-0.5 * r2)
    if kid == MATERN52:
        s = _SQRT5 * np.sqrt(r2)
        return -(5.0 / 6.0) * rho * (1.0 + s) * np.exp(-s)
    if kid == MATERN32:
        s = _SQRT3 * np.sqrt(r2)
        return -1.5 * rho * np.exp(-s)
    if kid == MATERN12:
        # kink at r = 0 (a candidate on top of an observation): the one-sided slopes are +-1, the symmetric
        # value 0 is used there so that gradients at training inputs stay finite
        r = np.sqrt(r2)
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.where(r > 0.0, -0.5 * rho * np.exp(-r) / r, 0.0)
    raise ValueError('unknown kernel id')


def kernel(kid, X, Z, ell, rho):
    """k(X, Z): scale both by 1/ell first, then difference (the order the device uses)."""
    ell = np.asarray(ell, dtype=float)
    return kern_from_r2(kid, sqdist(X / ell, Z / ell), rho)


def norm_cdf(z):
    return 0.5 * sps.erfc(-z * 0.70710678118654752440)


def norm_pdf(z):
    return 0.39894228040143267794 * np.exp(-0.5 * z * z)


class RFFSample(object):
    """One posterior function sample in random-Fourier-feature form (Rahimi & Recht 2007):
    f(x) = bias + sum_j theta_j cos(w_j.x + b_j), theta already carrying sqrt(2 rho / n).
    `.get(X, grad)` is the index pybo's Thompson policy returns (pybo/policies/simple.py:48)."""

    def __init__(self, W, b, theta, bias):
        self.W, self.b, self.theta, self.bias = W, b, theta, bias

    def get(self, X, grad=False):
        X = np.array(X, ndmin=2, dtype=float)
        Z = X @ self.W.T + self.b
        f = self.bias + np.cos(Z) @ self.theta
        if not grad:
            return f
        g = -(np.sin(Z) * self.theta) @ self.W
        return f, g

    __call__ = get


def rff_draw_spectral(kid, n, d, ell, rng):
    """Spectral frequencies W (n,d) and phases b (n,) for the stationary kernels; the draw ORDER is
    part of the definition: randn(n,d), [chisquare(2 nu, n) for Matern], rand(n)."""
    ell = np.asarray(ell, dtype=float)
    W = rng.randn(n, d)
    if kid != SE_ARD:
        nu = {MATERN52: 2.5, MATERN32: 1.5, MATERN12: 0.5}[kid]
        u = rng.chisquare(2.0 * nu, size=n)
        W = W * np.sqrt(2.0 * nu / u)[:, None]
    W = W / ell
    b = rng.rand(n) * 2.0 * np.pi
    return W, b


def rff_posterior_theta(A, v, n, rho, sn2, z):
    """Weight posterior draw given the raw feature Gram A = C^T C, v = C^T (y - bias) with
    C = cos(X W^T + b):  Phi = s C, s = sqrt(2 rho / n);
        theta_w ~ N( (Phi^T Phi + sn2 I)^-1 Phi^T r,  sn2 (Phi^T Phi + sn2 I)^-1 ),   z ~ N(0, I_n)
    returns theta = s * theta_w (so that f = bias + C(x) theta)."""
    s = np.sqrt(2.0 * rho / n)
    Am = (s * s) * A + sn2 * np.eye(n)
    L = np.linalg.cholesky(Am)
    mean = sla.cho_solve((L, True), s * v)
    noise = np.sqrt(sn2) * sla.solve_triangular(L, z, lower=True, trans='T')
    return s * (mean + noise)


class _ParamRef(object):
    def __init__(self):
        self.prior = None

    def set_prior(self, kind, *args):
        self.prior = (kind,) + tuple(np.array(a, dtype=float) for a in args)


class GPRef(object):
    """Exact GP regression with a constant mean and gaussian noise; fp64, numpy/scipy."""

    def __init__(self, sn2, rho, ell, bias=0.0, kernel='se'):
        self.sn2 = float(sn2)
        self.rho = float(rho)
        self.ell = np.array(ell, dtype=float, ndmin=1)
        self.bias = float(bias)
        self.kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        self.X = None
        self.Y = None
        self.L = None
        self.a = None
        self.params = {k: _ParamRef() for k in ('like.sn2', 'kern.rho', 'kern.ell', 'mean.bias')}

    # -- hyper-parameters / evidence (what a hyper-parameter sampler needs) ----------------------------
    def hyper_vector(self):
        return np.concatenate([[np.log(self.sn2), np.log(self.rho)], np.log(self.ell), [self.bias]])

    def set_hyper_vector(self, theta):
        theta = np.asarray(theta, dtype=float)
        d = len(self.ell)
        self.sn2, self.rho = float(np.exp(theta[0])), float(np.exp(theta[1]))
        self.ell = np.exp(theta[2:2 + d])
        self.bias = float(theta[2 + d])
        if self.X is not None:
            self._fit()

    def loglikelihood(self):
        """R&W eq. 2.30:  -1/2 a.a - sum log L_ii - N/2 log 2 pi."""
        n = len(self.X)
        return float(-0.5 * self.a @ self.a - np.sum(np.log(np.diag(self.L))) - 0.5 * n * np.log(2 * np.pi))

    # -- protocol ------------------------------------------------------------------------------
    def copy(self):
        new = GPRef(self.sn2, self.rho, self.ell.copy(), self.bias, self.kid)
        for k, p in self.params.items():
            new.params[k].prior = p.prior
        if self.X is not None:
            new.X, new.Y = self.X.copy(), self.Y.copy()
            new.L, new.a = self.L, self.a
        return new

    @property
    def ndata(self):
        return 0 if self.X is None else len(self.X)

    def add_data(self, X, Y):
        d = len(self.ell)
        X = np.array(X, dtype=float)
        X = X.reshape(-1, d) if X.ndim != 2 else X
        Y = np.array(Y, dtype=float).reshape(-1)
        if self.X is None:
            self.X, self.Y = X.copy(), Y.copy()
        else:
            self.X = np.vstack([self.X, X])
            self.Y = np.hstack([self.Y, Y])
        self._fit()

    def _fit(self):
        K = kernel(self.kid, self.X, self.X, self.ell, self.rho)
        K[np.diag_indices_from(K)] += self.sn2
        self.L = np.linalg.cholesky(K)                      # raises LinAlgError if not PD
        self.a = sla.solve_triangular(self.L, self.Y - self.bias, lower=True)

    def gram(self):
        K = kernel(self.kid, self.X, self.X, self.ell, self.rho)
        K[np.diag_indices_from(K)] += self.sn2
        return K

    def alpha(self):
        return sla.solve_triangular(self.L, self.a, lower=True, trans='T')

    def mean_at_obs(self):
        """Latent posterior mean at the observed inputs, closed form  y - sn2*alpha
        (== predict(X_obs)[0] up to round-off; what EI/PI use for their target,
        pybo/policies/simple.py:21,35)."""
        return self.Y - self.sn2 * self.alpha()

    def predict(self, X, grad=False, chunk=8192):
        X = np.array(X, ndmin=2, dtype=float)
        M, d = X.shape
        mu = np.empty(M)
        s2 = np.empty(M)
        if self.X is None:
            mu[:] = self.bias
            s2[:] = self.rho
            if grad:
                return mu, s2, np.zeros((M, d)), np.zeros((M, d))
            return mu, s2
        if grad:
            dmu = np.empty((M, d))
            ds2 = np.empty((M, d))
            alpha = self.alpha()
        Xs = self.X / self.ell
        for m0 in range(0, M, chunk):
            Z = X[m0:m0 + chunk]
            Zs = Z / self.ell
            r2 = sqdist(Xs, Zs)
            Ks = kern_from_r2(self.kid, r2, self.rho)                     # (N, m)
            V = sla.solve_triangular(self.L, Ks, lower=True)
            mu[m0:m0 + chunk] = self.bias + V.T @ self.a
            s2[m0:m0 + chunk] = np.maximum(self.rho - np.sum(V * V, axis=0), S2_FLOOR)
            if grad:
                G = dkern_dr2(self.kid, r2, self.rho)                     # dk/dr2 (N, m)
                Wm = sla.solve_triangular(self.L, V, lower=True, trans='T')   # K^-1 k*
                for j in range(d):
                    # d r2 / d z_j = 2 (z_j - x_j) / ell_j^2
                    dK = G * (2.0 * (Zs[:, j][None, :] - Xs[:, j][:, None]) / self.ell[j])
                    dmu[m0:m0 + chunk, j] = dK.T @ alpha
                    ds2[m0:m0 + chunk, j] = -2.0 * np.sum(dK * Wm, axis=0)
        if grad:
            return mu, s2, dmu, ds2
        return mu, s2

    def get_improvement(self, target, X, grad=False):
        """Expected improvement over `target`:  (mu-t) Phi(z) + s phi(z),  z = (mu-t)/s."""
        post = self.predict(X, grad=grad)
        mu, s2 = post[:2]
        s = np.sqrt(s2)
        dlt = mu - target
        z = dlt / s
        cdf, pdf = norm_cdf(z), norm_pdf(z)
        ei = dlt * cdf + s * pdf
        if not grad:
            return ei
        dmu, ds2 = post[2:]
        # d EI = Phi(z) dmu + phi(z) ds,  ds = ds2 / (2 s)
        dei = cdf[:, None] * dmu + (0.5 * pdf / s)[:, None] * ds2
        return ei, dei

    def get_tail(self, target, X, grad=False):
        """Probability of improvement  Phi((mu - t)/s)."""
        post = self.predict(X, grad=grad)
        mu, s2 = post[:2]
        s = np.sqrt(s2)
        z = (mu - target) / s
        pi = norm_cdf(z)
        if not grad:
            return pi
        dmu, ds2 = post[2:]
        dz = dmu / s[:, None] - (0.5 * z / s2)[:, None] * ds2
        return pi, norm_pdf(z)[:, None] * dz

    def sample_f(self, n, rng=None):
        rng = rng if isinstance(rng, np.random.RandomState) else np.random.RandomState(rng)
        d = len(self.ell)
        W, b = rff_draw_spectral(self.kid, n, d, self.ell, rng)
        z = rng.randn(n)
        if self.X is None:
            theta = np.sqrt(2.0 * self.rho / n) * z
            return RFFSample(W, b, theta, self.bias)
        C = np.cos(self.X @ W.T + b)
        A = C.T @ C
        v = C.T @ (self.Y - self.bias)
        theta = rff_posterior_theta(A, v, n, self.rho, self.sn2, z)
        return RFFSample(W, b, theta, self.bias)


def make_gp(sn2, rho, ell, bias=0.0, kernel='se'):
    """Same signature as reggie.make_gp (pybo/bayesopt.py:105), plus the kernel family."""
    return GPRef(sn2, rho, ell, bias, kernel)


# -- acquisition sweep + top-k as the reference solver does it on a grid --------------------------
def topk_desc(vals, k):
    """k best indices: value descending, ties by ascending index, NaN ranks last.
    (pybo uses np.argsort(finit)[::-1] whose tie order is unspecified -- SURVEY F14; this is the
    deterministic rule the build fixes.)"""
    v = np.where(np.isnan(vals), -np.inf, vals)
    order = np.lexsort((np.arange(len(v)), -v))
    return order[:k]


# ---------------------------------------------------------------------------------------------------
# candidate grids (checker for gpx_grid_create; include/gpx.h)
# ---------------------------------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11),
    vectorised over `counter` (n, 4) uint32 with one `key` (2,) uint32.  Known-answer vectors of the paper's
    reference implementation are checked in tests/test_oracle.py."""
    c = np.array(counter, dtype=np.uint64, ndmin=2).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & mask
        n3 = p0 & mask
        c = np.stack([n0, n1, n2, n3], axis=1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def grid_uniform(seed, bounds, M):
    """Host restatement of GPX_GRID_UNIFORM: counter = element-pair index, key = seed; each output gives the
    two 53-bit uniforms of elements 2c, 2c+1 of the row-major (M, d) array; x = lo + u * (hi - lo)."""
    b = np.array(bounds, dtype=float, ndmin=2)
    d = len(b)
    total = M * d
    pairs = (total + 1) // 2
    ctr = np.zeros((pairs, 4), dtype=np.uint64)
    idx = np.arange(pairs, dtype=np.uint64)
    ctr[:, 0] = idx & np.uint64(0xFFFFFFFF)
    ctr[:, 1] = idx >> np.uint64(32)
    out = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).astype(np.uint64)
    u = np.empty(2 * pairs)
    u[0::2] = ((out[:, 0] << np.uint64(32) | out[:, 1]) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    u[1::2] = ((out[:, 2] << np.uint64(32) | out[:, 3]) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    u = u[:total].reshape(M, d)
    return b[:, 0] + u * (b[:, 1] - b[:, 0])
