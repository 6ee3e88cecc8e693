"""
Generate tests/golden/*.npz by IMPORTING the reference's own modules from /root/reference in this
container (the reference cannot travel to the GPU box; only these vectors do).

What can be imported:
  pybo/policies/simple.py   natively under Python 3 (package __init__ bypassed by a namespace shim)
  pybo/utils.py             natively
  pybo/inits/methods.py     via in-memory lib2to3 (xrange); sobol.py likewise (print statements)
  pybo/solvers/lbfgs.py     natively once pybo.inits resolves
  pybo/recommenders.py      natively once pybo.solvers resolves
  pybo/bayesopt.py          via in-memory lib2to3 (cPickle, xrange) once a module named `reggie` exists: a
                            RECORDING stand-in is installed for it (make_gp / MCMC that log their arguments) --
                            so the reference's own solve_bayesopt loop, get_component and init_model RUN here,
                            over stub models, and their outputs are captured (round 2)
What cannot: any GP arithmetic (reggie is absent) -- parity of the GP moments is UNPINNED against the
reference and pinned by tests/test_oracle.py instead.

Run:  python tests/golden/make_golden.py      (needs /root/reference; writes next to this file)
"""
import importlib
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference/pybo'
HERE = os.path.dirname(os.path.abspath(__file__))


def _shim():
    pkg = types.ModuleType('pybo')
    pkg.__path__ = [REF]
    sys.modules['pybo'] = pkg

    def load_2to3(modname, path, package):
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
        src = open(path).read()
        if not src.endswith('\n'):
            src += '\n'
        code = str(tool.refactor_string(src, path))
        mod = types.ModuleType(modname)
        mod.__file__ = path
        mod.__package__ = package
        sys.modules[modname] = mod
        exec(compile(code, path, 'exec'), mod.__dict__)
        return mod

    importlib.import_module('pybo.utils')
    inits = types.ModuleType('pybo.inits')
    inits.__path__ = [os.path.join(REF, 'inits')]
    sys.modules['pybo.inits'] = inits
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        load_2to3('pybo.inits.sobol', os.path.join(REF, 'inits', 'sobol.py'), 'pybo.inits')
        methods = load_2to3('pybo.inits.methods', os.path.join(REF, 'inits', 'methods.py'), 'pybo.inits')
    for name in methods.__all__:
        setattr(inits, name, getattr(methods, name))
    inits.__all__ = list(methods.__all__)
    sys.modules['pybo'].inits = inits
    solvers = types.ModuleType('pybo.solvers')
    solvers.__path__ = [os.path.join(REF, 'solvers')]
    sys.modules['pybo.solvers'] = solvers
    lb = importlib.import_module('pybo.solvers.lbfgs')
    solvers.solve_lbfgs = lb.solve_lbfgs
    sys.modules['pybo'].solvers = solvers
    pol = importlib.import_module('pybo.policies.simple')
    rec = importlib.import_module('pybo.recommenders')
    return methods, lb, pol, rec


class RecordingParam(object):
    def __init__(self):
        self.prior = None

    def set_prior(self, kind, *args):
        self.prior = (kind,) + tuple(np.array(a, dtype=float) for a in args)


class RecordingGP(object):
    """What the stand-in `reggie.make_gp(sn2, rho, ell, bias)` returns: it only records what pybo does to it."""

    def __init__(self, sn2, rho, ell, bias):
        self.args = (float(sn2), float(rho), np.array(ell, dtype=float), float(bias))
        self.params = {k: RecordingParam() for k in ('like.sn2', 'kern.rho', 'kern.ell', 'mean.bias')}
        self.data = None

    def add_data(self, X, Y):
        self.data = (np.array(X, dtype=float), np.array(Y, dtype=float))


def load_reference_bayesopt(pol, rec):
    """Load /root/reference/pybo/bayesopt.py itself (in memory, through lib2to3) against a recording `reggie`."""
    from lib2to3 import refactor
    made = {}

    def make_gp(sn2, rho, ell, bias):
        made['gp'] = RecordingGP(sn2, rho, ell, bias)
        return made['gp']

    def MCMC(model, n=None, burn=None, rng=None):
        made['mcmc'] = dict(n=n, burn=burn, rng_is_state=isinstance(rng, np.random.RandomState))
        return model

    fake = types.ModuleType('reggie')
    fake.make_gp, fake.MCMC = make_gp, MCMC
    sys.modules['reggie'] = fake
    pkg = sys.modules['pybo']
    policies = types.ModuleType('pybo.policies')
    for name in pol.__all__:
        setattr(policies, name, getattr(pol, name))
    policies.__all__ = list(pol.__all__)
    sys.modules['pybo.policies'] = policies
    pkg.policies, pkg.recommenders = policies, rec
    # pybo/solvers/__init__.py imports the nlopt-backed DIRECT solver too; the shim package exposes what loaded
    pkg.solvers.__all__ = ['solve_lbfgs']
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    path = os.path.join(REF, 'bayesopt.py')
    code = str(tool.refactor_string(open(path).read(), path))
    mod = types.ModuleType('pybo.bayesopt')
    mod.__file__, mod.__package__ = path, 'pybo'
    sys.modules['pybo.bayesopt'] = mod
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        exec(compile(code, path, 'exec'), mod.__dict__)
    return mod, made


class SmootherModel(object):
    """A deterministic, DATA-DEPENDENT stand-in for a reggie model (so that the BO loop is not trivial): a
    kernel smoother with closed-form moments and gradients.  The same class lives in tests/helpers.py.
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
    """Tie-free analytic indices (value, gradient) for the solver fixtures."""
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


def main():
    methods, lb, pol, rec = _shim()
    out = {}

    # G3: initial designs for fixed seeds ---------------------------------------------------------
    b2 = [[0.0, 1.0], [2.0, 4.0]]
    b3 = [[-5.0, 10.0], [0.0, 15.0], [1.0, 3.0]]
    g3 = {}
    g3['middle_b2'] = methods.init_middle(np.array(b2))
    g3['middle_b3'] = methods.init_middle(np.array(b3))
    for name, bounds in (('b2', b2), ('b3', b3)):
        for seed in (0, 7):
            for n in (None, 5, 64):
                key = '%s_s%d_n%s' % (name, seed, n)
                g3['uniform_' + key] = methods.init_uniform(np.array(bounds), n, seed)
                g3['latin_' + key] = methods.init_latin(np.array(bounds), n, seed)
    g3['sobol_ref_b3_s0_n8'] = methods.init_sobol(np.array(b3), 8, 0)   # documentation only
    np.savez(os.path.join(HERE, 'inits.npz'), **g3)

    # G1: policy closures against the stub ---------------------------------------------------------
    g1 = {}
    rng = np.random.RandomState(3)
    Xobs = [rng.rand(3) for _ in range(7)]          # the list-of-points form solve_bayesopt passes
    Xq = rng.rand(11, 3)
    g1['Xobs'] = np.array(Xobs)
    g1['Xq'] = Xq
    for pname, kw in (('EI', {}), ('EI', {'xi': 0.25}), ('PI', {}), ('PI', {'xi': 0.3}),
                      ('UCB', {}), ('UCB', {'delta': 0.05, 'xi': 0.7})):
        stub = StubModel()
        index = getattr(pol, pname)(stub, None, Xobs, **kw)
        tag = pname + ''.join('_%s%g' % kv for kv in sorted(kw.items()))
        v = index(Xq)
        v2, g = index(Xq, grad=True)
        g1[tag + '_val'] = v
        g1[tag + '_val_g'] = v2
        g1[tag + '_grad'] = g
        g1[tag + '_log'] = np.array('|'.join(stub.log))
    # UCB beta known answer (F7): N = 3 observations, xi = 0.2, delta = 0.1
    stub = StubModel()
    X3 = [np.zeros(2)] * 3
    idx = pol.UCB(stub, None, X3)
    mu, s2 = StubModel.moments(Xq[:, :2])[:2]
    g1['UCB_beta_N3'] = np.array(((idx(Xq[:, :2]) - mu) ** 2 / s2)[0])
    np.savez(os.path.join(HERE, 'policies.npz'), **g1)

    # G2: solver selections on analytic indices ------------------------------------------------------
    g2 = {}
    for kind in ('bimodal2', 'tilted1', 'quad5'):
        f, bounds = analytic_index(kind)
        d = len(bounds)
        for nbest in (1, 3, 10):
            grid = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * np.random.RandomState(11).rand(200, d)
            x, fx = lb.solve_lbfgs(f, bounds, nbest=nbest, xgrid=grid)
            g2['%s_nb%d_x' % (kind, nbest)] = x
            g2['%s_nb%d_f' % (kind, nbest)] = np.array(fx)
        # default random grid drawn from the rng argument
        x, fx = lb.solve_lbfgs(f, bounds, nbest=4, ngrid=500, rng=5)
        g2['%s_rng5_x' % kind] = x
        g2['%s_rng5_f' % kind] = np.array(fx)
    np.savez(os.path.join(HERE, 'solver.npz'), **g2)

    # recommenders on the stub ------------------------------------------------------------------------
    g4 = {}
    stub = StubModel()
    bounds3 = np.array([[0.0, 1.0]] * 3)
    g4['incumbent'] = rec.best_incumbent(stub, bounds3, np.array(Xobs))
    g4['latent'] = rec.best_latent(stub, bounds3, np.array(Xobs))
    g4['Xobs'] = np.array(Xobs)
    np.savez(os.path.join(HERE, 'recommenders.npz'), **g4)

    # G5-G7: the reference's OWN bayesopt.py, loaded against a recording `reggie` -----------------------------
    ref_bo, made = load_reference_bayesopt(pol, rec)
    import pybo

    # G4 (was hand-derived in round 1): get_component through the real function
    g5 = {}
    rng = np.random.RandomState(0)
    table = [('ei', 'policies', ''), ('pi', 'policies', ''), ('ucb', 'policies', ''), ('thompson', 'policies', ''),
             ('lbfgs', 'solvers', 'solve_'), ('latent', 'recommenders', 'best_'),
             ('incumbent', 'recommenders', 'best_'), (('lbfgs', {'nbest': 3}), 'solvers', 'solve_'),
             (('ucb', {'xi': 0.5}), 'policies', ''), ('nope', 'policies', ''), (('ei', {'bogus': 1}), 'policies', ''),
             (('thompson', {'rng': 1}), 'policies', ''), ('EI', 'policies', ''), (('ei', 1, 2), 'policies', ''),
             ('solve_lbfgs', 'solvers', 'solve_'), ('best_latent', 'recommenders', 'best_')]
    rows = []
    for spec, modname, strip in table:
        module = getattr(pybo, modname)
        try:
            got = ref_bo.get_component(spec, module, rng, lstrip=strip)
            func = getattr(got, 'func', got)
            kw = dict(getattr(got, 'keywords', {}) or {})
            has_rng = kw.pop('rng', None) is rng
            rows.append('%r|%s|%s|ok|%s|%s|%d' % (spec, modname, strip, func.__name__, sorted(kw.items()), has_rng))
        except Exception as ex:                     # noqa: BLE001 -- the exception TYPE is the datum
            rows.append('%r|%s|%s|%s' % (spec, modname, strip, type(ex).__name__))
    g5['table'] = np.array(rows)
    np.savez(os.path.join(HERE, 'components.npz'), **g5)

    # G5: the whole solve_bayesopt loop (pybo/bayesopt.py:234-287) over the data-dependent stub
    g6 = {}
    for tag, bounds, kw in (
            ('ei_latent_2d', [[0.0, 1.0], [-0.5, 1.0]], dict(policy='ei', recommender='latent')),
            ('pi_incumbent_2d', [[0.0, 1.0], [-0.5, 1.0]], dict(policy=('pi', {'xi': 0.02}), recommender='incumbent')),
            ('ucb_latent_3d', [[0.0, 1.0]] * 3, dict(policy='ucb', recommender='latent',
                                                       solver=('lbfgs', {'nbest': 4, 'ngrid': 400})))):
        kw.setdefault('solver', ('lbfgs', {'ngrid': 300}))
        xbest, model, info = ref_bo.solve_bayesopt(loop_objective, bounds, model=SmootherModel(), niter=6, rng=4, **kw)
        g6[tag + '_x'], g6[tag + '_y'], g6[tag + '_xbest'] = info.x, info.y, info.xbest
        g6[tag + '_final'] = np.array(xbest)
        g6[tag + '_ndata'] = np.array(len(model.Y))
    np.savez(os.path.join(HERE, 'loop.npz'), **g6)

    # G6: init_model (pybo/bayesopt.py:60-118): design, heuristic hyper-parameters, priors, MCMC arguments
    g7 = {}
    for tag, bounds, kw in (('b2', [[0.0, 1.0], [-0.5, 1.0]], {}), ('b3_n5', [[-5.0, 10.0], [0.0, 15.0], [1.0, 3.0]], {'ninit': 5}),
                            ('flat', [[0.0, 1.0]], {})):
        f = (lambda x: 0.5) if tag == 'flat' else loop_objective
        ref_bo.init_model(f, bounds, rng=9, **kw)
        gp = made['gp']
        g7[tag + '_hypers'] = np.concatenate([[gp.args[0], gp.args[1]], gp.args[2], [gp.args[3]]])
        g7[tag + '_X'], g7[tag + '_Y'] = gp.data
        for name, prm in gp.params.items():
            g7['%s_prior_%s_kind' % (tag, name)] = np.array(prm.prior[0])
            for j, a in enumerate(prm.prior[1:]):
                g7['%s_prior_%s_%d' % (tag, name, j)] = np.atleast_1d(a)
        g7[tag + '_mcmc'] = np.array([made['mcmc']['n'], made['mcmc']['burn'], int(made['mcmc']['rng_is_state'])])
    np.savez(os.path.join(HERE, 'init_model.npz'), **g7)
    print('wrote', sorted(os.listdir(HERE)))


if __name__ == '__main__':
    main()
