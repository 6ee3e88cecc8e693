"""
Generate tests/golden/*.npz by IMPORTING the reference's own modules from /root/reference in this
container (the reference cannot travel to the GPU box; only these vectors do).

What can be imported:
  pybo/policies/simple.py   natively under Python 3 (package __init__ bypassed by a namespace shim)
  pybo/utils.py             natively
  pybo/inits/methods.py     via in-memory lib2to3 (xrange); sobol.py likewise (print statements)
  pybo/solvers/lbfgs.py     natively once pybo.inits resolves
  pybo/recommenders.py      natively once pybo.solvers resolves
What cannot: pybo/bayesopt.py (cPickle + reggie) and all GP arithmetic (reggie is absent) -- parity of
the GP moments is UNPINNED against the reference and pinned by tests/test_oracle.py instead.

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
    print('wrote', sorted(os.listdir(HERE)))


if __name__ == '__main__':
    main()
