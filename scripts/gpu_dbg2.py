import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np
from pybo_amd import models, solve_bayesopt, _lib
from helpers import branin
orig_check = _lib.Engine._check
def chk(self, rc):
    if rc != 0:
        print("ENGINE", self, "h=", self._h, "rc", rc, "N,d", self.N, self.d)
    return orig_check(self, rc)
_lib.Engine._check = chk
orig_fit = _lib.Engine.fit
def fit(self, X, y, kernel, ell, rho, sn2, bias, stage=3):
    print("fit", X.shape, y.shape, kernel, ell, rho, sn2, bias, "h=", self._h)
    return orig_fit(self, X, y, kernel, ell, rho, sn2, bias, stage)
_lib.Engine.fit = fit
bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
f = lambda x: float(-branin(x)[0] / 10.0)
ell = 0.25 * (bounds[:, 1] - bounds[:, 0])
X0 = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * np.random.RandomState(0).rand(12, 2)
y0 = np.array([f(x) for x in X0])
m = models.make_gp(1e-4, float(np.var(y0)), ell, float(np.mean(y0)))
m.add_data(X0, y0)
grid = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * np.random.RandomState(5).rand(4000, 2)
try:
    xb, mm, info = solve_bayesopt(f, bounds, model=m, niter=4, policy='ei', solver=('lbfgs', {'xgrid': grid, 'nbest': 5}), recommender='incumbent', rng=3, verbose=True)
    print(info)
except Exception:
    traceback.print_exc()
