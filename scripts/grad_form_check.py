"""One-pass against two-pass predict-with-gradients (gpx option grad_form) and both triangular matvec kernels, vs the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybo_amd._lib import Engine
from oracle import gp_ref
for (N, d, kern) in [(300, 3, 'se'), (2049, 8, 'matern5'), (5000, 2, 'se'), (4100, 15, 'matern3')]:
    rng = np.random.RandomState(N)
    X = rng.rand(N, d); y = -((X - 0.5) ** 2).sum(1) + 1e-3 * rng.randn(N)
    ell = 0.3 * np.ones(d) * np.sqrt(d); rho = float(np.var(y)); bias = float(y.mean()); sn2 = 1e-3 * rho
    Z = rng.rand(7, d)
    ref = gp_ref.GPRef(sn2, rho, ell, bias, kern); ref.add_data(X, y); want = ref.predict(Z, grad=True)
    res = {}
    for name, opts in [('two-pass rows', dict(grad_form=1, grad_kernel=0)), ('two-pass rb', dict(grad_form=1, grad_kernel=1)),
                       ('one-pass', dict(grad_form=2))]:
        e = Engine(0)
        for k, v in opts.items():
            e.set_option(k, v)
        e.fit(X, y, kern, ell, rho, sn2, bias)
        res[name] = e.predict(Z, grad=True)
        one = [e.predict(Z[i:i + 1], grad=True) for i in range(len(Z))]
        same = all(np.array_equal(np.concatenate([o[j] for o in one]), res[name][j]) for j in range(4))
        print(N, d, kern, name, 'rows independent of the batch:', same)
        e.close()
    base = res['two-pass rows']
    for name in res:
        print('   ', name, 'vs oracle: max rel diff mu %.2e s2 %.2e dmu %.2e ds2 %.2e' % tuple(
            np.max(np.abs(res[name][j] - want[j]) / (np.abs(want[j]).max() + 1e-300)) for j in range(4)))
    for name in ('two-pass rb', 'one-pass'):
        print('   ', name, 'vs two-pass rows: max rel diff mu %.2e s2 %.2e dmu %.2e ds2 %.2e' % tuple(
            np.max(np.abs(res[name][j] - base[j]) / (np.abs(base[j]).max() + 1e-300)) for j in range(4)))
