"""
Pin the GP arithmetic against `reggie` itself -- the day `reggie` is importable.

pybo holds no GP arithmetic: every `model.*` call (pybo/bayesopt.py:105-115,258,269; pybo/policies/simple.py:20-64;
pybo/recommenders.py:22-34) lands in the third-party package `reggie` (requirements.txt:8: a bare git URL, no pinned
version), which is absent from /root/reference, not installed and not fetchable here.  The oracle (oracle/gp_ref.py) is
therefore pinned to textbook known answers, scikit-learn and extended precision -- "parity against reggie: UNPINNED".

This script makes "unpinned" one command away from "pinned":

    python tests/golden/make_reggie_fixtures.py

* `import reggie` fails (today): prints why, writes nothing, exits 0.  tests/test_reggie_fixtures.py then reports its cases
  as xfail with that reason.
* `import reggie` succeeds: for small fixed problems built exactly as pybo builds its model
  (`reggie.make_gp(sn2, rho, ell, bias)`, pybo/bayesopt.py:105; `model.add_data(X, Y)`, :114) it records what pybo's call
  sites read -- and ONLY through the calls pybo itself makes:
      reggie_predict.npz   model.predict(Z) and model.predict(Z, grad=True)      (simple.py:21,35,62-70; recommenders.py:22-34)
      reggie_ei.npz        model.get_improvement(target, Z[, grad=True])         (simple.py:23-25)
      reggie_pi.npz        model.get_tail(target, Z[, grad=True])                (simple.py:37-39)
      reggie_sample_f.npz  model.sample_f(n, rng).get(Z[, grad=True]) for seeds  (simple.py:48)
      reggie_loglik.npz    model.loglikelihood() where the model offers it       (reggie.MCMC's target, bayesopt.py:115)
  next to this file.  The CPU test then holds the oracle to them, the -m gpu test the device path.

What the fixtures settle (the [RECALLED] points nobody can falsify today): whether predict returns the LATENT variance or
adds sn2; the draw order and scaling inside sample_f (W, b, the weight posterior), i.e. whether a given `rng` seed yields the
same function; the prior densities / log-likelihood constant.  A mismatch in the first is a one-line change in the oracle and
in k_acq; a mismatch in the draw order makes Thompson parity distributional instead of point-wise and the test says so.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# small fixed problems: (tag, N, d, sn2, rho, ell, bias, seed)
PROBLEMS = (('p1d', 9, 1, 1e-3, 1.3, [0.35], 0.2, 0),
            ('p2d', 40, 2, 1e-2, 0.8, [0.3, 0.6], -0.4, 1),
            ('p6d', 120, 6, 1e-4, 2.0, [0.5, 0.7, 0.4, 0.9, 0.6, 0.8], 0.0, 2))


def problem(tag, N, d, seed):
    rng = np.random.RandomState(1000 + seed)
    X = rng.rand(N, d)
    y = np.sin(3.0 * X.sum(1)) + 0.3 * X[:, 0] + 0.05 * rng.randn(N)
    Z = rng.rand(33, d)
    return X, y, Z


def main():
    try:
        import reggie
    except Exception as exc:                         # noqa: BLE001 -- any failure means "not available here"
        print('reggie is not importable in this container (%s: %s): no fixtures written; parity against reggie stays '
              'UNPINNED.  Install the version pybo was developed against and run this script again.'
              % (type(exc).__name__, exc))
        return 0
    out = {k: {} for k in ('predict', 'ei', 'pi', 'sample_f', 'loglik')}
    for tag, N, d, sn2, rho, ell, bias, seed in PROBLEMS:
        X, y, Z = problem(tag, N, d, seed)
        model = reggie.make_gp(sn2, rho, np.array(ell), bias)        # pybo/bayesopt.py:105
        model.add_data(X, y)                                         # pybo/bayesopt.py:114
        for name, a in (('X', X), ('y', y), ('Z', Z), ('hyp', np.concatenate([[sn2, rho], ell, [bias]]))):
            for k in out:
                out[k]['%s_%s' % (tag, name)] = a
        mu, s2 = model.predict(Z)[:2]
        mu2, s22, dmu, ds2 = model.predict(Z, grad=True)
        out['predict'].update({tag + '_mu': mu, tag + '_s2': s2, tag + '_dmu': dmu, tag + '_ds2': ds2,
                               tag + '_mu_obs': model.predict(X)[0]})
        target = float(model.predict(X)[0].max() + 0.01)             # simple.py:21
        ei, dei = model.get_improvement(target, Z, True)
        pi, dpi = model.get_tail(target, Z, True)
        out['ei'].update({tag + '_target': np.array(target), tag + '_val': model.get_improvement(target, Z, False),
                          tag + '_val_g': ei, tag + '_grad': dei})
        out['pi'].update({tag + '_target': np.array(target), tag + '_val': model.get_tail(target, Z, False),
                          tag + '_val_g': pi, tag + '_grad': dpi})
        for s in (0, 7):
            f = model.sample_f(100, np.random.RandomState(s))        # simple.py:48 (n = 100 features)
            out['sample_f']['%s_seed%d_val' % (tag, s)] = f.get(Z)
            try:
                out['sample_f']['%s_seed%d_grad' % (tag, s)] = f.get(Z, True)[1]
            except Exception:                                         # noqa: BLE001
                pass
        for attr in ('get_loglike', 'loglikelihood', 'get_logprior'):
            fn = getattr(model, attr, None)
            if callable(fn):
                try:
                    val = fn()
                    out['loglik']['%s_%s' % (tag, attr)] = np.array(val[0] if isinstance(val, tuple) else val, dtype=float)
                except Exception:                                     # noqa: BLE001
                    pass
    for k, d in out.items():
        d['reggie_version'] = np.array(str(getattr(reggie, '__version__', 'unknown')))
        np.savez(os.path.join(HERE, 'reggie_%s.npz' % k), **d)
    print('wrote', sorted(f for f in os.listdir(HERE) if f.startswith('reggie_')))
    return 0


if __name__ == '__main__':
    sys.exit(main())
