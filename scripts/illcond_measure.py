"""The ill-conditioned regime, measured: errors of the HIP path (explicit triangular inverse)
AND of oracle/gp_ref.py (substitution) against the long-double truth of tests/golden/illcond_ld.npz, for the inputs of
config B (N = 2048) and of the north-star workload (N = 8192) at sn2 = 1e-6 * rho and the literal sn2 = 1e-6 of
pybo/bayesopt.py:98.  Prints the table DESIGN.md section 6 quotes.   python scripts/illcond_measure.py [--opt name=value]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--cases', default='b,ns')
    args = ap.parse_args()
    import bench
    from oracle import gp_ref
    from pybo_amd._lib import Engine
    fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'illcond_ld.npz'))
    print('%-4s %-4s %-9s | %-27s | %-27s | %s' % ('case', 'sn2', 'sn2/rho', 'max |d mu| / sqrt(rho)  dev / oracle',
                                                    'max |d s2| / rho   dev / oracle', 'max |d s2| / s2   dev / oracle   (min s2/rho)'))
    for name in args.cases.split(','):
        w = bench.make_workload(name, 1 << 12)
        Z = fx['Z_' + name]
        for label in ('rel', 'lit'):
            sn2 = float(fx['sn2_%s_%s' % (name, label)])
            mt, st = fx['mu_%s_%s' % (name, label)], fx['s2_%s_%s' % (name, label)]
            e = Engine(0)
            for kv in args.opt:
                k, v = kv.split('=')
                e.set_option(k, int(v))
            e.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], sn2, w['bias'])
            md, sd = e.predict(Z)
            sw = e.sweep('mean', None, Z, k=0, want_all=False, want_moments=True)
            assert np.array_equal(sw['mu'], md) and np.array_equal(sw['s2'], sd)
            ref = gp_ref.make_gp(sn2, w['rho'], w['ell'], w['bias'], w['kernel'])
            ref.add_data(w['X'], w['y'])
            mo, so = ref.predict(Z)
            rho = w['rho']
            print('%-4s %-4s %-9.2e | %12.2e %12.2e   | %12.2e %12.2e   | %12.2e %12.2e   (%.1e)'
                  % (name, label, sn2 / rho, np.max(np.abs(md - mt)) / np.sqrt(rho), np.max(np.abs(mo - mt)) / np.sqrt(rho),
                     np.max(np.abs(sd - st)) / rho, np.max(np.abs(so - st)) / rho,
                     np.max(np.abs(sd - st) / st), np.max(np.abs(so - st) / st), st.min() / rho))
            tol_ok = np.all(np.abs(sd - st) <= 1e-6 * st + 1e-10 * rho) and np.all(np.abs(md - mt) <= 1e-6 * np.abs(mt) + 1e-9 * np.sqrt(rho))
            tol_ok_o = np.all(np.abs(so - st) <= 1e-6 * st + 1e-10 * rho) and np.all(np.abs(mo - mt) <= 1e-6 * np.abs(mt) + 1e-9 * np.sqrt(rho))
            print('          stated tolerances vs the truth: device %s, oracle %s' % (tol_ok, tol_ok_o))
            e.close()


if __name__ == '__main__':
    main()
