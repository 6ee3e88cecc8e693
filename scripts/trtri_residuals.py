"""Left / right residuals of the triangular inverse and its time, per association of the recursion (option trtri_left):
python scripts/trtri_residuals.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pybo_amd._lib import Engine
fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'illcond_ld.npz'))
for name in ('b', 'ns'):
    w = bench.make_workload(name, 1 << 12)
    for label in ('rel', 'lit'):
        sn2 = float(fx['sn2_%s_%s' % (name, label)])
        for left in (0, 1):
            e = Engine(0)
            e.set_option('trtri_left', left)
            ts = []
            for rep in range(3):
                e.timers(reset=True)
                e.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], sn2, w['bias'])
                e.sync()
                ts.append(e.timers(reset=True)['trtri'])
            L = e.get_matrix('L').astype(np.longdouble) if w['N'] <= 2048 else None
            msg = ''
            if L is not None:
                T = e.get_matrix('T').astype(np.longdouble)
                I = np.eye(len(L), dtype=np.longdouble)
                msg = '  max|T L - I| = %.2e   max|L T - I| = %.2e' % (float(np.abs(T @ L - I).max()), float(np.abs(L @ T - I).max()))
            print('%s %s sn2/rho=%.1e trtri_left=%d: trtri %.3f ms%s' % (name, label, sn2 / w['rho'], left, min(ts), msg), flush=True)
            e.close()
