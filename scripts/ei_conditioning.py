"""Measured conditioning of EI at BASELINE config B (DESIGN.md tolerance ladder): device vs oracle moments and EI on
the grid sub-sample + the device's 64 best, with the first-order prediction |z| dmu/s + (z^2/2) ds2/s2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import bench
from oracle import gp_ref
from pybo_amd._lib import Engine

name = sys.argv[1] if len(sys.argv) > 1 else 'b'
M = 1 << 20
w = bench.make_workload(name, M)
e = Engine(0)
e.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
_, mx = e.mean_at_obs()
r = e.sweep('ei', mx, w['Xc'], k=64, want_moments=True)
ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
ref.add_data(w['X'], w['y'])
ei = r['acq']
live_all = np.flatnonzero(ei > 1e-9 * ei.max())
pick = np.unique(np.concatenate([live_all[:: max(1, len(live_all) // 2048)], r['top_idx']]))
mr, sr = ref.predict(w['Xc'][pick])
t = ref.mean_at_obs().max()
eir = ref.get_improvement(t, w['Xc'][pick])
s = np.sqrt(sr)
z = (mr - t) / s
dmu, ds2 = np.abs(r['mu'][pick] - mr), np.abs(r['s2'][pick] - sr)
rel = np.abs(ei[pick] - eir) / eir
pred = np.abs(z) * dmu / s + 0.5 * z * z * ds2 / sr
print('workload %s: rho %.4g sn2 %.3g; EI > 1e-9 max on %d of %d candidates; %d compared' % (name, w['rho'], w['sn2'], len(live_all), M, len(pick)))
print('z range [%.3f, %.3f]; s2/rho range [%.3g, %.3g]' % (z.min(), z.max(), (sr / w['rho']).min(), (sr / w['rho']).max()))
print('max |dmu|/sqrt(rho) %.3g   max |dmu|/s %.3g   max |ds2|/s2 %.3g   max |ds2|/rho %.3g' % ((dmu / np.sqrt(w['rho'])).max(), (dmu / s).max(), (ds2 / sr).max(), (ds2 / w['rho']).max()))
print('EI relative difference: max %.3g (at z = %.3f), median %.3g; first-order prediction max %.3g' % (rel.max(), z[np.argmax(rel)], np.median(rel), pred.max()))
top = np.searchsorted(pick, r['top_idx'])
print('among the 64 best: z in [%.3f, %.3f], EI relative difference max %.3g' % (z[top].min(), z[top].max(), rel[top].max()))
for lo, hi in ((-8, -5), (-5, -4), (-4, -3), (-3, -2), (-2, -1), (-1, 1)):
    m = (z >= lo) & (z < hi)
    if m.any():
        print('  z in [%d,%d): n %5d   max rel dEI %.3g   max (z^2/2) ds2/s2 %.3g   max |z| dmu/s %.3g' % (lo, hi, m.sum(), rel[m].max(), (0.5 * z * z * ds2 / sr)[m].max(), (np.abs(z) * dmu / s)[m].max()))
