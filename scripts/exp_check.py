"""ulp error of gpx::exp_nonpos and of the device library exp against mpmath (50 digits)."""
import sys
import numpy as np
from mpmath import mp, mpf, exp as mexp
mp.dps = 50
raw = np.fromfile(sys.argv[1])
n = len(raw) // 3
x, mine, lib = raw[:n], raw[n:2 * n], raw[2 * n:]
assert np.isnan(mine[6]) and np.isnan(x[6])
pick = np.concatenate([np.arange(6), np.arange(7, n, 37)])
worst = {'mine': 0.0, 'lib': 0.0}
for i in pick:
    ref = mexp(mpf(float(x[i])))
    reff = float(ref)
    u = 4.9406564584124654e-324 if reff < 2.3e-308 else float(np.spacing(reff))
    for name, v in (('mine', mine[i]), ('lib', lib[i])):
        worst[name] = max(worst[name], abs(float((mpf(float(v)) - ref) / u)))
print('checked %d points; max error in ulp: exp_nonpos %.3f, library exp %.3f' % (len(pick), worst['mine'], worst['lib']))
print('specials:', x[:7], mine[:7])
assert worst['mine'] <= 1.0
