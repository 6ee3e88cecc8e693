"""Where a WARM plug-in iteration spends its time: wall-clock per phase of pybo_amd.bayesopt._bo_step at the north-star size
(policy, solver = grid stage + refinement, announce, add_data, recommender, checkpoint), and the device calls inside."""
import sys, os, time, tempfile, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import pybo_amd
from pybo_amd import models, inits, bayesopt, _lib
from pybo_amd.bayesopt import safe_dump, Info

if os.environ.get('PHASES_TORCH'):
    import torch
    if os.environ['PHASES_TORCH'] == '2':
        _t = torch.zeros(1 << 20, device='cuda:0'); torch.cuda.synchronize()
w = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else 'ns', 1 << 20)
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, d, M = w['N'], w['d'], w['M']
if os.environ.get('PHASES_PRE'):
    # what bench.py's process has done before its plugin_step: an engine on torch's stream, a cold step, warm steps
    import torch
    dev = torch.device('cuda', 0)
    dX = torch.from_numpy(w['X']).to(dev); dy = torch.from_numpy(w['y']).to(dev); dXc = torch.from_numpy(w['Xc']).to(dev)
    stream = torch.cuda.current_stream(dev)
    eng0 = _lib.Engine(0, stream.cuda_stream if os.environ['PHASES_PRE'] != '2' else None)
    eng0.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
    _, mx = eng0.mean_at_obs()
    eng0.set_option('sweep_cache', 1)
    eng0.sweep_dev('ei', mx, dXc.data_ptr(), M, 10)
    eng0.set_option('sweep_cache', 0)
    if os.environ['PHASES_PRE'] == '3':
        eng0.close(); del eng0, dX, dy, dXc; torch.cuda.empty_cache()
bounds = np.stack([w['lo'], w['hi']], axis=1)
acc = collections.defaultdict(list)

def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name].append(time.perf_counter() - t0)
    return wrapper

# phases of the loop
for name in ('safe_dump',):
    setattr(bayesopt, name, timed(name, getattr(bayesopt, name)))
E = _lib.Engine
for meth in ('predict', 'predict_mean', 'sweep_update', 'append', 'append_begin', 'mean_at_obs', 'var_at_obs', 'sweep_dev', 'sweep', 'fit'):
    if hasattr(E, meth):
        setattr(E, meth, timed('engine.' + meth, getattr(E, meth)))
G = models.GP
for meth in ('add_data', 'anticipate', 'copy', 'acq_topk', 'get_improvement', 'predict_mean', 'predict'):
    setattr(G, meth, timed('GP.' + meth, getattr(G, meth)))

gp = models.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], kernel=w['kernel'])
gp._X, gp._Y = np.array(w['X']), np.array(w['y'])
grid = inits.DeviceGrid('sobol', bounds, M)
rng = np.random.RandomState(11)
stamps = []
def objective(x):
    stamps.append(time.perf_counter())
    return float(w['f'](np.array(x, ndmin=2))[0] + 1e-3 * rng.randn())

from pybo_amd import policies, solvers, recommenders
pol = timed('policy', policies.EI)
sol = timed('solver', lambda index, b: solvers.solve_lbfgs(index, b, xgrid=grid, nbest=10))
rec = timed('recommender', recommenders.best_latent)
with tempfile.TemporaryDirectory() as tmp:
    log = os.path.join(tmp, 'bo.pkl')
    safe_dump(gp, Info(list(w['X']), list(w['y']), list(w['X'])), log)
    del gp
    pybo_amd.solve_bayesopt(objective, bounds, niter=N + nsteps, policy=pol, solver=sol, recommender=rec, log=log)
spans = np.diff(stamps) * 1e3
print('warm iterations (ms):', ' '.join('%.2f' % s for s in spans), ' mean of the last %d: %.2f' % (len(spans) - 2, spans[2:].mean()))
skip = 3      # cold iteration + first warm ones
for name in sorted(acc, key=lambda n: -sum(acc[n][skip:])):
    v = np.array(acc[name]) * 1e3
    per_iter = len(v) / float(nsteps + 1)
    tail = v[int(skip * per_iter):]
    print('%-24s calls/iter %5.1f   mean %8.3f ms   per iteration %8.3f ms' % (name, per_iter, tail.mean() if len(tail) else 0, tail.sum() / max(1, nsteps + 1 - skip)))
print('per iteration (ms), iterations 0..4:')
for name in ('policy', 'solver', 'GP.acq_topk', 'GP.anticipate', 'GP.add_data', 'engine.append', 'recommender', 'safe_dump'):
    print('  %-16s %s' % (name, ' '.join('%8.3f' % (1e3 * v) for v in acc[name][:5])))
