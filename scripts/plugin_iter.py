"""Where does a WARM iteration of pybo_amd.solve_bayesopt go at the north-star size?  (VERDICT round 2, next #6:
plugin-level warm iteration <= 1.3x the engine's warm step, no k_sweep_trmm launch in it.)
Runs the loop body (_bo_step) through the public plug-ins on bench.make_workload(...) with a DeviceGrid:
    python scripts/plugin_iter.py [--workload ns] [--iters 6] [--profile]      (cProfile of the warm iterations)
    rocprofv3 --kernel-trace --stats -d gpurun_out/plugin_trace -- python scripts/plugin_iter.py --iters 3
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='ns')
    ap.add_argument('--candidates', type=int, default=1 << 20)
    ap.add_argument('--iters', type=int, default=6)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--recommender', default='latent')
    args = ap.parse_args()
    import bench
    from pybo_amd import models, inits, policies, solvers, recommenders
    from pybo_amd.bayesopt import _bo_step, Info, _Rows, get_component
    w = bench.make_workload(args.workload, args.candidates)
    bounds = np.stack([w['lo'], w['hi']], axis=1)
    model = models.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], kernel=w['kernel'])
    model.add_data(w['X'], w['y'])
    grid = inits.DeviceGrid('sobol', bounds, args.candidates)
    trace = Info(_Rows(w['X']), list(w['y']), _Rows(w['X']))
    rng = np.random.RandomState(0)
    policy = get_component('ei' if w['acq'] == 'ei' else 'ucb', policies, rng)
    solver = get_component(('lbfgs', {'xgrid': grid}), solvers, rng, lstrip='solve_')
    recommender = get_component(args.recommender, recommenders, rng, lstrip='best_')
    noise = np.random.RandomState(1)
    objective = lambda x: float(w['f'](np.array(x, ndmin=2))[0] + 1e-3 * noise.randn())    # noqa: E731
    t0 = time.perf_counter()
    _bo_step(model, trace, objective, bounds, policy, solver, recommender)
    print('cold iteration: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
    eng = model._engine()
    prof = cProfile.Profile() if args.profile else None
    for i in range(args.iters):
        eng.timers(reset=True)
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        _bo_step(model, trace, objective, bounds, policy, solver, recommender)
        if prof:
            prof.disable()
        dt = (time.perf_counter() - t0) * 1e3
        tm = eng.timers()
        print('warm iteration %d: %.2f ms   device stages: append %.2f rank1 %.2f acq_topk %.2f | sweep launches %d'
              % (i, dt, tm['append'], tm['rank1'], tm['acq_topk'], tm['sweep_trmm_launches']))
    if prof:
        pstats.Stats(prof).sort_stats('cumulative').print_stats(35)


if __name__ == '__main__':
    main()
