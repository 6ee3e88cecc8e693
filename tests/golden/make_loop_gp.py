"""
Generate tests/golden/loop_gp.npz: the REFERENCE's solve_bayesopt (pybo/bayesopt.py:234-287, loaded in memory through
lib2to3 exactly as make_golden.py does), the REFERENCE's solve_lbfgs (pybo/solvers/lbfgs.py:17-68) and the REFERENCE's
policies EI / UCB (pybo/policies/simple.py) and recommender best_latent / best_incumbent (pybo/recommenders.py) driving a
REAL Gaussian process: oracle.gp_ref.GPRef with fixed hyper-parameters -- the model the -m gpu tests pin the device against.
This closes the triangle directly: reference loop + reference solver + reference policies over the oracle GP, replayed on the
GPU through pybo_amd.solve_bayesopt + the device GP (tests/test_gpu_parity.py) and on the CPU through pybo_amd.solve_bayesopt
+ the same oracle GP (tests/test_golden_host.py).

Per case the fixture holds the trace (info.x, info.y, info.xbest, the final recommendation) and, per iteration, what the
solver's grid stage selected: the indices of the `nbest` best grid points in the reference's own order
(np.argsort(finit)[::-1][:nbest], lbfgs.py:51) and the grid's best value.

Run:  python tests/golden/make_loop_gp.py      (needs /root/reference; writes next to this file)
The reference's sources are read where they lie; nothing of them is written anywhere.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))      # tests/helpers.py: the cases, shared with the tests that replay them

def main():
    import make_golden as mg
    from helpers import LOOP_GP_CASES, loop_gp_objective, recording_solver
    from oracle import gp_ref
    methods, lb, pol, rec = mg._shim()
    ref_bo, _ = mg.load_reference_bayesopt(pol, rec)
    out = {}
    for tag, bounds, okind, hyp, kern, kw, niter in LOOP_GP_CASES:
        grid_log = []
        kw = dict(kw)
        name, skw = kw.pop('solver')
        assert name == 'lbfgs'
        model = gp_ref.make_gp(hyp[0], hyp[1], np.array(hyp[2]), hyp[3], kern)
        # the REFERENCE's solver, unchanged; the index it is handed is wrapped so that the grid stage is seen (lbfgs.py:50-51)
        xbest, final, info = ref_bo.solve_bayesopt(loop_gp_objective(okind), bounds, model=model, niter=niter, rng=11,
                                                   solver=(recording_solver(lb.solve_lbfgs, grid_log), skw), **kw)
        assert model.ndata == 0 and final.ndata == niter + 1 and len(grid_log) >= niter
        # the recommender best_latent runs the solver too (recommenders.py:22-34): two grid stages per iteration then
        per = len(grid_log) // niter
        assert per * niter == len(grid_log)
        out[tag + '_x'], out[tag + '_y'], out[tag + '_xbest'] = info.x, info.y, info.xbest
        out[tag + '_final'] = np.array(xbest)
        out[tag + '_grid_top'] = np.array([g[0] for g in grid_log[::per]])          # the policy's grid stage
        out[tag + '_grid_best'] = np.array([g[1] for g in grid_log[::per]])
        out[tag + '_stages_per_iter'] = np.array(per)
        print(tag, 'x[-1] =', info.x[-1], 'y max =', info.y.max(), 'grid stages per iteration =', per)
    np.savez(os.path.join(HERE, 'loop_gp.npz'), **out)
    print('wrote', os.path.join(HERE, 'loop_gp.npz'))


if __name__ == '__main__':
    main()
