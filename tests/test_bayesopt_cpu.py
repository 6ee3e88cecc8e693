"""The BO driver on CPU (BASELINE config[0], 'plumbing, no GPU'): pybo_amd.solve_bayesopt accepts any object
with the model protocol, so here it is driven with the ORACLE model -- which is allowed in tests only -- to
check the loop, the plugin wiring, checkpoint/resume and the reference's quirks without a GPU."""
import os
import pickle

import numpy as np
import pytest

from oracle import gp_ref
from pybo_amd import bayesopt, solve_bayesopt, policies, solvers
from pybo_amd.bayesopt import Info


def gramacy_lee(x):
    """The objective of pybo/demos/animated.py:23-37 (maximised): xopt = 0.54856343."""
    x = float(np.ravel(x)[0])
    return -(np.sin(10 * np.pi * x) / (2 * x) + (x - 1) ** 4)


def test_animated_demo_headless_finds_the_optimum():
    bounds = [[0.5, 2.5]]
    # the demo's model: make_gp(0.01, 1.9, 0.1, 0) (animated.py:47), EI with xi = 0.1 (animated.py:64)
    model = gp_ref.make_gp(0.01, 1.9, 0.1, 0.0)
    xbest, model, info = solve_bayesopt(gramacy_lee, bounds, model=model, niter=30, policy=('ei', {'xi': 0.1}),
                                        recommender='incumbent', rng=0)
    assert info.x.shape == (31, 1) and info.y.shape == (31,) and info.xbest.shape == (30, 1)
    assert model.ndata == 31
    assert abs(info.x[np.argmax(info.y)][0] - 0.54856343) < 2e-2
    assert info.y.max() > 0.8


@pytest.mark.parametrize('policy', ['ei', 'pi', 'ucb', 'thompson'])
def test_every_policy_runs_and_is_seed_deterministic(policy):
    bounds = [[0.0, 1.0], [0.0, 1.0]]
    f = lambda x: float(-np.sum((np.asarray(x) - 0.3) ** 2))        # noqa: E731
    runs = []
    for _ in range(2):
        m = gp_ref.make_gp(1e-4, 1.0, [0.3, 0.3], 0.0)
        xb, mm, info = solve_bayesopt(f, bounds, model=m, niter=6, policy=policy,
                                      solver=('lbfgs', {'ngrid': 400, 'nbest': 3}), rng=11)
        runs.append(info)
    np.testing.assert_array_equal(runs[0].x, runs[1].x)
    np.testing.assert_array_equal(runs[0].xbest, runs[1].xbest)
    assert runs[0].y.max() > -0.1


def test_checkpoint_resume(tmp_path):
    bounds = [[0.0, 1.0]]
    f = lambda x: float(np.sin(6 * x[0]))                            # noqa: E731
    log = str(tmp_path / 'bo.pkl')
    m = gp_ref.make_gp(1e-4, 1.0, [0.2], 0.0)
    _, _, full = solve_bayesopt(f, bounds, model=m.copy(), niter=6, rng=5,
                                solver=('lbfgs', {'ngrid': 200}))
    calls = []

    def f_counted(x):
        calls.append(1)
        return f(x)
    _, _, part = solve_bayesopt(f_counted, bounds, model=m.copy(), niter=3, rng=5, log=log,
                                solver=('lbfgs', {'ngrid': 200}))
    assert os.path.exists(log) and len(part.xbest) == 3 and len(calls) == 4
    model_, info_ = pickle.load(open(log, 'rb'))
    assert model_.ndata == 4 and len(info_.x) == 4
    # resume: the loop restarts at len(info.xbest) (bayesopt.py:262) and the stored model is used
    _, mm, resumed = solve_bayesopt(f_counted, bounds, model=None, niter=6, rng=5, log=log,
                                    solver=('lbfgs', {'ngrid': 200}))
    assert len(calls) == 7 and len(resumed.xbest) == 6 and mm.ndata == 7
    np.testing.assert_array_equal(resumed.x[:4], part.x)


def test_given_model_starts_from_the_middle_and_is_not_mutated():
    bounds = [[0.0, 2.0], [1.0, 3.0]]
    m = gp_ref.make_gp(1e-3, 1.0, [0.5, 0.5], 0.0)
    _, mm, info = solve_bayesopt(lambda x: -float(np.sum(x)), bounds, model=m, niter=2, rng=0,
                                 solver=('lbfgs', {'ngrid': 100}))
    np.testing.assert_array_equal(info.x[0], [1.0, 2.0])
    assert m.ndata == 0 and mm.ndata == 3


def test_default_model_needs_the_device_and_fails_loudly(gpu_available):
    if gpu_available:
        pytest.skip('GPU present: covered by the gpu suite')
    from pybo_amd._lib import GpxError
    with pytest.raises(GpxError):
        solve_bayesopt(lambda x: 0.0, [[0.0, 1.0]], niter=1, rng=0)     # init_model builds the HIP GP


def test_info_is_a_namedtuple_of_arrays():
    m = gp_ref.make_gp(1e-3, 1.0, [0.5], 0.0)
    xb, mm, info = solve_bayesopt(lambda x: float(x[0]), [[0.0, 1.0]], model=m, niter=1, rng=0,
                                  solver=('lbfgs', {'ngrid': 50}))
    assert isinstance(info, Info) and info._fields == ('x', 'y', 'xbest')
    assert all(isinstance(a, np.ndarray) for a in info)


def test_lockstep_refinement_equals_sequential_and_batches_the_calls():
    """select='best': the nbest L-BFGS-B runs in lock-step give exactly the sequential answer, with one
    batched index call per round instead of one call per instance and iteration."""
    from pybo_amd import solvers
    from helpers import analytic_index
    f, bounds = analytic_index('bimodal2')
    calls = {'n': 0, 'rows': 0}

    def counted(X, grad=False):
        if grad:
            calls['n'] += 1
            X2 = np.atleast_2d(X)
            calls['rows'] += 1 if (len(X2) == 2 and np.array_equal(X2[0], X2[1])) else len(X2)   # (a lone row travels twice)
        return f(X, grad)

    grid = np.random.RandomState(3).rand(400, 2)
    xa, fa = solvers.solve_lbfgs(counted, bounds, nbest=6, xgrid=grid, select='best', batched=False)
    seq_calls = calls['n']
    calls.update(n=0, rows=0)
    xb, fb = solvers.solve_lbfgs(counted, bounds, nbest=6, xgrid=grid, select='best', batched=True)
    np.testing.assert_array_equal(xa, xb)
    assert fa == fb
    assert calls['rows'] == seq_calls            # same evaluations ...
    assert calls['n'] < seq_calls / 2            # ... in far fewer calls
    assert np.max(np.abs(f(xb[None], grad=True)[1])) < 1e-4      # a stationary point of the index


def test_lockstep_refinement_surfaces_index_errors():
    from pybo_amd import solvers

    def broken(X, grad=False):
        X = np.atleast_2d(X)
        if grad:
            raise ValueError('index failed')
        return -np.sum((X - 0.5) ** 2, axis=1)

    with pytest.raises(ValueError, match='index failed'):
        solvers.solve_lbfgs(broken, [[0, 1], [0, 1]], nbest=3, xgrid=np.random.RandomState(0).rand(20, 2),
                            select='best')



def test_trace_rows_behave_like_the_reference_lists_and_convert_in_one_copy():
    """The trace columns are lists of points (pybo/bayesopt.py:271 appends to them) that also hand numpy one array."""
    from pybo_amd.bayesopt import _Rows
    import pickle
    rows = _Rows()
    assert len(rows) == 0 and not rows and np.array(rows).size == 0
    pts = np.random.RandomState(0).rand(40, 3)
    for p in pts[:20]:
        rows.append(p)
    rows.extend(pts[20:])
    assert len(rows) == 40 and rows[-1].shape == (3,) and np.array_equal(rows[17], pts[17])
    np.testing.assert_array_equal(np.array(rows, ndmin=2, dtype=float), pts)
    np.testing.assert_array_equal(np.array(list(rows)), pts)
    back = pickle.loads(pickle.dumps(rows))
    assert isinstance(back, _Rows) and np.array_equal(np.array(back), pts)
    with pytest.raises(ValueError):
        rows.append(np.zeros(4))


def test_checkpoint_stores_the_trace_as_arrays_and_loads_lists(tmp_path):
    from pybo_amd.bayesopt import safe_dump, safe_load, Info
    log = str(tmp_path / 'c.pkl')
    X = np.random.RandomState(1).rand(7, 2)
    safe_dump({'m': 1}, Info(list(X), list(X[:, 0]), list(X[:5])), log)
    model, info = safe_load(log)
    assert model == {'m': 1} and isinstance(info.x, list) and len(info.x) == 7 and len(info.xbest) == 5
    np.testing.assert_array_equal(np.array(info.x), X)
    assert info.y == list(X[:, 0])


def test_solver_shard_option_resolves_through_the_plugin_api():
    from pybo_amd.bayesopt import get_component
    from pybo_amd import solvers
    from helpers import analytic_index
    solver = get_component(('lbfgs', {'shard': False, 'ngrid': 500}), solvers, np.random.RandomState(0), lstrip='solve_')
    f, bounds = analytic_index('bimodal2')
    x, fx = solver(f, bounds)
    assert abs(fx - 2.0) < 0.1
    with pytest.raises(ValueError):
        solvers.solve_lbfgs(f, bounds, shard='maybe')


def test_bench_self_launch_builds_the_driver_shaped_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run."""
    import subprocess
    import sys
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, 'call', lambda cmd, **kw: seen.update(cmd=cmd, kw=kw) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2'])
    assert bench.self_launch(4) == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert '--master-addr' in cmd and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and cmd[-5].endswith('bench.py')


def test_trace_rows_mirror_follows_every_list_mutation():
    """_Rows mirrored only append / extend; np.array(trace.x) -- what the policies and recommenders
    read (pybo/policies/simple.py:20, pybo/recommenders.py:22) -- must follow the other list mutators as well."""
    import pickle
    from pybo_amd.bayesopt import _Rows
    r = _Rows([[1., 2.], [3., 4.], [5., 6.]])
    r.pop(0)
    np.testing.assert_array_equal(np.array(r), [[3, 4], [5, 6]])
    r.insert(0, [9., 9.])
    r[1] = [0., 0.]
    np.testing.assert_array_equal(np.array(r), [[9, 9], [0, 0], [5, 6]])
    r += [[7., 7.]]
    assert isinstance(r, _Rows) and np.array(r).shape == (4, 2)
    del r[0]
    r.reverse()
    np.testing.assert_array_equal(np.array(r), [[7, 7], [5, 6], [0, 0]])
    r.sort(key=lambda v: v[0])
    np.testing.assert_array_equal(np.array(r)[:, 0], [0, 5, 7])
    np.testing.assert_array_equal(np.array(pickle.loads(pickle.dumps(r))), np.array(r))
    r.remove(r[0])
    r.clear()
    assert len(r) == 0 and np.array(r).size == 0
    r.append([1., 1.])
    assert np.array(r).shape == (1, 2)
