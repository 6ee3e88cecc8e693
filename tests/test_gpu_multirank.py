"""The N > 1 path with the REAL HIP engine, provable on a one-GPU box: 2 and 4 gloo ranks share device 0, each
with its own process, HIP context and gpx handle, and run what bench.py's step() runs on config B's inputs
(N = 2048, d = 2, SE-ARD, EI, 2^20 Sobol candidates): replicated fit, contiguous candidate shard, local top-k,
ONE all-gather, deterministic merge.  Asserted:

  * every rank's Cholesky factor L and triangular inverse T are BITWISE equal to the single-rank ones
    ("every rank fits redundantly, deterministic" -- the claim the whole zero-communication
    fit rests on);
  * the merged top-k (values and global indices) is bit-identical on every rank and to the single-rank sweep
    of the whole grid;
  * the same through the plugin layer: solve_lbfgs(ShardedIndex(policies.EI(GP, ...))) picks the same seed;
  * Thompson / batch-BO: draws sharded by draw index, one all-gather of (value, index) per draw, equals the
    single-rank sweep of all draws.

The reference has no distributed code (its only hint is the comment pybo/solvers/lbfgs.py:60); the layout is
SURVEY.md 8(e).  RCCL itself needs one GPU per rank, so the transport here is gloo; the RCCL binding of the
same exchange (gpx_topk_allgather) is exercised with a 1-rank communicator below.
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 10
DRAWS = 8


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _thompson_draw(w, s):
    """Spectral draw s of bench.py's Thompson step (same order as GP.sample_f / the oracle)."""
    rng = np.random.RandomState(100 + s)
    W = rng.randn(100, w['d']) / w['ell']
    b = rng.rand(100) * 2 * np.pi
    z = rng.randn(100)
    return W, b, z


def _rank_work(rank, world, w):
    """What one rank does in a step; returns plain numpy results.  Runs inside an initialised process group
    (or stand-alone with world == 1)."""
    from pybo_amd import dist as pdist, models, policies, solvers
    from pybo_amd._lib import Engine
    out = {}
    eng = Engine(0)
    eng.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
    out['L'] = _digest(eng.get_matrix('L'))
    out['T'] = _digest(eng.get_matrix('T'))
    _, target = eng.mean_at_obs()
    out['target'] = target
    lo, hi = pdist.shard_bounds(w['M'], rank, world)
    r = eng.sweep('ei', target, w['Xc'][lo:hi], k=K, want_all=False)
    out['topk'] = pdist.gather_topk(r['top_val'], r['top_idx'] + lo, K)
    # Thompson: draws s = rank (mod world), every rank sweeps ALL candidates for its draws
    mine = [s for s in range(DRAWS) if s % world == rank]
    Ws, bs, zs = zip(*[_thompson_draw(w, s) for s in mine])
    As, vs = eng.rff_gram_batch(np.array(Ws), np.array(bs))
    sc = np.sqrt(2.0 * w['rho'] / 100)
    ths = []
    for A, v, z in zip(As, vs, zs):
        Lw = np.linalg.cholesky(sc * sc * A + w['sn2'] * np.eye(100))
        ths.append(sc * (np.linalg.solve(Lw.T, np.linalg.solve(Lw, sc * v)) + np.sqrt(w['sn2']) * np.linalg.solve(Lw.T, z)))
    rr = eng.rff_sweep(np.array(Ws), np.array(bs), np.array(ths), w['bias'], w['Xc'][:1 << 16], k=1, want_all=False)
    tv, ti = pdist.gather_pairs(rr['top_val'][:, 0], rr['top_idx'][:, 0])
    order = np.argsort(np.concatenate([[s for s in range(DRAWS) if s % world == r2] for r2 in range(world)]),
                       kind='stable')
    out['thompson'] = (tv[order], ti[order])            # back to draw order
    eng.close()
    # the same through pybo's plugin API
    gp = models.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], kernel=w['kernel'])
    gp.add_data(w['X'], w['y'])
    bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
    index = pdist.ShardedIndex(policies.EI(gp, bounds, w['X']))
    out['plugin_topk'] = index.topk(w['Xc'], K)
    out['plugin_x'] = solvers.solve_lbfgs(index, bounds, nbest=K, xgrid=w['Xc'])
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    import bench
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        w = bench.make_workload('b', 1 << 20)
        q.put((rank, _rank_work(rank, world, w)))
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope='module')
def single():
    sys.path.insert(0, ROOT)
    import bench
    w = bench.make_workload('b', 1 << 20)
    return _rank_work(0, 1, w)


@pytest.mark.parametrize('world', [2, 4, 8])
def test_ranks_sharing_one_gpu_reproduce_the_single_rank_step(world, single):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == list(range(world))
    for rank, res in got.items():
        assert res['L'] == single['L'], 'rank %d: Cholesky factor differs bitwise from the single-rank fit' % rank
        assert res['T'] == single['T'], 'rank %d: triangular inverse differs bitwise' % rank
        assert res['target'] == single['target']
        for a, b in zip(res['topk'], single['topk']):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(res['plugin_topk'], single['plugin_topk']):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(res['plugin_x'][0], single['plugin_x'][0])
        assert res['plugin_x'][1] == single['plugin_x'][1]
        for a, b in zip(res['thompson'], single['thompson']):
            np.testing.assert_array_equal(a, b)
    # engine level and plugin level agree with each other too
    np.testing.assert_array_equal(single['topk'][1], single['plugin_topk'][1])


# ---- configs D and E at world = 8: Thompson draws sharded by draw index, full 2^20 grid, ONE all-gather ----------------
def _draw_work(rank, world, name, S):
    """bench.py's Thompson step for this rank's draws (s = rank mod world) over ALL 2^20 candidates: replicated fit at the
    workload's N (N = 16384 / 8192: the persistent task-graph factorisation, here with `world` copies of it sharing one
    GPU), weight posteriors on the device, sweep, one (value, index) pair per draw, ONE all-gather."""
    sys.path.insert(0, ROOT)
    import bench
    from pybo_amd import dist as pdist
    from pybo_amd._lib import Engine
    w = bench.make_workload(name, 1 << 20)
    eng = Engine(0)
    eng.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'], stage=2)
    mine = [s for s in range(S) if s % world == rank]
    tri = [bench.thompson_draw(w, s) for s in mine]
    Wa, ba, za = (np.array([t[i] for t in tri]) for i in range(3))
    th = eng.rff_posterior(Wa, ba, za, np.sqrt(2.0 * w['rho'] / 100))
    rr = eng.rff_sweep(Wa, ba, th, w['bias'], w['Xc'], k=1, want_all=False)
    tv, ti = pdist.gather_pairs(rr['top_val'][:, 0], rr['top_idx'][:, 0])
    order = np.argsort(np.concatenate([[s for s in range(S) if s % world == r2] for r2 in range(world)]), kind='stable')
    fb = eng.timers()['chol_fallbacks']
    eng.close()
    return tv[order], ti[order], fb


def _draw_worker(rank, world, port, q, name, S):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _draw_work(rank, world, name, S)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name,S', [('e', 8), ('d', 64)])
def test_eight_ranks_shard_the_thompson_draws_of_configs_d_and_e(name, S):
    """BASELINE configs[3] / [4] as the 8-GPU node will run them (bench.py, draws mod 8), on one GPU: every rank returns the
    winners of ALL draws after the all-gather, identical on every rank and to one rank doing all the draws."""
    world = 8
    want_v, want_i, _ = _draw_work(0, 1, name, S)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_draw_worker, args=(r, world, port, q, name, S)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=1500) for _ in range(world))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank in range(world):
        np.testing.assert_array_equal(got[rank][1], want_i)
        np.testing.assert_array_equal(got[rank][0], want_v)
    assert len(want_i) == S


# ---- the reference's DEFAULT model (an ensemble of 10 GPs, pybo/bayesopt.py:115) over 8 ranks -------------------------------
def _ensemble_work(rank, world, n=10, M=1 << 18):
    """bench.py --ensemble's step for one rank on config B's inputs: every rank fits all n members (bitwise-equal factors, no
    communication), sweeps ITS contiguous slice of the grid with ONE gpx_ensemble_sweep call, one all-gather of the top-k.
    (Members are NOT the sharded unit: 10 members over 8 ranks would leave six ranks with half the work of the other two.)
    Also through the plugin layer: ShardedIndex over policies.EI of an ensemble model."""
    sys.path.insert(0, ROOT)
    import bench
    from pybo_amd import dist as pdist
    from pybo_amd._lib import Engine
    w = bench.make_workload('b', M)
    hyp = bench.ensemble_hypers(w, n)
    engines = []
    for sn2, rho, ell, bias in hyp:
        e = Engine(0)
        e.fit(w['X'], w['y'], w['kernel'], ell, rho, sn2, bias)
        engines.append(e)
    lo, hi = pdist.shard_bounds(M, rank, world)
    target = float(np.max(w['y']))
    r = Engine.ensemble_sweep(engines, 'ei', target, w['Xc'][lo:hi], k=K, want_all=False)
    tv, ti = pdist.gather_topk(r['top_val'], np.where(r['top_idx'] >= 0, r['top_idx'] + lo, r['top_idx']), K)
    r2 = Engine.ensemble_sweep(engines, 'ucb', 2.5, w['Xc'][lo:hi], k=K, want_all=False)
    uv, ui = pdist.gather_topk(r2['top_val'], np.where(r2['top_idx'] >= 0, r2['top_idx'] + lo, r2['top_idx']), K)
    digest = [_digest(e.get_matrix('L')) for e in engines[:3]]
    for e in engines:
        e.close()
    return tv, ti, uv, ui, digest


def _ensemble_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _ensemble_work(rank, world)))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_shard_the_candidates_of_a_ten_member_ensemble():
    """the default model's sweep over 8 ranks (sharing the one GPU): the merged top-k of the
    member-averaged EI and of the mixture UCB is bit-identical on every rank and to ONE rank sweeping the whole grid, and the
    members' factors are bitwise equal across ranks."""
    world = 8
    want = _ensemble_work(0, 1)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ensemble_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank in range(world):
        for a, b in zip(got[rank][:4], want[:4]):
            np.testing.assert_array_equal(a, b)
        assert got[rank][4] == want[4]
    assert want[1][0] >= 0 and len(set(want[1])) == K


def test_rccl_exchange_behind_the_c_abi_with_a_one_rank_communicator():
    """gpx_comm_unique_id / gpx_comm_init / gpx_topk_allgather on the real RCCL with nranks = 1: the pairs are
    read from the device buffers the sweep left behind, offset, gathered, merged on the device."""
    from pybo_amd._lib import Engine, Comm, GpxError
    from helpers import synth_problem
    X, y, ell = synth_problem(300, 3, seed=4)
    e = Engine(0)
    e.fit(X, y, 'se', ell, 1.2, 1e-3, 0.1)
    Z = np.random.RandomState(3).rand(5000, 3)
    c = Comm(e, 0, 1, Comm.unique_id())
    with pytest.raises(GpxError):                     # nothing on the device yet
        c.topk_allgather(7, 0, 7)
    r = e.sweep('ei', 0.4, Z, k=7, want_all=False)
    tv, ti = c.topk_allgather(7, 1000, 7)
    np.testing.assert_array_equal(tv, r['top_val'])
    np.testing.assert_array_equal(ti, r['top_idx'] + 1000)
    with pytest.raises(GpxError):                     # n must be what the last sweep produced
        c.topk_allgather(5, 0, 5)
    # no-merge mode (batch-BO): S draws x top-1, rank order
    rng = np.random.RandomState(0)
    W, b, th = rng.randn(3, 20, 3), rng.rand(3, 20), rng.randn(3, 20)
    rr = e.rff_sweep(W, b, th, 0.1, Z, k=1, want_all=False)
    tv, ti = c.topk_allgather(3, 0, 0)
    np.testing.assert_array_equal(tv, rr['top_val'][:, 0])
    np.testing.assert_array_equal(ti, rr['top_idx'][:, 0])
    c.close()
    e.close()


def _nccl_one_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import datetime
    import torch
    import torch.distributed as dist
    from pybo_amd import dist as pdist
    try:
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=120))
        probe = torch.tensor([1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(probe)
        te = torch.tensor([2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize(dev)
        table = pdist._allgather_f64(np.arange(6, dtype=np.float64))        # the device-tensor all-gather of the exchange
        recs = [None]
        dist.all_gather_object(recs, {'rank': 0, 'ms': 1.5})
        box = [7]
        dist.broadcast_object_list(box, src=0)
        ver = torch.cuda.nccl.version()
        dist.destroy_process_group()
        q.put(('ok', float(probe.item()), float(te.item()), table.tolist(), recs, box, tuple(ver)))
    except BaseException as exc:      # noqa: reported to the parent
        q.put(('error', repr(exc)))


def test_torch_distributed_over_rccl_with_one_rank():
    """The calls bench.py and pybo_amd.dist make at N > 1 -- init_process_group('nccl', device_id=...), all_reduce (SUM and
    MAX) of device float64 tensors, barrier, the all-gather of the exchange, all_gather_object, broadcast_object_list,
    the RCCL version query -- on the REAL RCCL with a one-rank world (the most a one-GPU box can run: RCCL refuses two
    ranks on one device).  Proves the library loads and every call has the signature this torch build expects."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert got[0] == 'ok', got
    assert got[1] == 1.0 and got[2] == 2.5
    assert got[3] == [[0.0, 1.0, 2.0, 3.0, 4.0, 5.0]]
    assert got[4] == [{'rank': 0, 'ms': 1.5}] and got[5] == [7]
    assert len(got[6]) >= 2


def _one_librccl_worker(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import datetime
    import torch
    import torch.distributed as dist
    try:
        from pybo_amd._lib import Engine, Comm
        from helpers import synth_problem

        def mapped():
            return sorted(set(ln.split()[-1] for ln in open('/proc/self/maps') if 'librccl' in ln))
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=120))
        probe = torch.tensor([1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize(dev)
        before = mapped()                                   # torch's RCCL is in the process now
        X, y, ell = synth_problem(300, 3, seed=4)
        e = Engine(0, torch.cuda.current_stream(dev).cuda_stream)
        e.fit(X, y, 'se', ell, 1.2, 1e-3, 0.1)
        c = Comm(e, 0, 1, Comm.unique_id())                 # libgpx binds librccl by soname (comm.hip: dlopen)
        after = mapped()
        r = e.sweep('ei', 0.4, np.random.RandomState(3).rand(4000, 3), k=5, want_all=False)
        tv, ti = c.topk_allgather(5, 100, 5)                # libgpx's communicator ...
        dist.all_reduce(probe)                              # ... and torch's, interleaved in one process
        torch.cuda.synchronize(dev)
        ok = bool(np.array_equal(ti, r['top_idx'] + 100) and np.array_equal(tv, r['top_val']))
        c.close()
        e.close()
        dist.destroy_process_group()
        q.put(('ok', before, after, ok, float(probe.item())))
    except BaseException as exc:      # noqa: reported to the parent
        q.put(('error', repr(exc)))


def test_libgpx_binds_the_librccl_torch_has_loaded_one_instance_per_process():
    """bench.py --gpus N runs torch.distributed's RCCL process group AND libgpx's own communicator (the device-side exchange,
    the default transport since round 6) in one process.  comm.hip dlopen()s librccl by its soname, which resolves to the copy
    torch has already mapped: ONE instance of the library (one set of RCCL globals) in /proc/self/maps, and collectives of
    both communicators interleave."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_one_librccl_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=600)
    p.join(timeout=120)
    assert got[0] == 'ok', got
    before, after = got[1], got[2]
    assert len(before) == 1, before                     # torch brought exactly one librccl
    assert after == before, (before, after)             # libgpx did not map a second one
    assert got[3] and got[4] == 1.0


def _exchange_fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import argparse
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench
        from pybo_amd._lib import Engine
        e = Engine(0)
        args = argparse.Namespace(exchange='auto', share_device=-1, backend='gloo')      # as if every rank had a GPU of its own
        got = bench.bring_up_gpx_exchange(args, e, rank, world, dist, torch.device('cuda', 0))
        q.put((rank, got if isinstance(got, str) else 'gpx'))
        e.close()
    except BaseException as exc:      # noqa: reported to the parent
        q.put((rank, 'error: %r' % (exc,)))
    finally:
        dist.destroy_process_group()


def test_device_side_exchange_falls_through_together_when_rccl_refuses_the_communicator():
    """bench.py's default transport is libgpx's own RCCL communicator.  Here two ranks (gloo process group) sit on ONE GPU without
    saying so: the binding loads on both, ncclCommInitRank refuses the duplicate device on both, and BOTH ranks come back with the
    torch transport and the reason -- nobody hangs in a rendezvous, nobody is left alone on the other transport."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
    assert set(got) == {0, 1}
    for r in (0, 1):
        assert got[r].startswith('torch (fallback: gpx_comm_init failed'), got


# ---- the whole loop SPMD: rank 0 evaluates the objective, everyone absorbs the same observation --------------------
def _spmd_loop_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank,) + _spmd_loop(rank))
    finally:
        dist.destroy_process_group()


def _spmd_loop(rank):
    import pybo_amd
    from pybo_amd import models, inits
    from helpers import branin
    bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
    rng = np.random.RandomState(0)
    X = bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * rng.rand(500, 2)
    y = -branin(X) / 10.0 + 1e-3 * rng.randn(500)
    model = models.make_gp(1e-4 * np.var(y), np.var(y), 0.25 * (bounds[:, 1] - bounds[:, 0]), np.mean(y))
    model.add_data(X, y)
    noise = np.random.RandomState(4000 + rank)        # NOISY objective: rank-dependent if it were called per rank
    calls = []

    def objective(x):
        calls.append(1)
        return float(-branin(x)[0] / 10.0 + 0.05 * noise.randn())

    grid = inits.init_sobol_device(bounds, 40000, rng=9)          # every rank holds the grid, sweeps its view of it
    xbest, fitted, info = pybo_amd.solve_bayesopt(objective, bounds, model=model, niter=5, policy='ei',
                                                  solver=('lbfgs', {'xgrid': grid}), recommender='latent', rng=1, spmd=True)
    return len(calls), info.x, info.y, info.xbest, _digest(fitted._engine().get_matrix('L'))


def test_spmd_loop_with_a_noisy_objective_keeps_every_rank_s_model_bitwise_equal():
    """VERDICT round 2, missing #2: under torch.distributed every rank ran the loop AND called objective(x) itself.
    Now rank 0 evaluates and broadcasts (pybo_amd.dist.spmd_objective, wired into solve_bayesopt), and the solver
    shards the grid stage (solve_bayesopt(..., spmd=True) wraps the index in dist.ShardedIndex): after 5 iterations with a noisy objective the ranks hold bitwise
    equal factors, identical traces, and the objective ran 6 times in total (box centre + 5), all on rank 0."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_spmd_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((g[0], g[1:]) for g in (q.get(timeout=600) for _ in range(world)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == 6 and got[1][0] == 0
    for a, b in zip(got[0][1:4], got[1][1:4]):
        np.testing.assert_array_equal(a, b)
    assert got[0][4] == got[1][4]
