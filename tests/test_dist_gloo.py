"""The N > 1 path on CPU: world_size-2 gloo process group exercising the candidate sharding and the
(value, global index) all-gather + deterministic merge that bench.py and pybo_amd.dist use on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pybo_amd import dist as pdist


def test_shard_bounds_partition():
    for M in (1, 2, 7, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            cuts = [pdist.shard_bounds(M, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == M
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_merge_rule():
    v = np.array([1.0, 5.0, 5.0, np.nan, 2.0, -np.inf, 9.0])
    i = np.array([10, 7, 3, 1, 4, -1, -1])
    mv, mi = pdist.merge_topk(v, i, 4)
    assert list(mi) == [3, 7, 4, 10] and list(mv) == [5.0, 5.0, 2.0, 1.0]


class _HostIndex(object):
    """An index over a fixed value table (no GPU): topk ranks the slice it is handed."""

    def __init__(self, table):
        self.table = table

    def topk(self, xgrid, k):
        v = self.table[xgrid[:, 0].astype(int)]
        vv = np.where(np.isnan(v), -np.inf, v)
        order = np.lexsort((np.arange(len(v)), -vv))[:k]
        return v[order], order


def _worker(rank, world, port, M, k, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(0)              # same table on every rank
        table = rng.randn(M)
        table[rng.randint(0, M, 5)] = table.max() + 1.0          # exact ties across shards
        xgrid = np.arange(M, dtype=float)[:, None]
        vals, idx = pdist.sharded_topk(_HostIndex(table), xgrid, k)
        wrapped = pdist.ShardedIndex(_HostIndex(table))
        v2, i2 = wrapped.topk(xgrid, k)
        q.put((rank, vals, idx, v2, i2))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('M,k', [(1000, 10), (3, 5), (4097, 64)])
def test_world2_sharded_topk_equals_global(M, k):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.RandomState(0)
    table = rng.randn(M)
    table[rng.randint(0, M, 5)] = table.max() + 1.0
    order = np.lexsort((np.arange(M), -table))[:k]
    for rank, vals, idx, v2, i2 in got:
        np.testing.assert_array_equal(idx, order)
        np.testing.assert_array_equal(vals, table[order])
        np.testing.assert_array_equal(i2, order)


def _pairs_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # batch-BO: each rank holds the (value, index) winners of ITS draws; one all-gather, rank order, no merge
        vals = np.array([10.0 * rank + 0.5, 10.0 * rank + 1.5, -np.inf])
        idx = np.array([1000 * rank + 7, 2 ** 40 + rank, -1], dtype=np.int64)
        q.put((rank,) + tuple(pdist.gather_pairs(vals, idx)))
    finally:
        dist.destroy_process_group()


def test_world2_gather_pairs_is_one_collective_in_rank_order():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pairs_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, vals, idx in got:
        np.testing.assert_array_equal(vals, [0.5, 1.5, -np.inf, 10.5, 11.5, -np.inf])
        np.testing.assert_array_equal(idx, [7, 2 ** 40, -1, 1007, 2 ** 40 + 1, -1])       # indices beyond 2^32 survive


def test_device_exchange_branch_checks_its_preconditions():
    """sharded_topk(comm=...) (libgpx's own RCCL exchange) without a GPU: the communicator must be bound to the
    engine that ran the sweep, and every shard must hold at least k candidates."""
    class Eng(object):
        pass

    class FakeComm(object):
        def __init__(self, engine, rank, nranks):
            self._engine, self.rank, self.nranks, self.calls = engine, rank, nranks, []

        def topk_allgather(self, n, off, k):
            self.calls.append((n, off, k))
            return np.arange(k, dtype=float), np.arange(k, dtype=np.int64)

    eng, other = Eng(), Eng()
    seen = []

    def index(X, grad=False):
        return np.zeros(len(X))

    def topk(xgrid, k):
        seen.append((len(xgrid), k))
        return np.zeros(k), np.arange(k)
    index.topk = topk
    index.topk_engine = lambda: eng
    grid = np.arange(100.0)[:, None]
    comm = FakeComm(eng, rank=1, nranks=4)
    v, i = pdist.ShardedIndex(index, comm=comm).topk(grid, 5)
    assert seen == [(25, 5)] and comm.calls == [(5, 25, 5)] and len(v) == 5
    with pytest.raises(ValueError):
        pdist.sharded_topk(index, grid, 30, comm=comm)                 # 25 candidates per shard < k
    with pytest.raises(ValueError):
        pdist.sharded_topk(index, grid, 5, comm=FakeComm(other, 1, 4))  # bound to another engine
    del index.topk_engine
    with pytest.raises(ValueError):
        pdist.sharded_topk(index, grid, 5, comm=comm)                  # a host index has no device pairs


# ---- SPMD runs of the whole loop: the objective is evaluated ONCE per iteration (pybo/bayesopt.py:268) -------------
def _spmd_worker(rank, world, port, q, mode):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pybo_amd
        from helpers import SmootherModel
        calls = []
        noise = np.random.RandomState(1000 + rank)           # a NOISY objective: every rank would see other values

        def objective(x):
            calls.append(np.array(x))
            return float(-np.sum((np.ravel(x) - 0.3) ** 2) + 0.3 * noise.randn())

        bounds = [[0.0, 1.0], [0.0, 1.0]]
        kw = dict(model=SmootherModel(), niter=5, policy='ei', solver=('lbfgs', {'ngrid': 200}), recommender='incumbent')
        if mode == 'resume':
            # a run that is resumed from rank 0's checkpoint: the file exists ONLY where rank 0 looks (rank 1's path is a
            # directory of its own: no shared file system) -- 3 iterations are on disk, 5 are asked for
            import tempfile
            log = os.path.join(tempfile.mkdtemp(prefix='rank%d_' % rank), 'bo.pkl')
            if rank == 0:
                first = dict(kw, niter=3, rng=3, log=log)
                pybo_amd.solve_bayesopt(lambda x: float(-np.sum((np.ravel(x) - 0.3) ** 2)), bounds, **first)
                assert os.path.exists(log)
            dist.barrier()
            kw.update(rng=3, spmd=True, log=log)
        elif mode == 'corrupt':
            # rank 0's checkpoint is unreadable: EVERY rank must raise (the others used to block in the broadcast for ever)
            import tempfile
            log = os.path.join(tempfile.mkdtemp(prefix='rank%d_' % rank), 'bo.pkl')
            if rank == 0:
                with open(log, 'wb') as fh:
                    fh.write(b'not a pickle')
            dist.barrier()
            kw.update(rng=3, spmd=True, log=log)
            try:
                pybo_amd.solve_bayesopt(objective, bounds, **kw)
                q.put((rank, 'no error'))
            except RuntimeError as exc:
                q.put((rank, 'raised' if 'could not load the checkpoint' in str(exc) else repr(exc)))
            return
        elif mode == 'seeded':
            kw.update(rng=3, spmd=True)
        elif mode == 'unseeded':          # rng=None: OS entropy per rank unless the loop broadcasts one seed
            kw.update(rng=None, spmd=True)
        else:                             # 'independent': a process group exists, the loop was NOT asked to use it
            kw.update(rng=10 + rank)
        xbest, model, info = pybo_amd.solve_bayesopt(objective, bounds, **kw)
        q.put((rank, len(calls), info.x, info.y, model.X, model.Y, xbest))
    finally:
        dist.destroy_process_group()


def _run_spmd(mode, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_spmd_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((g[0], g[1:]) for g in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize('mode', ['seeded', 'unseeded'])
def test_world2_spmd_loop_evaluates_the_objective_on_one_rank_and_keeps_the_models_equal(mode):
    """with rng=None every rank used to draw its own grids (OS entropy), refine different seeds and
    absorb (its own x, rank 0's y).  Now the seed is broadcast once and the pair (x, y) rank 0 evaluated is what every
    rank absorbs."""
    got = _run_spmd(mode)
    assert got[0][0] == 6 and got[1][0] == 0            # box centre + 5 iterations, all on rank 0
    for a, b in zip(got[0][1:], got[1][1:]):            # traces, model data and recommendation: bitwise equal
        np.testing.assert_array_equal(a, b)
    assert len(got[0][2]) == 6


def test_world2_spmd_resume_loads_the_checkpoint_on_rank_0_and_broadcasts_it():
    """every rank used to safe_load(log) on its own although only rank 0 writes that file: without
    a shared file system (or with start-up skew) the ranks resumed from different traces, issued different numbers of
    objective exchanges and recorded rank 0's pair under the wrong index.  Rank 0 loads, the state is broadcast."""
    got = _run_spmd('resume')
    assert got[0][0] == 2 and got[1][0] == 0            # iterations 3 and 4 only, on rank 0
    for a, b in zip(got[0][1:], got[1][1:]):
        np.testing.assert_array_equal(a, b)
    assert len(got[0][2]) == 6                          # box centre + 5 iterations in the trace of BOTH ranks


def test_world2_spmd_resume_from_an_unreadable_checkpoint_fails_on_every_rank():
    """rank 0 loaded the checkpoint BEFORE the broadcast; a corrupt file raised there and left the other
    ranks blocked in broadcast_object_list.  Now the failure is what is broadcast."""
    got = _run_spmd('corrupt')
    assert got[0] == ('raised',) and got[1] == ('raised',)


def test_a_process_group_alone_does_not_make_the_loop_collective():
    """SPMD is opt-in.  Two ranks with an initialised group and DIFFERENT seeds run two
    independent optimisations: each evaluates its own objective, nothing is exchanged, nothing hangs."""
    got = _run_spmd('independent')
    assert got[0][0] == 6 and got[1][0] == 6
    assert not np.array_equal(got[0][1], got[1][1])


def test_spmd_objective_broadcasts_the_query_point():
    """The wrapper's exchange() returns rank 0's (x, y) on every rank and counts ranks that asked about another point."""
    got = _run_pairs()
    assert got[0] == (0, [0.25, 0.5], 7.0) and got[1] == (1, [0.25, 0.5], 7.0)


def _pair_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        f = pdist.spmd_objective(lambda x: 7.0)
        x, y = f.exchange(np.array([0.25, 0.5]) + rank)      # rank 1 proposes another point
        assert pdist.broadcast_seed(5) == 5
        seed = pdist.broadcast_seed(None)
        q.put((rank, (f.mismatches, list(map(float, x)), float(y), int(seed))))
    finally:
        dist.destroy_process_group()


def _run_pairs(world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][3] == got[1][3]                        # one seed for everybody
    return {r: (got[r][0], got[r][1], got[r][2]) for r in got}


def test_spmd_objective_is_the_identity_without_a_process_group():
    f = lambda x: 1.0                                    # noqa: E731
    assert pdist.spmd_objective(f) is f and pdist.world_size() == 1
