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
