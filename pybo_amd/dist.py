"""
Multi-GPU layout of the acquisition sweep: one process per GPU, candidates split into contiguous index
ranges, every rank fits the GP redundantly (deterministic -> bitwise-identical factor, no communication),
and the only exchange is ONE all-gather of each rank's k best (value, GLOBAL index) pairs -- 16*k bytes per
rank -- after which every rank computes the same merged top-k with the deterministic rule
(value descending, then global index ascending).  RCCL has no MAXLOC, hence gather + local merge
(SURVEY.md F13).  The reference has no distributed code at all; this is new.

Works with any initialised torch.distributed backend: 'nccl' (= RCCL over xGMI on MI355X; tensors staged on
the current device) or 'gloo' (CPU tensors; used by the world_size-2 tests).
"""
import numpy as np

__all__ = ['shard_bounds', 'merge_topk', 'gather_topk', 'sharded_topk', 'ShardedIndex']


def shard_bounds(M, rank, world):
    """Contiguous slice [lo, hi) of M candidates owned by `rank` (sizes differ by at most one)."""
    return (M * rank) // world, (M * (rank + 1)) // world


def merge_topk(vals, idx, k):
    """k best of concatenated candidates: value descending, ties by ascending global index;
    NaN ranks last; padding entries (index < 0) are dropped."""
    vals = np.asarray(vals, dtype=float).reshape(-1)
    idx = np.asarray(idx, dtype=np.int64).reshape(-1)
    keep = idx >= 0
    vals, idx = vals[keep], idx[keep]
    v = np.where(np.isnan(vals), -np.inf, vals)
    order = np.lexsort((idx, -v))[:k]
    return vals[order], idx[order]


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except ImportError:
        pass
    return None


def gather_topk(vals, idx, k, group=None):
    """All-gather per-rank (value, global index) candidates and merge; identical result on every rank.
    Without an initialised process group this is a local merge."""
    dist = _dist()
    vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
    if dist is None or dist.get_world_size(group) == 1:
        return merge_topk(vals, idx, k)
    import torch
    world = dist.get_world_size(group)
    if len(vals) > k:                      # only the k best of a rank can make the global top-k
        vals, idx = merge_topk(vals, idx, k)
    n = len(vals)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
        else torch.device('cpu')
    # ranks may hold fewer than k pairs (k > shard size): pad to k with index -1 (no extra collective needed)
    nmax = int(k)
    pv = np.full(nmax, -np.inf)
    pi = np.full(nmax, -1, dtype=np.int64)
    pv[:n], pi[:n] = vals, idx
    # ONE collective per step: values and indices travel in the same float64 buffer (indices < 2^53 are exact)
    mine = torch.from_numpy(np.concatenate([pv, pi.astype(np.float64)])).to(dev)
    everyone = torch.empty(world * 2 * nmax, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    table = everyone.cpu().numpy().reshape(world, 2, nmax)
    return merge_topk(table[:, 0, :], table[:, 1, :].astype(np.int64), k)


def sharded_topk(index, xgrid, k, group=None):
    """Rank-local `index.topk` on this rank's slice of `xgrid`, then the exchange."""
    dist = _dist()
    rank = dist.get_rank(group) if dist else 0
    world = dist.get_world_size(group) if dist else 1
    xgrid = np.asarray(xgrid)
    lo, hi = shard_bounds(len(xgrid), rank, world)
    if hi > lo:
        vals, idx = index.topk(xgrid[lo:hi], min(k, hi - lo))
        idx = np.asarray(idx, dtype=np.int64)
        good = idx >= 0
        vals, idx = np.asarray(vals)[good], idx[good] + lo
    else:
        vals, idx = np.empty(0), np.empty(0, dtype=np.int64)
    return gather_topk(vals, idx, k, group)


class ShardedIndex(object):
    """Wrap an index so that `solve_lbfgs` (or any caller of `.topk`) transparently sweeps the grid across
    all ranks of the process group:  solver(ShardedIndex(policy(model, bounds, X)), bounds)."""

    def __init__(self, index, group=None):
        self._index, self._group = index, group

    def __call__(self, X, grad=False):
        return self._index(X, grad=grad)

    def topk(self, xgrid, k):
        return sharded_topk(self._index, xgrid, k, self._group)
