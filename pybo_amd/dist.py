"""
Multi-GPU layout of the acquisition sweep: one process per GPU, candidates split into contiguous index
ranges, every rank fits the GP redundantly (deterministic -> bitwise-identical factor, no communication),
and the only exchange is ONE all-gather of each rank's k best (value, GLOBAL index) pairs -- 16*k bytes per
rank -- after which every rank computes the same merged top-k with the deterministic rule
(value descending, then global index ascending).  RCCL has no MAXLOC, hence gather + local merge
(SURVEY.md F13).  The reference has no distributed code at all; this is new.

Two transports for that exchange:
  * any initialised torch.distributed backend: 'nccl' (= RCCL over xGMI on MI355X; tensors staged on the
    current device) or 'gloo' (CPU tensors; used by the multi-rank tests) -- the default;
  * `comm=` a pybo_amd._lib.Comm: libgpx's own RCCL binding (gpx_comm_init / gpx_topk_allgather, the C-ABI a
    non-Python consumer uses): the pairs go device -> xGMI -> device merge, only the k winners reach the host.
"""
import numpy as np

__all__ = ['shard_bounds', 'merge_topk', 'gather_topk', 'gather_pairs', 'sharded_topk', 'ShardedIndex', 'world_size',
           'spmd_objective', 'broadcast_seed']


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


def _allgather_f64(mine, group=None):
    """ONE all-gather of a float64 vector (same length on every rank) -> (world, len) array on the host."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' \
        else torch.device('cpu')
    send = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.float64)).to(dev)
    everyone = torch.empty(world * send.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(everyone, send, group=group)
    return everyone.cpu().numpy().reshape(world, -1)


def gather_pairs(vals, idx, group=None):
    """All-gather of n (value, index) pairs per rank (same n everywhere) WITHOUT a merge: every rank receives
    all world*n pairs in rank order.  One collective: values and indices travel in the same float64 buffer
    (indices < 2^53 are exact).  Batch-BO / Thompson: one recommendation per posterior draw, draws sharded
    over ranks."""
    dist = _dist()
    vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
    if dist is None or dist.get_world_size(group) == 1:
        return vals, idx
    n = len(vals)
    table = _allgather_f64(np.concatenate([vals, idx.astype(np.float64)]), group).reshape(-1, 2, n)
    return table[:, 0, :].reshape(-1), table[:, 1, :].astype(np.int64).reshape(-1)


def gather_topk(vals, idx, k, group=None):
    """All-gather per-rank (value, global index) candidates and merge; identical result on every rank.
    Without an initialised process group this is a local merge."""
    dist = _dist()
    vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
    if dist is None or dist.get_world_size(group) == 1:
        return merge_topk(vals, idx, k)
    if len(vals) > k:                      # only the k best of a rank can make the global top-k
        vals, idx = merge_topk(vals, idx, k)
    n = len(vals)
    # ranks may hold fewer than k pairs (k > shard size): pad to k with index -1 (no extra collective needed)
    nmax = int(k)
    pv = np.full(nmax, -np.inf)
    pi = np.full(nmax, -1, dtype=np.int64)
    pv[:n], pi[:n] = vals, idx
    allv, alli = gather_pairs(pv, pi, group)
    return merge_topk(allv, alli, k)


def sharded_topk(index, xgrid, k, group=None, comm=None):
    """Rank-local `index.topk` on this rank's slice of `xgrid`, then the exchange.

    comm=None: torch.distributed all-gather of the host pairs (any backend).
    comm=pybo_amd._lib.Comm: the pairs are taken from the device buffers the local sweep left behind and
    gathered + merged by libgpx over RCCL (gpx_topk_allgather); needs every shard to hold at least k
    candidates and `index` to come from a device model driven by the communicator's engine."""
    dist = _dist()
    if comm is not None:
        rank, world = comm.rank, comm.nranks
    else:
        rank = dist.get_rank(group) if dist else 0
        world = dist.get_world_size(group) if dist else 1
    from ._lib import DeviceGrid
    M = len(xgrid)
    lo, hi = shard_bounds(M, rank, world)
    # this rank's rows: a host array is sliced; a DeviceGrid (the whole grid resident on this rank's GPU) is
    # viewed in place -- the same view object every call, so a device model's warm sweep cache keeps working
    if isinstance(xgrid, DeviceGrid):
        mine = xgrid.view(lo, hi)
    else:
        if not isinstance(xgrid, np.ndarray):
            xgrid = np.asarray(xgrid, dtype=float)
        mine = xgrid[lo:hi]
    if comm is not None:
        # every precondition is checked BEFORE the local sweep and depends on arguments all ranks share (M, k, world)
        # or on this rank's own objects, so a violation raises on every rank alike instead of leaving the others
        # blocked in the collective
        if M // world < k:
            raise ValueError('device exchange needs at least k candidates on every rank')
        owner = getattr(index, 'topk_engine', None)
        if owner is None or owner() is not comm._engine:
            raise ValueError('the communicator is bound to another engine than the one that runs the sweep')
        index.topk(mine, k)
        return comm.topk_allgather(k, lo, k)
    if hi > lo:
        vals, idx = index.topk(mine, min(k, hi - lo))
        idx = np.asarray(idx, dtype=np.int64)
        good = idx >= 0
        vals, idx = np.asarray(vals)[good], idx[good] + lo
    else:
        vals, idx = np.empty(0), np.empty(0, dtype=np.int64)
    return gather_topk(vals, idx, k, group)


class ShardedIndex(object):
    """Wrap an index so that `solve_lbfgs` (or any caller of `.topk`) transparently sweeps the grid across
    all ranks of the process group:  solver(ShardedIndex(policy(model, bounds, X)), bounds)."""

    def __init__(self, index, group=None, comm=None):
        self._index, self._group, self._comm = index, group, comm

    def __call__(self, X, grad=False):
        return self._index(X, grad=grad)

    def topk(self, xgrid, k):
        return sharded_topk(self._index, xgrid, k, self._group, self._comm)


# ---- SPMD runs of the whole loop: ONE objective evaluation per iteration ---------------------------------------
def world_size(group=None):
    dist = _dist()
    return dist.get_world_size(group) if dist else 1


def spmd_objective(objective, group=None, src=0):
    """pybo evaluates the black box ONCE per iteration in one process (pybo/bayesopt.py:268).  When the whole loop
    runs SPMD (one rank per GPU under torch.distributed, every rank executing solve_bayesopt(..., spmd=True)), only
    rank `src` calls `objective`; the query point it used AND the value it got are broadcast, and every rank feeds its
    replicated model that SAME observation -- with a noisy or expensive objective, P independent evaluations would
    give P different models whose shard-local top-k lists cannot be merged.

    The wrapper is called like the objective, `y = f(x)`; `f.exchange(x)` returns `(x_src, y)`, the pair every rank
    must absorb (the loop uses this form).  A rank whose own x differs from rank src's is counted in `f.mismatches`
    (replicated models are bitwise equal, so this stays 0 unless the ranks were seeded differently) -- the src's point
    wins either way, so the models cannot drift apart silently.  Without a process group (or with one rank) the
    objective is returned unchanged."""
    dist = _dist()
    if dist is None or dist.get_world_size(group) == 1:
        return objective
    rank = dist.get_rank(group)

    def exchange(x):
        box = [None]
        if rank == src:
            try:
                box[0] = ('ok', np.array(x, dtype=float), objective(x))
            except BaseException as exc:          # noqa: every rank must leave the collective, then re-raise
                box[0] = ('error', None, repr(exc))
                dist.broadcast_object_list(box, src=src, group=group)
                raise
        dist.broadcast_object_list(box, src=src, group=group)
        status, x_src, value = box[0]
        if status != 'ok':
            raise RuntimeError('objective failed on rank %d: %s' % (src, value))
        mine = np.asarray(x, dtype=float)
        if mine.shape != x_src.shape or not np.array_equal(mine, x_src):
            evaluate.mismatches += 1
        return x_src.reshape(mine.shape) if mine.size == x_src.size else x_src, value

    def evaluate(x):
        return exchange(x)[1]

    evaluate.exchange = exchange
    evaluate.mismatches = 0
    evaluate.spmd = True
    evaluate.group = group
    return evaluate


def broadcast_seed(rng, group=None, src=0):
    """SPMD runs: every rank must draw the SAME initial design, candidate grids and posterior samples.  An integer seed
    or a RandomState the caller seeded identically on every rank passes through; `None` (OS entropy: a different
    stream per rank) is replaced by one seed drawn on rank `src` and broadcast."""
    dist = _dist()
    if rng is not None or dist is None or dist.get_world_size(group) == 1:
        return rng
    box = [int(np.random.RandomState().randint(0, 2 ** 31 - 1))] if dist.get_rank(group) == src else [None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]
