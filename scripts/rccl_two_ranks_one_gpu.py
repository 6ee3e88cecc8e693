"""Probe: can TWO ranks of one RCCL communicator live on ONE GPU?  (It decides whether the N > 1 path of
gpx_topk_allgather can be exercised on a one-GPU box.)  Spawns two processes that build a 2-rank gpx_comm on
device 0 and run one exchange; prints what happened.  Run under `timeout`: a refused communicator may block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.multiprocessing as mp


def work(rank, q_in, q_out):
    from pybo_amd._lib import Engine, Comm
    try:
        e = Engine(0)
        rng = np.random.RandomState(0)
        X = rng.rand(200, 2); y = np.sin(3 * X.sum(1))
        e.fit(X, y, 'se', [0.3, 0.3], 1.0, 1e-3, 0.0)
        uid = q_in.get(timeout=60)
        c = Comm(e, rank, 2, uid)
        Z = np.random.RandomState(1).rand(4000, 2)
        lo, hi = (0, 2000) if rank == 0 else (2000, 4000)
        e.sweep('ei', 0.3, Z[lo:hi], k=5, want_all=False)
        q_out.put((rank, 'ok', c.topk_allgather(5, lo, 5)))
    except Exception as ex:          # noqa: BLE001
        q_out.put((rank, 'error', repr(ex)))


if __name__ == '__main__':
    from pybo_amd._lib import Comm
    ctx = mp.get_context('spawn')
    qs = [ctx.Queue(), ctx.Queue()]
    out = ctx.Queue()
    ps = [ctx.Process(target=work, args=(r, qs[r], out)) for r in range(2)]
    for p in ps:
        p.start()
    uid = Comm.unique_id()
    for q in qs:
        q.put(uid)
    for _ in range(2):
        try:
            print(out.get(timeout=90))
        except Exception as ex:      # noqa: BLE001
            print('no answer within 90 s:', repr(ex))
    for p in ps:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
