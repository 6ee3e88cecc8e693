"""CPU tests of the task-graph factorisation's task lists (pybo_amd/csrc/kernels_chol_tg.hip, host part; exported as
gpx_chol_tasks2): replayed under the device's own protocol -- tickets drawn in list order, a task starts when its
dependencies are met, tasks complete in any order -- the lists must (1) never dead-lock, (2) apply every block row to every tile exactly once and
in ascending order (what makes the factor bit-identical to the stream schedule's), (3) BE a Cholesky factorisation when the
tasks are executed with numpy on a small block size.  Serves `model.add_data` (pybo/bayesopt.py:114,258,269)."""
import numpy as np
import pytest

from pybo_amd import _lib

TRSM, UPD, UPDQ, SHADOW, TRSMU = 1, 2, 3, 4, 5


def tasks(nP, chunks=0):
    """The lists as the replay understands them.  On the device the tiles next to the diagonal belong to nine dedicated
    workgroups that FOLLOW the diagonal factorisation instead of drawing tasks (kernels_chol_tg.hip, "the shadows"): the first
    list holds one descriptor per block row p -- ord: chunks of every tile of row p (what a solve of that row waits for);
    [k0, k1) / aux: the final chunk of row p+1 and its ordinal; rsv: the first block row of the chunk before it on the diagonal
    tile (p+1, p+1), or -1.  Expanded here into the tasks those workgroups stand for, in an order a single critical list could
    run them: per block row the solves of tiles (p, p+1 .. p+3) [S1, S2, S3], the final chunks of tiles (p+1, p+2), (p+1, p+3)
    [V, V2: two workgroups each], the diagonal tile (p+2, p+2)'s chunk before its final one [U0], the diagonal tile (p+1, p+1)'s final chunk [U]."""
    q = _lib.chol_tasks(nP, chunks)
    d = q[0]
    assert len(d) == max(nP - 1, 0) and (len(d) == 0 or np.all(d[:, 0] == SHADOW))
    crit = []
    for p in range(nP - 1):
        typ, I, J, k0, k1, ordn, aux, rsv = (int(v) for v in d[p])
        assert (I, J, k1) == (p, p + 1, p + 1) and 0 <= k0 <= p
        for Jc in range(p + 1, min(p + 4, nP)):
            for h in range(2):
                crit.append((TRSM, p, Jc, 0, 0, ordn, h, 0))
        for Jc in range(p + 2, min(p + 4, nP)):
            crit.append((UPD, p + 1, Jc, k0, k1, aux, 0, 0))
        if p + 2 < nP:
            r2 = int(d[p + 1][7])
            if r2 >= 0:
                crit.append((UPD, p + 2, p + 2, r2, p + 1, int(d[p + 1][6]) - 1, 0, 0))
        for piece in range(6):
            crit.append((UPDQ, p + 1, p + 1, k0, k1, aux, piece, 0))
    # below the shadows' band a column's solve and the final chunk of the tile under it are ONE task per 64-column half
    # (TG_TRSMU: I = p, J, [k0, k1) = that chunk, aux = half, rsv = its ordinal); the pair of halves stands for two solve halves
    # and one tile update
    work = []
    w = q[1]
    i = 0
    while i < len(w):
        if int(w[i, 0]) != TRSMU:
            work.append(tuple(int(v) for v in w[i]))
            i += 1
            continue
        a, b = w[i], w[i + 1]
        assert int(b[0]) == TRSMU and tuple(a[1:6]) == tuple(b[1:6]) and (int(a[6]), int(b[6])) == (0, 1) and a[7] == b[7]
        typ, I, J, k0, k1, ordn, aux, rsv = (int(v) for v in a)
        assert J >= I + 4 and k1 == I + 1
        work.append((TRSM, I, J, 0, 0, ordn, 0, 0))
        work.append((TRSM, I, J, 0, 0, ordn, 1, 0))
        work.append((UPD, I + 1, J, k0, k1, rsv, 0, 0))
        i += 2
    return [np.array(crit, dtype=np.int64).reshape(-1, 8), np.array(work, dtype=np.int64).reshape(-1, 8)]



class Replay(object):
    """The control block of k_chol_tg and, optionally, the matrix it works on (block size nb instead of 128)."""

    def __init__(self, nP, queues, nb=0, seed=0):
        self.nP, self.q = nP, queues
        self.head = [0, 0]
        self.seq = np.zeros((nP, nP), dtype=int)        # chunks applied per tile
        self.applied = np.zeros((nP, nP), dtype=int)    # block rows applied per tile
        self.solved = np.zeros((nP, 2), dtype=int)
        self.diag = np.zeros(nP, dtype=bool)
        self.quad = np.zeros(nP, dtype=int)
        self.next_potrf = 0
        self.nb = nb
        self.rng = np.random.RandomState(seed)
        if nb:
            n = nP * nb
            A = self.rng.randn(n, n)
            self.K = A @ A.T + n * np.eye(n)
            self.S = np.triu(self.K).copy()
            self.R = np.zeros((n, n))

    def ready(self, t):
        typ, I, J, k0, k1, ordn = (int(v) for v in t[:6])
        if typ == TRSM:
            return bool(self.diag[I]) and self.seq[I, J] == ordn
        return self.seq[I, J] == ordn and min(self.solved[I].min(), self.solved[J].min()) >= k1

    def blk(self, M, I, J):
        nb = self.nb
        return M[I * nb:(I + 1) * nb, J * nb:(J + 1) * nb]

    def run_task(self, t):
        typ, I, J, k0, k1, ordn, aux = (int(v) for v in t[:7])
        nb, h = self.nb, self.nb // 2
        if typ == TRSM:
            assert self.applied[I, J] == I, 'panel solve of an incomplete tile'
            if nb:
                Rpp = self.blk(self.R, I, I)
                cols = slice(J * nb + aux * h, J * nb + (aux + 1) * h)
                self.R[I * nb:(I + 1) * nb, cols] = np.linalg.solve(Rpp.T, self.S[I * nb:(I + 1) * nb, cols])
        else:
            if typ == UPD:
                assert self.applied[I, J] == k0, 'chunks out of order: tile (%d, %d) has %d, task starts at %d' % (I, J, self.applied[I, J], k0)
            else:
                assert I == J and self.applied[I, I] == k0
            assert 0 <= k0 < k1 <= I
            if nb:
                A = self.R[k0 * nb:k1 * nb, I * nb:(I + 1) * nb]
                B = self.R[k0 * nb:k1 * nb, J * nb:(J + 1) * nb]
                D = A.T @ B
                if typ == UPD:
                    self.blk(self.S, I, J)[...] -= D
                else:
                    qd, half, h2 = aux >> 1, aux & 1, h // 2
                    r = slice(h, nb) if qd == 2 else slice(0, h)
                    c0 = (0 if qd == 0 else h) + h2 * half
                    c = slice(c0, c0 + h2)
                    self.blk(self.S, I, I)[r, c] -= D[r, c]

    def finish(self, t):
        typ, I, J, k0, k1, ordn, aux = (int(v) for v in t[:7])
        if typ == TRSM:
            assert self.solved[J, aux] == I
            self.solved[J, aux] = I + 1
        elif typ == UPD:
            self.applied[I, J] = k1
            self.seq[I, J] = ordn + 1
        else:
            self.quad[I] += 1
            if self.quad[I] == 6:
                self.applied[I, I] = k1

    def potrf(self, p):
        assert self.applied[p, p] == p or (p > 0 and self.quad[p] == 6)
        if self.nb:
            D = self.blk(self.S, p, p)
            D = np.triu(D) + np.triu(D, 1).T
            self.blk(self.R, p, p)[...] = np.linalg.cholesky(D).T

    def run(self, max_inflight=7):
        inflight = []          # tasks taken, not yet published
        steps = 0
        total = sum(len(q) for q in self.q) + self.nP
        done = 0
        while done < total:
            steps += 1
            assert steps < 50 * total + 1000, 'no progress: dead-lock in the task lists'
            moves = []
            # (the critical list has its own workgroups on the device: its tasks do not compete for the workers' slots)
            nwork = sum(1 for kind, t in inflight if kind == 'task' and not t[-1])
            for qi in range(2):
                if (qi == 0 or nwork < max_inflight) and self.head[qi] < len(self.q[qi]) and self.ready(self.q[qi][self.head[qi]]):
                    moves.append(('take', qi))
            if True:
                p = self.next_potrf
                if p < self.nP and not any(t[0] == 'potrf' for t in inflight) and (p == 0 or self.quad[p] == 6):
                    moves.append(('potrf', p))
            for i in range(len(inflight)):
                moves.append(('finish', i))
            assert moves, 'dead-lock: nothing ready, nothing in flight (heads %s, next diagonal block %d)' % (self.head, self.next_potrf)
            m = moves[self.rng.randint(len(moves))]
            if m[0] == 'take':
                t = self.q[m[1]][self.head[m[1]]]
                self.head[m[1]] += 1
                self.run_task(t)           # (results become visible at 'finish'; nobody may read them before)
                inflight.append(('task', tuple(int(v) for v in t) + (m[1] == 0,)))
            elif m[0] == 'potrf':
                self.potrf(m[1])
                inflight.append(('potrf', m[1]))
            else:
                kind, t = inflight.pop(m[1])
                if kind == 'potrf':
                    self.diag[t] = True
                    self.next_potrf = t + 1
                else:
                    self.finish(t)
                done += 1
        for I in range(self.nP):
            for J in range(I, self.nP):
                assert self.applied[I, J] == I, (I, J, self.applied[I, J])
        assert self.diag.all() and (self.solved[1:] == np.arange(1, self.nP)[:, None]).all()


def run_ticketed(r, nworkers=5, nside=2):
    """The device's protocol (kernels_chol_tg.hip: tg_take): a workgroup without a ticket draws the next one of its list AT
    ONCE (fetch-and-add), ready or not, and waits with it in hand; side-kicks (critical list) START their task on the tile's
    earlier chunks alone and wait for the last dependency inside the task.  Returns when everything is done; asserts that
    some workgroup can always move."""
    rng = r.rng
    total = sum(len(q) for q in r.q) + r.nP
    done = 0
    wgs = [{'queue': 0, 'held': None, 'busy': None, 'started': False} for _ in range(nside)] + \
          [{'queue': 1, 'held': None, 'busy': None, 'started': False} for _ in range(nworkers)]
    potrf_busy = None
    steps = 0

    def pre_ready(t):            # critical list: the tile's earlier chunks only
        return r.seq[int(t[1]), int(t[2])] == int(t[5])

    while done < total:
        steps += 1
        assert steps < 200 * total + 2000, 'no progress'
        moves = []
        for wi, wg in enumerate(wgs):
            qi = wg['queue']
            if wg['busy'] is not None:
                if wg['started']:
                    moves.append(('finish', wi))
                elif r.ready(wg['busy']):            # (a side-kick's task in hand may still wait for its last dependency)
                    moves.append(('start', wi))
                continue
            if wg['held'] is not None:
                if (pre_ready if qi == 0 else r.ready)(wg['held']):
                    moves.append(('run_held', wi))
            elif r.head[qi] < len(r.q[qi]):
                moves.append(('draw', wi))
        p = r.next_potrf
        if potrf_busy is None and p < r.nP and (p == 0 or r.quad[p] == 6):
            moves.append(('potrf', p))
        if potrf_busy is not None:
            moves.append(('potrf_done',))
        assert moves, 'dead-lock: heads %s, next diagonal block %d, held %s' % (r.head, r.next_potrf, [w['held'] for w in wgs])
        m = moves[rng.randint(len(moves))]
        if m[0] == 'draw':
            wg = wgs[m[1]]
            qi = wg['queue']
            wg['held'] = r.q[qi][r.head[qi]]
            r.head[qi] += 1
        elif m[0] == 'run_held':
            wg = wgs[m[1]]
            wg['busy'], wg['held'] = wg['held'], None
        elif m[0] == 'start':
            wg = wgs[m[1]]
            r.run_task(wg['busy'])
            wg['started'] = True
        elif m[0] == 'finish':
            wg = wgs[m[1]]
            r.finish(wg['busy'])
            wg['busy'] = None
            wg['started'] = False
            done += 1
        elif m[0] == 'potrf':
            r.potrf(m[1])
            potrf_busy = m[1]
        else:
            r.diag[potrf_busy] = True
            r.next_potrf = potrf_busy + 1
            potrf_busy = None
            done += 1
    for I in range(r.nP):
        for J in range(I, r.nP):
            assert r.applied[I, J] == I
    assert r.diag.all()


@pytest.mark.parametrize('nP,chunks,nworkers,nside', [(9, 0, 5, 2), (9, 1124, 3, 1), (14, 1248, 7, 8), (6, 11, 1, 1),
                                                      (20, 0, 40, 8), (12, 12489, 9, 3), (17, 0, 30, 12)])
def test_ticketed_protocol_with_held_tickets_never_deadlocks(nP, chunks, nworkers, nside):
    q = tasks(nP, chunks)
    for seed in range(4):
        r = Replay(nP, q, nb=4 if seed == 0 else 0, seed=seed)
        run_ticketed(r, nworkers=nworkers, nside=nside)
        if r.nb:
            R = np.triu(r.R)
            np.testing.assert_allclose(R.T @ R, r.K, rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize('nP', [1, 2, 3, 5, 8, 17, 40])
@pytest.mark.parametrize('chunks', [0, 1124, 14, 1128, 11, 1224, 1248, 12489, 149])
def test_lists_complete_in_order_without_deadlock(nP, chunks):
    q = tasks(nP, chunks)
    assert len(q) == 2
    ntr = sum(int((a[:, 0] == TRSM).sum()) for a in q)
    assert ntr == nP * (nP - 1)                                                 # two halves per off-diagonal tile
    for seed in range(3):
        Replay(nP, q, seed=seed).run(max_inflight=1 + 3 * seed)


@pytest.mark.parametrize('nP,chunks', [(2, 0), (7, 0), (12, 1124), (12, 13), (9, 1128)])
def test_lists_are_a_cholesky_factorisation(nP, chunks):
    q = tasks(nP, chunks)
    r = Replay(nP, q, nb=4, seed=nP)
    r.run(max_inflight=5)
    R = np.triu(r.R)
    np.testing.assert_allclose(R.T @ R, r.K, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(R, np.linalg.cholesky(r.K).T, rtol=1e-10, atol=1e-10)


def test_chunks_are_graded_towards_the_pivot():
    """Default lists: every tile's last chunk is one block (the update the next diagonal block waits for is short) and
    chunks never grow towards the pivot; within a step the solves come first, then the rows nearest the pivot."""
    nP = 24
    q = tasks(nP)
    both = np.concatenate([q[0], q[1]])
    upd = both[both[:, 0] == UPD]
    for I in range(3, nP - 1):
        mine = upd[(upd[:, 1] == I) & (upd[:, 2] == nP - 1)]
        sizes = (mine[:, 4] - mine[:, 3])[np.argsort(mine[:, 3])]
        assert sizes[-1] == 1 and sizes.sum() == I
        assert np.all(np.diff(sizes[1:]) <= 0)              # (the first chunk absorbs a short remainder)
    # generation order inside step p = 3: every link of block row 3 (solve + the final chunk of the tile below it) precedes
    # every other chunk that ends at boundary 4
    w = q[1]
    first_upd = min(i for i in range(len(w)) if w[i, 0] == UPD and w[i, 4] == 4 and w[i, 1] > 4)
    last_trsm = max(i for i in range(len(w)) if w[i, 0] == TRSM and w[i, 1] == 3)
    assert last_trsm < first_upd


def test_bad_arguments():
    assert _lib.load().gpx_chol_tasks2(0, 0, None, 0, _lib._ptr(np.zeros(2, dtype=np.int64))) == -1
    assert _lib.load().gpx_chol_tasks2(4, 0, None, 0, None) == -1
