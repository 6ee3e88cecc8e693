"""Discrete-event model of the task-graph Cholesky, v3: per-row task lists, workers take the first READY row head
(nearest row first = earliest deadline first).  Design aid for kernels_chol_tg.hip (not product code)."""
import heapq, argparse
from sim2 import boundaries, parse_D

class Sim:
    def __init__(s, nP, D, nworkers, nside, c, minfirst=2):
        s.nP, s.c = nP, c
        s.bnd = [boundaries(I, D, minfirst) for I in range(nP)]
        # row lists
        s.rows = []
        s.crit = []
        for I in range(nP):
            l = []
            nb = len(s.bnd[I]) - 1
            for j in range(nb):
                k1 = s.bnd[I][j + 1]
                for J in range(I, nP):
                    if k1 == I and J == I: continue
                    l.append(('upd', I, J, k1, j))
            for J in range(I + 2, nP):
                for h in (0, 1): l.append(('trsm', I, J, h))
            s.rows.append(l)
        for p in range(nP - 1):
            s.crit += [('trsm', p, p + 1, 0), ('trsm', p, p + 1, 1)] + [('updq', p + 1, q, p + 1) for q in range(3)]
        s.nchunk = [len(b) - 1 for b in s.bnd]
        s.head = [0] * nP
        s.chead = 0
        s.seq, s.cnt, s.qd = {}, {}, {}
        s.solved = [[0, 0] for _ in range(nP)]
        s.diag_done = [False] * nP
        s.t, s.ev, s.n = 0.0, [], 0
        s.idle_side, s.idle = nside, nworkers
        s.potrf_p, s.potrf_busy = 0, False
        s.work = 0.0; s.traffic = 0.0
        s.first_row = 0

    def ready(s, t):
        if t[0] == 'trsm':
            _, p, J, h = t
            return s.diag_done[p] and s.seq.get((p, J), 0) == s.nchunk[p]
        if t[0] == 'upd':
            _, I, J, k1, j = t
            return s.seq.get((I, J), 0) == j and min(s.solved[I] + s.solved[J]) >= k1
        _, I, q, k1 = t
        return s.seq.get((I, I), 0) == s.nchunk[I] - 1 and min(s.solved[I]) >= k1

    def start(s, t, crit):
        c = s.c
        if t[0] == 'trsm': dur = c['trsm']
        elif t[0] == 'updq': dur = c['updq']
        else:
            _, I, J, k1, j = t
            K = k1 - s.cnt.get((I, J), 0)
            assert K > 0
            near = (I - k1) <= c['fast_d']
            dur = c['ovh'] + (c['kblk_u'] if near else c['kblk']) * K
            s.work += c['kblk'] * K
            s.traffic += 0.262 + 0.262 * K
        if not crit: dur += c['claim']
        heapq.heappush(s.ev, (s.t + dur + c['hop'], s.n, t, crit)); s.n += 1

    def complete(s, t):
        if t[0] == 'trsm': s.solved[t[2]][t[3]] = t[1] + 1
        elif t[0] == 'upd':
            s.cnt[(t[1], t[2])] = t[3]; s.seq[(t[1], t[2])] = s.seq.get((t[1], t[2]), 0) + 1
        elif t[0] == 'updq':
            I = t[1]; s.qd[I] = s.qd.get(I, 0) + 1
            if s.qd[I] == 3: s.cnt[(I, I)] = t[3]; s.seq[(I, I)] = s.seq.get((I, I), 0) + 1
        else:
            s.diag_done[t[1]] = True; s.potrf_busy = False; s.potrf_p += 1

    def dispatch(s):
        if not s.potrf_busy and s.potrf_p < s.nP:
            p = s.potrf_p
            if p == 0 or s.seq.get((p, p), 0) == s.nchunk[p]:
                s.potrf_busy = True
                heapq.heappush(s.ev, (s.t + s.c['potrf'] + s.c['hop'], s.n, ('potrf', p), True)); s.n += 1
        while s.idle_side > 0 and s.chead < len(s.crit) and s.ready(s.crit[s.chead]):
            s.start(s.crit[s.chead], True); s.chead += 1; s.idle_side -= 1
        while s.first_row < s.nP and s.head[s.first_row] >= len(s.rows[s.first_row]): s.first_row += 1
        if s.c.get('lst'):
            while s.idle > 0:
                best, br = None, -1
                for r in range(s.first_row, s.nP):
                    if s.head[r] < len(s.rows[r]):
                        t = s.rows[r][s.head[r]]
                        if s.ready(t):
                            if t[0] == 'trsm': key = -1e9 + r
                            else: key = s.c['per'] * t[1] - s.c['kblk'] * (t[1] - s.cnt.get((t[1], t[2]), 0)) * s.c['lst']
                            if best is None or key < best: best, br = key, r
                if br < 0: break
                s.start(s.rows[br][s.head[br]], False); s.head[br] += 1; s.idle -= 1
            return
        r = s.first_row
        while s.idle > 0 and r < s.nP:
            if s.head[r] < len(s.rows[r]) and s.ready(s.rows[r][s.head[r]]):
                s.start(s.rows[r][s.head[r]], False); s.head[r] += 1; s.idle -= 1
            else:
                r += 1

    def run(s, trace=0):
        s.dispatch()
        while s.ev:
            tm, _, t, crit = heapq.heappop(s.ev)
            s.t = tm
            s.complete(t)
            if t[0] == 'potrf': pass
            elif crit: s.idle_side += 1
            else: s.idle += 1
            if trace and t[0] == 'potrf' and t[1] % trace == 0:
                per = s.c['potrf'] + 3 * s.c['hop'] + 10
                print(f"  potrf {t[1]:3d} done {s.t:8.1f} us (ideal {(t[1]+1)*per-per+s.c['potrf']+s.c['hop']:8.1f}) idle={s.idle}")
            s.dispatch()
        return s.t, all(s.diag_done)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--nP', type=int, default=64)
    ap.add_argument('--chunks', default='1,1,2,4*')
    ap.add_argument('--fast_d', type=int, default=2)
    ap.add_argument('--workers', type=int, default=506)
    ap.add_argument('--side', type=int, default=4)
    ap.add_argument('--potrf', type=float, default=30)
    ap.add_argument('--hop', type=float, default=2.5)
    ap.add_argument('--claim', type=float, default=3)
    ap.add_argument('--kblk', type=float, default=32)
    ap.add_argument('--kblk_u', type=float, default=18)
    ap.add_argument('--trace', type=int, default=0)
    ap.add_argument('--lst', type=float, default=0)
    a = ap.parse_args()
    c = dict(potrf=a.potrf, hop=a.hop, trsm=5, updq=5, ovh=4, kblk=a.kblk, kblk_u=a.kblk_u, fast_d=a.fast_d, claim=a.claim, lst=a.lst, per=a.potrf + 3 * a.hop + 10)
    D = parse_D(a.chunks, a.nP)
    s = Sim(a.nP, D, a.workers, a.side, c)
    t, ok = s.run(a.trace)
    flop = a.nP ** 3 / 3 * 128 ** 3
    print(f"nP={a.nP} chunks={a.chunks}: {t/1000:.3f} ms ok={ok} chain={a.nP*(a.potrf+3*a.hop+10)/1000:.3f} work={s.work/a.workers/1000:.3f} ms  "
          f"{flop/t/1e6:.1f} TFLOP/s traffic={s.traffic/1000:.1f} GB tasks={sum(len(r) for r in s.rows)}")
