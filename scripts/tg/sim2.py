"""Discrete-event model of the task-graph Cholesky, v2: graded chunk boundaries per row + multi-level general queues.
Design aid for kernels_chol_tg.hip (not product code).  Costs in microseconds."""
import heapq, argparse

LAZY = [0.0, 0]
def boundaries(I, D, minfirst):
    lim = max(LAZY[1], LAZY[0] * I) if LAZY[0] > 0 else 1e9
    b = sorted({I - d for d in D if I - d > 0 and d <= lim} | {0, I})
    # drop a short first chunk (merge into the next one)
    while len(b) > 2 and b[1] - b[0] < minfirst:
        del b[1]
    return b

def level_of(d, lv):
    # lv: list of distance thresholds; level = first j with d <= lv[j]
    for j, th in enumerate(lv):
        if d <= th: return j
    return len(lv)

def build(nP, D, lv, minfirst=2):
    """returns crit list, general queues (list per level); tasks ('trsm',p,J,h) ('upd',I,J,k1) ('updq',I,q,k1)"""
    nq = len(lv) + 1
    crit, gen = [], [[] for _ in range(nq)]
    bnd = [boundaries(I, D, minfirst) for I in range(nP)]
    ends = [dict() for _ in range(nP)]          # step p -> list of rows whose chunk ends at p+1
    for I in range(nP):
        for b in bnd[I][1:]:
            ends[b - 1].setdefault(None, []).append(I)
    for p in range(nP):
        for J in range(p + 1, nP):
            for h in (0, 1):
                (crit if J == p + 1 else gen[0]).append(('trsm', p, J, h))
        rows = sorted(ends[p].get(None, []))
        for I in rows:                          # nearest rows first
            d = I - (p + 1)
            l = level_of(d, lv)
            for J in range(I, nP):
                if d == 0 and J == I:
                    for q in range(3): crit.append(('updq', I, q, p + 1))
                else:
                    gen[l].append(('upd', I, J, p + 1))
    return crit, gen

class Sim:
    def __init__(s, nP, D, lv, nworkers, nside, c, reserve=None):
        s.nP, s.c = nP, c
        s.crit, s.gen = build(nP, D, lv)
        s.q = {'c': s.crit}
        for l, g in enumerate(s.gen): s.q[l] = g
        s.nl = len(s.gen)
        per_tile = {}
        for qn, q in s.q.items():
            for i, t in enumerate(q):
                if t[0] == 'upd': per_tile.setdefault((t[1], t[2]), []).append((t[3], str(qn), qn, i))
                elif t[0] == 'updq': per_tile.setdefault((t[1], t[1]), []).append((t[3], 'c', qn, i))
        s.ordinal, s.ntile = {}, {}
        for tile, l in per_tile.items():
            l.sort(key=lambda x: x[0])
            o, lastk = 0, None
            for k1, _, qn, i in l:
                if lastk == k1: s.ordinal[(qn, i)] = o - 1
                else: s.ordinal[(qn, i)] = o; o += 1
                lastk = k1
            s.ntile[tile] = o
        s.seq, s.cnt, s.qd = {}, {}, {}
        s.solved = [[0, 0] for _ in range(nP)]
        s.diag_done = [False] * nP
        s.heads = {k: 0 for k in s.q}
        s.t, s.ev, s.n = 0.0, [], 0
        s.idle_side, s.idle = nside, nworkers
        s.nw = nworkers
        s.reserve = reserve or [0] * s.nl       # workers kept free for levels < l when taking from level l
        s.potrf_p, s.potrf_busy = 0, False
        s.work = 0.0; s.traffic = 0.0
        s.potrf_wait = 0.0; s.potrf_free_at = 0.0

    def ready(s, qn, i):
        t = s.q[qn][i]
        if t[0] == 'trsm':
            _, p, J, h = t
            return s.diag_done[p] and s.seq.get((p, J), 0) == s.ntile.get((p, J), 0)
        if t[0] == 'upd':
            _, I, J, k1 = t
            if s.seq.get((I, J), 0) != s.ordinal[(qn, i)]: return False
            return min(s.solved[I] + s.solved[J]) >= k1
        _, I, q, k1 = t
        return s.seq.get((I, I), 0) == s.ordinal[(qn, i)] and min(s.solved[I]) >= k1

    def start(s, qn, i):
        t = s.q[qn][i]; c = s.c
        if t[0] == 'trsm': dur = c['trsm'] if qn == 'c' else c['trsm_w']
        elif t[0] == 'updq': dur = c['updq']
        else:
            _, I, J, k1 = t
            K = k1 - s.cnt.get((I, J), 0)
            assert K > 0
            fast = qn != 'c' and qn <= c['fast_levels']
            u = 1.0 - s.idle / s.nw
            dur = c['ovh'] + c['kalone'] * (1 + (c['share_u'] if fast else c['share']) * u) * K
            s.work += c['kblk'] * K
            s.traffic += 0.262 + 0.262 * K      # MB: S tile r+w, two operand panels
        heapq.heappush(s.ev, (s.t + dur + c['hop'], s.n, t, qn)); s.n += 1

    def complete(s, t):
        if t[0] == 'trsm': s.solved[t[2]][t[3]] = t[1] + 1
        elif t[0] == 'upd':
            s.cnt[(t[1], t[2])] = t[3]; s.seq[(t[1], t[2])] = s.seq.get((t[1], t[2]), 0) + 1
        elif t[0] == 'updq':
            I = t[1]; s.qd[I] = s.qd.get(I, 0) + 1
            if s.qd[I] == 3: s.cnt[(I, I)] = t[3]; s.seq[(I, I)] = s.seq.get((I, I), 0) + 1
        else:
            s.diag_done[t[1]] = True; s.potrf_busy = False; s.potrf_p += 1; s.potrf_free_at = s.t

    def dispatch(s):
        if not s.potrf_busy and s.potrf_p < s.nP:
            p = s.potrf_p
            if p == 0 or s.seq.get((p, p), 0) == s.ntile.get((p, p), 0):
                s.potrf_busy = True
                heapq.heappush(s.ev, (s.t + s.c['potrf'] + s.c['hop'], s.n, ('potrf', p), 'p')); s.n += 1
        while s.idle_side > 0 and s.heads['c'] < len(s.crit) and s.ready('c', s.heads['c']):
            s.start('c', s.heads['c']); s.heads['c'] += 1; s.idle_side -= 1
        for l in range(s.nl):
            q = s.gen[l]
            while s.idle > s.reserve[l] and s.heads[l] < len(q) and s.ready(l, s.heads[l]):
                s.start(l, s.heads[l]); s.heads[l] += 1; s.idle -= 1

    def run(s, trace=False):
        s.dispatch()
        while s.ev:
            tm, _, t, qn = heapq.heappop(s.ev)
            s.t = tm
            s.complete(t)
            if qn == 'c': s.idle_side += 1
            elif qn != 'p': s.idle += 1
            if trace and t[0] == 'potrf' and t[1] % trace == 0:
                print(f"  potrf {t[1]:3d} done {s.t:8.1f} us (ideal {(t[1]+1)*(s.c['potrf']+3*s.c['hop']+10)-3*s.c['hop']-10+s.c['hop']:8.1f})  heads " +
                      ' '.join(f"{s.heads[l]}/{len(s.gen[l])}" for l in range(s.nl)) + f" idle={s.idle}")
            s.dispatch()
        return s.t, all(s.diag_done)

def parse_D(spec, nP):
    # spec like "1,1,2,4*"  : chunk sizes from the pivot outwards, last one repeated
    parts = spec.split(',')
    sizes = []
    rep = None
    for x in parts:
        if x.endswith('*'): rep = int(x[:-1])
        else: sizes.append(int(x))
    D, d = [0], 0
    for sz in sizes: d += sz; D.append(d)
    while rep and d < nP: d += rep; D.append(d)
    return D

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--nP', type=int, default=64)
    ap.add_argument('--chunks', default='1,1,2,4*')
    ap.add_argument('--levels', default='0,2,4,8')       # distance thresholds: d<=0 -> level0, d<=2 -> 1 ...
    ap.add_argument('--fast', type=int, default=1)       # levels <= this run at the priority rate
    ap.add_argument('--workers', type=int, default=506)
    ap.add_argument('--side', type=int, default=4)
    ap.add_argument('--potrf', type=float, default=30)
    ap.add_argument('--hop', type=float, default=2.5)
    ap.add_argument('--kblk', type=float, default=32)
    ap.add_argument('--kalone', type=float, default=21)
    ap.add_argument('--share', type=float, default=0.5)
    ap.add_argument('--ovh', type=float, default=6)
    ap.add_argument('--trsm', type=float, default=18)
    ap.add_argument('--trsm_w', type=float, default=22)
    ap.add_argument('--updq', type=float, default=10)
    ap.add_argument('--share_u', type=float, default=0.25)
    ap.add_argument('--reserve', default='')
    ap.add_argument('--trace', type=int, default=0)
    ap.add_argument('--lazy', type=float, default=0)
    ap.add_argument('--lazymin', type=int, default=4)
    a = ap.parse_args()
    LAZY[0] = a.lazy; LAZY[1] = a.lazymin
    c = dict(potrf=a.potrf, hop=a.hop, trsm=a.trsm, trsm_w=a.trsm_w, updq=a.updq, ovh=a.ovh, kalone=a.kalone, share=a.share, kblk=a.kblk, share_u=a.share_u, fast_levels=a.fast)
    D = parse_D(a.chunks, a.nP)
    lv = [int(x) for x in a.levels.split(',')]
    res = [int(x) for x in a.reserve.split(',')] if a.reserve else None
    s = Sim(a.nP, D, lv, a.workers, a.side, c, res)
    t, ok = s.run(a.trace)
    flop = a.nP ** 3 / 3 * 128 ** 3 * 2 / 2
    print(f"nP={a.nP} chunks={a.chunks} levels={a.levels}: {t/1000:.3f} ms ok={ok} chain={a.nP*(a.potrf+3*a.hop+10)/1000:.3f} work={s.work/a.workers/1000:.3f} ms  "
          f"{flop/t/1e6:.1f} TFLOP/s traffic={s.traffic/1000:.1f} GB tasks={len(s.crit)}+" + '/'.join(str(len(g)) for g in s.gen))
