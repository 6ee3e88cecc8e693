"""Discrete-event model of the task-graph Cholesky (design aid for kernels_chol_tg.hip; not product code).

Three queues (crit: dedicated sidekick workgroups; urgent, far: general workers, urgent first), tickets taken only
when the head's dependencies are met (no worker ever blocks holding a ticket).  Costs in microseconds.
"""
import heapq, sys, argparse

def build(nP, Wn, Wk, far_order='panel'):
    # tile task lists: per tile ordinal counters
    nt = {}                                # (I,J) -> tasks so far
    def ordn(I, J):
        o = nt.get((I, J), 0); nt[(I, J)] = o + 1; return o
    crit, urgent, far = [], [], []
    far_by_panel = {}
    # far tasks are generated per step where a Wk-chunk completes; urgent per step
    # we must assign ordinals in dependency order per tile: far chunks ascending, then window tasks ascending
    # generate in step order p and remember
    for p in range(nP):
        # after potrf(p): trsm row p
        for J in range(p + 1, nP):
            for h in (0, 1):
                t = ('trsm', p, J, h)
                (crit if J == p + 1 else urgent).append(t)
        # window updates of step p: rows p+1..p+Wn to k1 = p+1
        for I in range(p + 1, min(p + Wn, nP - 1) + 1):
            for J in range(I, nP):
                if I == p + 1 and J == I:
                    for q in range(3):
                        crit.append(('updq', I, q, p + 1))
                else:
                    urgent.append(('upd', I, J, p + 1, p + 1))
        # far chunks, per-row alignment: row I's chunk boundaries end at I - Wn (so the window entry is always K = 1)
        lst = []
        for I in range(p + Wn + 1, nP):
            e = I - Wn                      # last far boundary of row I
            if (e - (p + 1)) % Wk == 0 and p + 1 <= e:
                # first boundary b1 >= Wk unless the row has fewer blocks
                if p + 1 < Wk and p + 1 != e: continue     # merged into the first (longer) chunk
                for J in range(I, nP):
                    lst.append(('upd', I, J, p + 1, e))
        far_by_panel[p + 1] = lst
    if far_order == 'panel':
        for k in sorted(far_by_panel):
            far.extend(far_by_panel[k])
    return crit, urgent, far

class Sim:
    def __init__(s, nP, Wn, Wk, nworkers, nside, nurg, c):
        s.nP = nP; s.c = c
        s.crit, s.urgent, s.far = build(nP, Wn, Wk)
        # ordinals
        s.seq_need = {}
        cnt = {}
        # assign ordinals in the global dependency-consistent order: far chunk k1 ascending then window; emulate by sorting per tile by k1
        per_tile = {}
        for qn, q in (('c', s.crit), ('u', s.urgent), ('f', s.far)):
            for i, t in enumerate(q):
                if t[0] == 'upd':
                    per_tile.setdefault((t[1], t[2]), []).append((t[3], qn, i))
                elif t[0] == 'updq':
                    per_tile.setdefault((t[1], t[1]), []).append((t[3], qn, i))
        s.ordinal = {}
        s.ntile = {}
        for tile, l in per_tile.items():
            l.sort()
            o = 0
            lastk = None
            for k1, qn, i in l:
                # the three quadrant tasks share an ordinal
                if lastk is not None and k1 == lastk and qn == 'c':
                    s.ordinal[(qn, i)] = o - 1
                else:
                    s.ordinal[(qn, i)] = o; o += 1
                lastk = k1
            s.ntile[tile] = o
        s.seq = {}          # tile -> completed count
        s.cnt = {}          # tile -> k applied
        s.qd = {}           # diag I -> quadrant completions
        s.solved = [[0, 0] for _ in range(nP)]
        s.diag_done = [False] * nP
        s.heads = {'c': 0, 'u': 0, 'f': 0}
        s.q = {'c': s.crit, 'u': s.urgent, 'f': s.far}
        s.t = 0.0
        s.ev = []
        s.idle = {'side': nside, 'gen': nworkers, 'urg': nurg}
        s.potrf_p = 0
        s.potrf_busy = False
        s.n = 0
        s.busy_time = 0.0
        s.stall = 0.0

    def ready(s, qn, i):
        t = s.q[qn][i]
        if t[0] == 'trsm':
            _, p, J, h = t
            return s.diag_done[p] and s.seq.get((p, J), 0) == s.ntile.get((p, J), 0)
        if t[0] == 'upd':
            _, I, J, k1, cap = t
            if s.seq.get((I, J), 0) != s.ordinal[(qn, i)]: return False
            return min(s.solved[I][0], s.solved[I][1], s.solved[J][0], s.solved[J][1]) >= k1
        if t[0] == 'updq':
            _, I, q, k1 = t
            if s.seq.get((I, I), 0) != s.ordinal[(qn, i)]: return False
            return min(s.solved[I]) >= k1

    def start(s, qn, i, prio):
        t = s.q[qn][i]; c = s.c
        if t[0] == 'trsm':
            dur = c['trsm']
        elif t[0] == 'updq':
            dur = c['updq']
        else:
            _, I, J, k1, cap = t
            k0 = s.cnt.get((I, J), 0)
            avail = min(s.solved[I][0], s.solved[I][1], s.solved[J][0], s.solved[J][1], I)
            khi = max(k1, min(cap, avail))
            if qn == 'f' and c.get('merge_round'):
                khi = max(k1, (khi // c['merge_round']) * c['merge_round'])
            K = khi - k0
            t = ('upd', I, J, khi, cap)
            if K <= 0: dur = 0.5
            else:
                dur = c['ovh'] + (c['kblk_u'] if prio else c['kblk']) * K
                s.busy_time += (c['kblk']) * K
        heapq.heappush(s.ev, (s.t + dur + c['hop'], s.n, 'done', t, qn)); s.n += 1

    def complete(s, t):
        if t[0] == 'trsm':
            _, p, J, h = t; s.solved[J][h] = p + 1
        elif t[0] == 'upd':
            _, I, J, khi, cap = t
            s.cnt[(I, J)] = max(khi, s.cnt.get((I, J), 0)); s.seq[(I, J)] = s.seq.get((I, J), 0) + 1
        elif t[0] == 'updq':
            _, I, q, k1 = t
            s.qd[I] = s.qd.get(I, 0) + 1
            if s.qd[I] == 3:
                s.cnt[(I, I)] = k1; s.seq[(I, I)] = s.seq.get((I, I), 0) + 1
        elif t[0] == 'potrf':
            s.diag_done[t[1]] = True; s.potrf_busy = False; s.potrf_p += 1

    def dispatch(s):
        # critical WG
        if not s.potrf_busy and s.potrf_p < s.nP:
            p = s.potrf_p
            if p == 0 or s.seq.get((p, p), 0) == s.ntile.get((p, p), 0):
                s.potrf_busy = True
                heapq.heappush(s.ev, (s.t + s.c['potrf'] + s.c['hop'], s.n, 'done', ('potrf', p), 'p')); s.n += 1
        # sidekicks
        while s.idle['side'] > 0 and s.heads['c'] < len(s.crit) and s.ready('c', s.heads['c']):
            s.start('c', s.heads['c'], True); s.heads['c'] += 1; s.idle['side'] -= 1
        # urgent-only workers then general
        for pool in ('urg', 'gen'):
            while s.idle[pool] > 0 and s.heads['u'] < len(s.urgent) and s.ready('u', s.heads['u']):
                s.start('u', s.heads['u'], True); s.heads['u'] += 1; s.idle[pool] -= 1
        while s.idle['gen'] > 0 and s.heads['f'] < len(s.far) and s.ready('f', s.heads['f']):
            s.start('f', s.heads['f'], False); s.heads['f'] += 1; s.idle['gen'] -= 1

    def run(s, trace=False):
        pool_of = {}
        s.dispatch()
        last_potrf_end = 0
        while s.ev:
            tm, _, kind, t, qn = heapq.heappop(s.ev)
            s.t = tm
            s.complete(t)
            if qn == 'c': s.idle['side'] += 1
            elif qn == 'u' or qn == 'f':
                # return to whichever pool has deficit (approximation: urgent-only first)
                if s.idle['urg'] < s.c['nurg'] and qn == 'u' and s.c['nurg'] > 0 and s.urg_out > 0:
                    s.idle['urg'] += 1; s.urg_out -= 1
                else: s.idle['gen'] += 1
            if trace and t[0] == 'potrf' and t[1] % 8 == 0:
                print(f"  potrf {t[1]:3d} done at {s.t:8.1f} us  heads u={s.heads['u']}/{len(s.urgent)} f={s.heads['f']}/{len(s.far)}")
            s.dispatch()
        done = all(s.diag_done)
        return s.t, done

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--nP', type=int, default=64)
    ap.add_argument('--Wn', type=int, default=3)
    ap.add_argument('--Wk', type=int, default=4)
    ap.add_argument('--workers', type=int, default=506)
    ap.add_argument('--side', type=int, default=4)
    ap.add_argument('--potrf', type=float, default=30)
    ap.add_argument('--hop', type=float, default=2.5)
    ap.add_argument('--kblk', type=float, default=32)
    ap.add_argument('--kblk_u', type=float, default=18)
    ap.add_argument('--trace', action='store_true')
    a = ap.parse_args()
    c = dict(potrf=a.potrf, hop=a.hop, trsm=5, updq=5, ovh=4, kblk=a.kblk, kblk_u=a.kblk_u, nurg=0)
    s = Sim(a.nP, a.Wn, a.Wk, a.workers, a.side, 0, c); s.urg_out = 0
    t, ok = s.run(a.trace)
    flop = a.nP ** 3 / 3 * 128 ** 3
    print(f"nP={a.nP} Wn={a.Wn} Wk={a.Wk}: {t/1000:.3f} ms ok={ok} chain={a.nP*(a.potrf+3*a.hop+10)/1000:.3f} ms  work={s.busy_time/a.workers/1000:.3f} ms  {flop/t/1e6:.1f} TFLOP/s  tasks c/u/f={len(s.crit)}/{len(s.urgent)}/{len(s.far)}")

def debug(nP=16, Wn=2, Wk=4):
    c = dict(potrf=30, hop=2.5, trsm=5, updq=5, ovh=4, kblk=32, kblk_u=18, nurg=0)
    s = Sim(nP, Wn, Wk, 506, 4, 0, c); s.urg_out = 0
    log = []
    orig = s.complete
    def comp(t):
        orig(t)
        if t[0] == 'potrf' or (t[0] == 'upd' and t[2] - t[1] <= 1) or t[0] == 'updq' or (t[0]=='trsm' and t[2]-t[1] <= 2):
            log.append((round(s.t,1), t))
    s.complete = comp
    s.run()
    for l in log[:120]: print(l)
