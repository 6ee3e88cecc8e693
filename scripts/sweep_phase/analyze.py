"""Phase analysis of a sweep_probe.bin stamp dump.

For every compute unit: the workgroups that lived there, the windows in which wave 0 of each was NOT in its matrix phase
(end of the 128 MFMAs -> barrier -> LDS write -> barrier -> next step), how much of those windows the co-resident
workgroup spent outside ITS matrix phase too (= the matrix pipe of that SIMD had nobody to serve), and the phase offset
between the two workgroups' k-steps.

usage: analyze.py dump.bin [max_cus]
"""
import sys
import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint32)
    n, maxstep = int(raw[0]), int(raw[1])
    rec = 1 + 8 + maxstep * 4
    body = raw[4:4 + n * rec].reshape(n, rec)
    blk = body[:, 0]
    meta = body[:, 1:9]
    st = body[:, 9:].reshape(n, maxstep, 4)
    return blk, meta, st


def unwrap(lo, ref):
    """32-bit stamps -> 64-bit, relative to the 64-bit reference that precedes them by less than 2^32."""
    lo = lo.astype(np.int64)
    d = (lo - (ref & 0xffffffff)) & 0xffffffff
    return ref + d


def main():
    blk, meta, st = load(sys.argv[1])
    maxcu = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
    hw, xcc, arr = meta[:, 0], meta[:, 1], meta[:, 2]
    t0 = meta[:, 3].astype(np.int64) | (meta[:, 4].astype(np.int64) << 32)
    t1 = meta[:, 5].astype(np.int64) | (meta[:, 6].astype(np.int64) << 32)
    nst = meta[:, 7].astype(np.int64)
    key = (xcc.astype(np.int64) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    simd = (hw >> 4) & 3
    print('workgroups %d, compute units %d, steps per workgroup: min %d max %d' % (len(blk), len(np.unique(key)), nst.min(), nst.max()))
    print('wave 0 lands on SIMD: %s' % np.bincount(simd, minlength=4))
    life = (t1 - t0)
    print('workgroup lifetime (clocks): mean %.0f  min %d  max %d  -> per k-step %.0f' % (life.mean(), life.min(), life.max(), (life / nst).mean()))

    # per workgroup 64-bit stamps
    T = np.zeros(st.shape, dtype=np.int64)
    for i in range(len(blk)):
        T[i] = unwrap(st[i].reshape(-1), t0[i]).reshape(-1, 4)
    # durations inside a workgroup
    mat, bar1, wr, per = [], [], [], []
    for i in range(len(blk)):
        n = nst[i]
        s = T[i, :n]
        mat.append(s[:, 1] - s[:, 0])
        bar1.append(s[:, 2] - s[:, 1])
        wr.append(s[:, 3] - s[:, 2])
        per.append(np.diff(s[:, 0]))
    mat, bar1, wr, per = map(np.concatenate, (mat, bar1, wr, per))

    def q(x):
        return 'mean %7.0f  p10 %6.0f  p50 %6.0f  p90 %6.0f  p99 %6.0f' % ((x.mean(),) + tuple(np.percentile(x, [10, 50, 90, 99])))
    print('per k-step, wave 0, in clocks of s_memtime:')
    print('  step period                    %s' % q(per))
    print('  matrix phase (loads + 128 MFMA) %s' % q(mat))
    print('  wait at first barrier           %s' % q(bar1))
    print('  LDS write + second barrier      %s' % q(wr))
    nonmat = bar1 + wr
    print('  not in matrix phase             %s   = %.2f %% of the period' % (q(nonmat), 100.0 * nonmat.sum() / (mat.sum() + nonmat.sum())))

    # co-residency: per CU, sweep the timeline
    both_out = 0
    total = 0
    any_out = 0
    offs = []
    pairs = 0
    ncu = 0
    solo = 0
    for k in np.unique(key)[:maxcu]:
        idx = np.where(key == k)[0]
        idx = idx[np.argsort(t0[idx])]
        ncu += 1
        # events: (time, +1/-1 "a workgroup leaves / re-enters its matrix phase"), resident intervals
        ev = []
        for i in idx:
            n = nst[i]
            s = T[i, :n]
            ev.append(np.stack([np.full(1, t0[i]), np.full(1, 2)], 1))          # resident from
            ev.append(np.stack([np.full(1, t1[i]), np.full(1, -2)], 1))         # resident until
            ev.append(np.stack([s[:, 1], np.full(n, 1)], 1))                    # leaves the matrix phase
            ev.append(np.stack([s[:, 3], np.full(n, -1)], 1))                   # back in it
        ev = np.concatenate(ev)
        ev = ev[np.argsort(ev[:, 0], kind='stable')]
        res = 0
        out = 0
        # before its first step a workgroup is outside the matrix phase, but that prologue is short; ignore
        tprev = ev[0, 0]
        for tt, kind in ev:
            dt = tt - tprev
            if res == 2:
                total += dt
                if out >= 2:
                    both_out += dt
                if out >= 1:
                    any_out += dt
            elif res == 1:
                solo += dt
            tprev = tt
            if kind == 2:
                res += 1
            elif kind == -2:
                res -= 1
            elif kind == 1:
                out += 1
            else:
                out -= 1
        # phase offsets between co-resident workgroups: start of B's steps relative to A's steps
        for a in range(len(idx)):
            for b in range(a + 1, len(idx)):
                ia, ib = idx[a], idx[b]
                lo, hi = max(t0[ia], t0[ib]), min(t1[ia], t1[ib])
                if hi - lo < 0.5 * min(life[ia], life[ib]):
                    continue
                pairs += 1
                sa = T[ia, :nst[ia], 0]
                sb = T[ib, :nst[ib], 0]
                sb = sb[(sb > lo) & (sb < hi)]
                j = np.searchsorted(sa, sb) - 1
                ok = (j >= 0) & (j + 1 < len(sa))
                ph = (sb[ok] - sa[j[ok]]) / (sa[j[ok] + 1] - sa[j[ok]])
                offs.append(ph)
    offs = np.concatenate(offs) if offs else np.zeros(0)
    print('compute units analysed %d, co-resident pairs %d' % (ncu, pairs))
    print('time with two resident workgroups: %.3e clocks; with one: %.3e' % (total, solo))
    print('  wave 0 of AT LEAST ONE outside its matrix phase: %.2f %%' % (100.0 * any_out / total))
    print('  wave 0 of BOTH outside their matrix phase at once: %.2f %%  <- nobody on that SIMD issues MFMA' % (100.0 * both_out / total))
    ind = (any_out / total / 2.0) ** 2 if total else 0
    print('  (if the two were independent: %.2f %%)' % (100.0 * (nonmat.sum() / (mat.sum() + nonmat.sum())) ** 2))
    if len(offs):
        h, _ = np.histogram(offs, bins=10, range=(0, 1))
        print('phase of B\'s step start inside A\'s step (0 = in phase, 0.5 = anti-phase), deciles: %s' % (np.round(h / h.sum(), 3)))


if __name__ == '__main__':
    main()
