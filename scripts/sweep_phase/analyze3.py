"""Per-SIMD timeline from a sweep_probe.bin dump with stamps of ALL four waves of every workgroup.
usage: analyze3.py dump.bin [cu_index=0] [periods=2]
Prints, for every SIMD of one compute unit, what its (two) resident waves did over a few step periods in the middle of the
launch, and chip-wide (all dumped CUs) the pipe accounting per SIMD: clocks in which wave X is between its first and last
MFMA group of a step ('issuing'), overlap of the two waves' issuing windows, and gaps in which neither is.
"""
import sys
import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint32)
    n, maxstep, nslot, nw = [int(x) for x in raw[:4]]
    rec = 1 + 8 + nw + nw * maxstep * nslot
    body = raw[4:4 + n * rec].reshape(n, rec)
    return body[:, 0], body[:, 1:9], body[:, 9:9 + nw], body[:, 9 + nw:].reshape(n, nw, maxstep, nslot)


def main():
    blk, meta, whw, st = load(sys.argv[1])
    cu_i = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nper = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    hw, xcc = meta[:, 0], meta[:, 1]
    t0 = meta[:, 3].astype(np.int64) | (meta[:, 4].astype(np.int64) << 32)
    t1 = meta[:, 5].astype(np.int64) | (meta[:, 6].astype(np.int64) << 32)
    n = int(meta[:, 7].max())
    key = (xcc.astype(np.int64) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    T = ((st[:, :, :n, :11].astype(np.int64) - (t0[:, None, None, None] & 0xffffffff)) & 0xffffffff) + t0[:, None, None, None]
    simd = (whw >> 4) & 3
    print('workgroups %d, CUs %d; waves of a workgroup on distinct SIMDs in %.1f %% of the workgroups'
          % (len(blk), len(np.unique(key)), 100.0 * np.mean([len(set(s)) == 4 for s in simd])))
    keys = np.unique(key)
    # ---- timeline of one CU
    k = keys[cu_i]
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(t0[idx])]
    mid = len(idx) // 2
    a, b = idx[mid], idx[mid + 1]
    lo = max(t0[a], t0[b])
    hi = min(t1[a], t1[b])
    if hi - lo < 10 * 17000:
        a, b = idx[mid - 1], idx[mid]
        lo, hi = max(t0[a], t0[b]), min(t1[a], t1[b])
    tref = T[a, 0, n // 2, 0]
    tend = T[a, 0, n // 2 + nper, 0]
    names = {0: 'top', 1: 'g1', 2: 'g2', 3: 'g3', 4: 'g4', 5: 'g5', 6: 'g6', 7: 'g7', 8: 'g8(end mfma)', 9: 'past barrier 1', 10: 'past write+barrier 2'}
    print('CU key %#x: workgroups %d (A) and %d (B), window of %d periods of A' % (k, blk[a], blk[b], nper))
    for sd in range(4):
        ev = []
        for tag, wg in (('A', a), ('B', b)):
            for w in range(4):
                if simd[wg, w] != sd:
                    continue
                tt = T[wg, w]
                for stp in range(n):
                    for sl in range(11):
                        if tref <= tt[stp, sl] <= tend:
                            ev.append((tt[stp, sl] - tref, tag, w, stp, sl))
        ev.sort()
        print(' SIMD %d' % sd)
        prev = 0
        for (t, tag, w, stp, sl) in ev:
            print('   %7d (+%5d)  %s w%d step %3d  %s' % (t, t - prev, tag, w, stp, names[sl]))
            prev = t
    # ---- accounting over all dumped CUs: per SIMD and pair of co-resident waves
    tot = iss1 = iss2 = none = 0
    gtime = []
    for k in keys:
        idx = np.where(key == k)[0]
        for ii in range(len(idx)):
            for jj in range(ii + 1, len(idx)):
                a, b = idx[ii], idx[jj]
                lo, hi = max(T[a, :, 0, 0].max(), T[b, :, 0, 0].max()), min(T[a, :, -1, 10].min(), T[b, :, -1, 10].min())
                if hi - lo < 50 * 17000:
                    continue
                for sd in range(4):
                    wa = np.where(simd[a] == sd)[0]
                    wb = np.where(simd[b] == sd)[0]
                    if len(wa) != 1 or len(wb) != 1:
                        continue
                    ev = []
                    for wg, w in ((a, wa[0]), (b, wb[0])):
                        s0 = T[wg, w, :, 1]      # first group issued: the wave has the pipe
                        s1 = T[wg, w, :, 8]
                        m = (s0 > lo) & (s1 < hi)
                        ev += [(x, 1) for x in s0[m]] + [(x, -1) for x in s1[m]]
                        gtime.append(np.diff(T[wg, w, :, 1:9], axis=1)[m].reshape(-1))
                    ev.sort()
                    c = 0
                    tp = ev[0][0]
                    for x, dlt in ev:
                        d = x - tp
                        tot += d
                        if c == 0: none += d
                        elif c == 1: iss1 += d
                        else: iss2 += d
                        tp = x
                        c += dlt
    g = np.concatenate(gtime)
    print('per SIMD (pairs of co-resident waves, %.2e clocks): exactly one wave between its first and last group %.1f %%, both %.1f %%, neither %.1f %%'
          % (tot, 100.0 * iss1 / tot, 100.0 * iss2 / tot, 100.0 * none / tot))
    print('clocks per group of 16 MFMAs (groups 2..8): mean %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f' % ((g.mean(),) + tuple(np.percentile(g, [10, 50, 90, 99]))))


if __name__ == '__main__':
    main()
