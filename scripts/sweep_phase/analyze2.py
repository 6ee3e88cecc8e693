"""Attribution of the matrix pipe's idle clocks from a sweep_probe.bin dump with 12 stamps per k-step (slot 0: top of
step, 1..8: wave 0 has issued its k-th group of 16 MFMAs, 9: past the first barrier, 10: past LDS write + second barrier).

A SIMD hosts one wave of each of the CU's two workgroups; per step period it must issue 2 x 128 MFMAs x 64 clocks = 16384
clocks of matrix work.  Every group interval of a workgroup is classified by how much of it the OTHER workgroup of the CU
spent outside its matrix phase (slot 8 -> slot 10 of its own steps): 'alone' (>= 99 %: the group had the pipe to itself,
ideal 1024 clocks per 16 MFMAs), 'shared' (0 %: ideal 2048), 'mixed'.  usage: analyze2.py dump.bin [max_cus]
"""
import sys
import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint32)
    n, maxstep, nslot = int(raw[0]), int(raw[1]), int(raw[2])
    rec = 1 + 8 + maxstep * nslot
    body = raw[4:4 + n * rec].reshape(n, rec)
    return body[:, 0], body[:, 1:9], body[:, 9:].reshape(n, maxstep, nslot)


def main():
    blk, meta, st = load(sys.argv[1])
    maxcu = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
    hw, xcc = meta[:, 0], meta[:, 1]
    t0 = meta[:, 3].astype(np.int64) | (meta[:, 4].astype(np.int64) << 32)
    t1 = meta[:, 5].astype(np.int64) | (meta[:, 6].astype(np.int64) << 32)
    nst = meta[:, 7].astype(np.int64)
    key = (xcc.astype(np.int64) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    n = int(nst.max())
    T = ((st[:, :n, :11].astype(np.int64) - (t0[:, None, None] & 0xffffffff)) & 0xffffffff) + t0[:, None, None]
    life = t1 - t0
    per = np.diff(T[:, :, 0], axis=1)
    print('workgroups %d on %d compute units; %d k-steps each; lifetime %.0f clocks = %.0f per step; step period p50 %.0f mean %.0f'
          % (len(blk), len(np.unique(key)), n, life.mean(), (life / nst).mean(), np.median(per), per.mean()))
    grp = np.diff(T[:, :, 0:9], axis=2)              # (wg, step, 8): clocks per group of 16 MFMAs (group 1 includes the global-load issue)
    out = T[:, :, 10] - T[:, :, 8]                    # outside the matrix phase
    b1 = T[:, :, 9] - T[:, :, 8]
    print('outside the matrix phase per step: mean %.0f (first barrier %.0f, LDS write + second barrier %.0f) = %.2f %% of the period'
          % (out.mean(), b1.mean(), (out - b1).mean(), 100 * out.mean() / per.mean()))
    alone, shared, mixed = [[] for _ in range(8)], [[] for _ in range(8)], [[] for _ in range(8)]
    idle_alone = idle_shared = idle_mixed = 0.0
    tot = 0.0
    nsteps = 0
    for k in np.unique(key)[:maxcu]:
        idx = np.where(key == k)[0]
        for i in idx:
            others = [j for j in idx if j != i and min(t1[i], t1[j]) - max(t0[i], t0[j]) > 0]
            if not others:
                continue
            ws = np.concatenate([T[j, :, 8] for j in others])
            we = np.concatenate([T[j, :, 10] for j in others])
            o = np.argsort(ws)
            ws, we = ws[o], we[o]
            cum = np.concatenate([[0], np.cumsum(we - ws)])

            def cov(t):
                ii = np.searchsorted(ws, t, 'right') - 1
                c = cum[np.maximum(ii, 0)] + np.clip(np.minimum(t, we[np.maximum(ii, 0)]) - ws[np.maximum(ii, 0)], 0, None)
                return np.where(ii >= 0, c, 0)
            # steps of i while a partner is resident for the whole step
            lo = max(t0[j] for j in others if True)
            x0, x1 = T[i, :, 0:8], T[i, :, 1:9]
            f = (cov(x1) - cov(x0)) / np.maximum(x1 - x0, 1)
            resident = np.zeros(n, bool)
            for j in others:
                resident |= (T[i, :, 0] > T[j, 0, 0]) & (T[i, :, 10] < T[j, -1, 10])
            g = grp[i]
            for kk in range(8):
                a = resident & (f[:, kk] >= 0.99)
                s = resident & (f[:, kk] <= 0.0)
                m = resident & ~a & ~s
                alone[kk].append(g[a, kk]); shared[kk].append(g[s, kk]); mixed[kk].append(g[m, kk])
    print('clocks per group of 16 MFMAs of wave 0 (ideal: alone 1024, shared 2048); group 1 also issues the 16 global loads and waits for its first LDS reads')
    print('  group      alone: n   mean   p50   p90 |    shared: n   mean   p50   p90 |   mixed: n   mean')
    tot_al = tot_sh = 0
    for kk in range(8):
        a, s, m = np.concatenate(alone[kk]), np.concatenate(shared[kk]), np.concatenate(mixed[kk])
        print('  %d     %9d %6.0f %5.0f %5.0f |  %9d %6.0f %5.0f %5.0f |  %9d %6.0f' % (
            kk + 1, len(a), a.mean() if len(a) else 0, np.median(a) if len(a) else 0, np.percentile(a, 90) if len(a) else 0,
            len(s), s.mean() if len(s) else 0, np.median(s) if len(s) else 0, np.percentile(s, 90) if len(s) else 0, len(m), m.mean() if len(m) else 0))
    A = np.concatenate([np.concatenate(x) for x in alone]); S = np.concatenate([np.concatenate(x) for x in shared]); Mx = np.concatenate([np.concatenate(x) for x in mixed])
    ng = len(A) + len(S) + len(Mx)
    print('share of groups: alone %.1f %%, shared %.1f %%, mixed %.1f %%' % (100 * len(A) / ng, 100 * len(S) / ng, 100 * len(Mx) / ng))
    print('excess over the ideal per group: alone %+.0f clocks (x %.1f %% of groups), shared %+.0f (half of it is this wave\'s), mixed: see below'
          % (A.mean() - 1024, 100 * len(A) / ng, S.mean() - 2048))
    # per-step budget on one SIMD: period = 16384 + idle.  While a wave runs alone every excess clock is an idle pipe clock;
    # while two share, the pipe idles (excess of both) / 2 per group pair -> per group of this wave: excess / 2... the pair's
    # groups overlap in time, so idle per clock = 1 - 2048 / mean.
    p = per.mean()
    ia = (A.mean() - 1024) * 8 * len(A) / ng
    isn = (1 - 2048.0 / S.mean()) * S.mean() * 8 * len(S) / ng
    print('idle pipe clocks per step period (%.0f - 16384 = %.0f): while one wave runs alone %.0f per workgroup -> x2 = %.0f; while both share %.0f x2/2 = %.0f; rest (mixed groups, tile prologue/epilogue) %.0f'
          % (p, p - 16384, ia, 2 * ia, isn, isn, p - 16384 - 2 * ia - isn))


if __name__ == '__main__':
    main()
