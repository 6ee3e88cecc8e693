#!/usr/bin/env python
"""
bench.py -- BO-step wall-clock of the MI355X engine: GP fit + 1e6-candidate acquisition sweep.

One "step" = what one iteration of pybo's loop asks of the model for the configured policy
(/root/reference/pybo/bayesopt.py:262-269):
    model.add_data        -> Gram build + Cholesky + triangular inverse + alpha      (gpx_fit_dev)
    policy(model, ., X)   -> target = max posterior mean at the data (+xi)          (gpx_mean_at_obs)
    solver(index, bounds) -> index on the whole candidate grid + top-k               (gpx_sweep_dev)
    [N > 1 ranks]         -> all-gather of the per-rank (value, global index) top-k   (RCCL)
Inputs (observations, candidates) are resident in HBM before the timed region starts; every rank
refits redundantly (bitwise-identical factor, zero communication) and sweeps its contiguous candidate
slice, so the job is STRONG scaling: total work fixed at M candidates.

Launch:  python bench.py [--gpus 1 --steps K --warmup W --workload ns|b|c|d|e]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# (pybo_amd/_lib.py sets the same default on import; here it has to happen before `import torch` can initialise the runtime:
#  bench.py keeps its own engine alive beside the plug-in layer's handles -- see the comment there)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix (= fp64 vector) peak, dense
HBM_PEAK_GBS = 8000.0
SWEEP_KERNEL = 'k_sweep_trmm_l<32, 2, 1, 2, true>'      # the default schedule from 32 block rows on (tile_order 19) as rocprofv3 names it
SWEEP_KERNEL_SHORT = 'k_sweep_trmm_w'                    # ... and below (tile_order 7: config B)
PMC_TRAFFIC_FILES = ('r06_pmc_traffic.json', 'r06_pmc_traffic_b.json')      # (N = 8192 and config B's N = 2048 launch geometries)


def hartmann6(X):
    A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14], [3, 3.5, 1.7, 10, 17, 8],
                  [17, 8, 0.05, 10, 0.1, 14]])
    P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886], [2329, 4135, 8307, 3736, 1004, 9991],
                         [2348, 1451, 3522, 2883, 3047, 6650], [4047, 8828, 8732, 5743, 1091, 381]])
    c = np.array([1.0, 1.2, 3.0, 3.2])
    return -(c * np.exp(-(A[None] * (X[:, None, :] - P[None]) ** 2).sum(-1))).sum(-1)


def branin(X):
    a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5.0 / np.pi, 6.0, 10.0, 1.0 / (8 * np.pi)
    return a * (X[:, 1] - b * X[:, 0] ** 2 + c * X[:, 0] - r) ** 2 + s * (1 - t) * np.cos(X[:, 0]) + s


def make_workload(name, M):
    """Synthetic inputs of SURVEY.md 8(d): identical arrays feed the GPU path and the CPU baseline."""
    from scipy.stats import qmc
    if name == 'ns':      # north-star target: N=8192, d=8, SE-ARD, EI
        N, d, seed, kernel, acq = 8192, 8, 2, 'se', 'ei'
        lo, hi = np.zeros(d), np.ones(d)
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        f = lambda Z: -((Z - 0.5) ** 2).sum(1)                       # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'north-star: d=8 synthetic quadratic, N=8192 observed, SE-ARD, EI over 2^20 Sobol candidates'
    elif name == 'ns2':   # the north-star workload with its optimum moved off the Sobol' centre (0.5, ..): the winner of 'ns'
        # is Sobol' point 1, which any sane implementation picks; here the selection has to be earned.  A second, recorded
        # run (profiles/); the headline stays 'ns'.
        N, d, seed, kernel, acq = 8192, 8, 2, 'se', 'ei'
        lo, hi = np.zeros(d), np.ones(d)
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        opt = np.array([0.37, 0.61, 0.43, 0.58, 0.29, 0.66, 0.52, 0.41])
        f = lambda Z: -((Z - opt) ** 2).sum(1)                       # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'north-star variant: d=8 quadratic with its optimum off the grid centre, N=8192, SE-ARD, EI over 2^20 Sobol candidates'
    elif name == 'b':     # BASELINE configs[1]
        N, d, seed, kernel, acq = 2048, 2, 0, 'se', 'ei'
        lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        f = lambda Z: -branin(Z) / 10.0                               # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'config B: Branin d=2, N=2048 observed, SE-ARD, EI over 2^20 Sobol candidates'
    elif name == 'c':     # BASELINE configs[2]
        N, d, seed, kernel, acq = 8192, 6, 1, 'matern5', 'ucb'
        lo, hi = np.zeros(d), np.ones(d)
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        f = lambda Z: -hartmann6(Z)                                   # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'config C: Hartmann-6, N=8192 observed, Matern-5/2, UCB over 2^20 Sobol candidates'
    elif name == 'd':     # BASELINE configs[3]
        N, d, seed, kernel, acq = 16384, 32, 3, 'se', 'thompson'
        lo, hi = -np.ones(d), np.ones(d)
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        f = lambda Z: -(Z ** 2).sum(1)                                # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'config D: d=32 quadratic, N=16384 observed, SE-ARD, Thompson (64 RFF draws x 100 features)'
    elif name == 'e':     # BASELINE configs[4]: batch-BO q=8, one Thompson recommendation per GPU
        N, d, seed, kernel, acq = 8192, 6, 1, 'matern5', 'thompson'
        lo, hi = np.zeros(d), np.ones(d)
        rng = np.random.RandomState(seed)
        X = lo + (hi - lo) * rng.rand(N, d)
        f = lambda Z: -hartmann6(Z)                                   # noqa: E731
        y = f(X) + 1e-3 * rng.randn(N)
        desc = 'config E: batch-BO q=8 on Hartmann-6, N=8192, Matern-5/2, 8 Thompson draws (100 RFF), one per GPU'
    else:
        raise SystemExit('unknown workload %r' % name)
    ell = 0.25 * (hi - lo)
    rho, bias = float(np.var(y)), float(np.mean(y))
    sn2 = 1e-4 * rho
    Xc = lo + (hi - lo) * qmc.Sobol(d, scramble=False).random(M)
    return dict(name=name, desc=desc, N=N, d=d, M=M, kernel=kernel, acq=acq, X=X, y=y, ell=ell, rho=rho,
                sn2=sn2, bias=bias, Xc=Xc, f=f, lo=lo, hi=hi)


def ucb_beta(nobs, delta=0.1, xi=0.2):
    """pybo/policies/simple.py:58-66 -- `d` is the number of observations."""
    return xi * 2 * np.log(np.pi ** 2 / 3 / delta) + xi * (4 + nobs) * np.log(nobs + 1)


def sample_indices(nc, span):
    """The candidates the CPU baseline is timed on: TWO disjoint runs of nc/2 candidates, at the head and at the tail
    of the first `span` candidates (rank 0's shard) -- timed separately, so that the bench line shows how linear the
    extrapolation to the whole grid is."""
    half = max(nc // 2, 1)
    a = np.arange(0, min(half, span))
    b = np.arange(max(span - half, len(a)), span) if nc > 1 else np.arange(0)
    return a, b


def thompson_draw(w, s):
    """Host draws of Thompson sample s (order fixed: randn(n, d), [chisquare], rand(n), randn(n) -- GP.sample_f)."""
    nu = {'matern5': 2.5, 'matern3': 1.5, 'matern1': 0.5}.get(w['kernel'])
    rng = np.random.RandomState(100 + s)
    Wd = rng.randn(100, w['d'])
    if nu is not None:                        # Matern spectral density = scale mixture of normals
        Wd = Wd * np.sqrt(2.0 * nu / rng.chisquare(2.0 * nu, size=100))[:, None]
    return Wd / w['ell'], rng.rand(100) * 2 * np.pi, rng.randn(100)


def cpu_baseline(w, budget_candidates, span, draws, topk_idx=()):
    """Time the CPU restatement (oracle/, numpy+scipy on the host's BLAS threads) on a bounded sample: the fit in full,
    the sweep on `budget_candidates` of the M candidates in two disjoint halves (sample_indices), extrapolated linearly.
    Thompson workloads: every one of the step's `draws` posterior samples (sample_f + its values on the sample).
    Returns (the cpu_baseline record, the oracle's values on the sample for the parity record)."""
    from oracle import gp_ref
    threads = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        pass
    t0 = time.perf_counter()
    ref = gp_ref.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], w['kernel'])
    ref.add_data(w['X'], w['y'])
    t_fit = time.perf_counter() - t0
    ia, ib = sample_indices(budget_candidates, span)
    vals = {'index': np.concatenate([ia, ib])}
    parts, times, t_draws = [], [], 0.0
    samples = None
    if w['acq'] == 'thompson':
        t0 = time.perf_counter()
        samples = [ref.sample_f(100, rng=100 + s) for s in draws]
        t_draws = time.perf_counter() - t0
    for idx in (ia, ib):
        if len(idx) == 0:
            continue
        Z = w['Xc'][idx]
        t0 = time.perf_counter()
        if w['acq'] in ('ei', 'ucb'):
            mu, s2 = ref.predict(Z)
            if w['acq'] == 'ei':                # the arithmetic of GPRef.get_improvement on the moments just formed
                target = ref.mean_at_obs().max()
                s = np.sqrt(s2)
                z = (mu - target) / s
                v = (mu - target) * gp_ref.norm_cdf(z) + s * gp_ref.norm_pdf(z)
                vals['target'] = float(target)
            else:
                v = mu + np.sqrt(ucb_beta(w['N']) * s2)
            parts.append(dict(mu=mu, s2=s2, acq=v))
        else:
            parts.append(dict(acq=np.array([smp.get(Z) for smp in samples])))       # (draws, len(Z))
        times.append(time.perf_counter() - t0)
    for key in parts[0]:
        vals[key] = np.concatenate([p_[key] for p_ in parts], axis=-1)
    vals['best'] = np.argmax(vals['acq'], axis=-1)
    if len(topk_idx) and w['acq'] in ('ei', 'ucb'):
        # NOT timed: the oracle at the candidates the DEVICE ranked best over the whole grid (global indices), for the
        # order check of the parity record
        ti = np.asarray(topk_idx, dtype=np.int64)
        mu, s2 = ref.predict(w['Xc'][ti])
        if w['acq'] == 'ei':
            target = ref.mean_at_obs().max()
            s_ = np.sqrt(s2)
            z_ = (mu - target) / s_
            v = (mu - target) * gp_ref.norm_cdf(z_) + s_ * gp_ref.norm_pdf(z_)
        else:
            v = mu + np.sqrt(ucb_beta(w['N']) * s2)
        vals['topk'] = dict(index=ti, mu=mu, s2=s2, acq=v)
    nsamp = len(vals['index'])
    t_sw = float(sum(times))
    step = t_fit + t_draws + t_sw * (w['M'] / float(nsamp))
    how = 'in full' if nsamp == w['M'] else 'extrapolated linearly to M'
    per_cand = [t / max(len(i), 1) for t, i in zip(times, (ia, ib))]
    spread = (max(per_cand) - min(per_cand)) / (sum(per_cand) / len(per_cand)) if len(per_cand) > 1 else 0.0
    rec = dict(value=1.0 / step, unit='steps/s', cores=int(threads), kind='port',
               sample='fit in full (N=%d: %.2f s)%s + sweep on %d of %d candidates in two disjoint runs (%s s), sweep %s; '
                      'numpy/scipy on %d BLAS threads; host has %d logical cpus, %d in affinity'
                      % (w['N'], t_fit, (' + %d posterior samples (%.2f s)' % (len(draws), t_draws)) if samples else '',
                         nsamp, w['M'], ' + '.join('%.2f' % t for t in times), how, threads, os.cpu_count(),
                         len(os.sched_getaffinity(0))),
               sample_frac=nsamp / float(w['M']), sample_candidates=int(nsamp),
               seconds_per_candidate_run=[float(v) for v in per_cand], linearity_spread=float(spread),
               extrapolated=nsamp != w['M'], seconds_per_step=step)
    return rec, vals


def cpu_baseline_unpinned(workload, M, nc, span, draws, topk_idx=()):
    """torch.distributed.run pins OMP_NUM_THREADS=1 in every rank of a multi-rank launch, and a BLAS that was
    initialised with one thread cannot safely be widened afterwards: rank 0 therefore times the baseline in a child
    process of its own with the pin removed (the other ranks have left the process group and exited by then), and reads the record and
    the oracle's values back from a scratch file."""
    import pickle
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items()
           if k not in ('OMP_NUM_THREADS', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR',
                        'MASTER_PORT', 'GROUP_RANK', 'ROLE_RANK', 'ROLE_WORLD_SIZE', 'TORCHELASTIC_RUN_ID')}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'cpu.pkl')
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', path, '--workload',
                               workload, '--candidates', str(M), '--cpu-candidates', str(nc), '--cpu-span', str(span),
                               '--cpu-draws', ','.join(str(v) for v in draws),
                               '--cpu-topk', ','.join(str(int(v)) for v in topk_idx)], env=env,
                              stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        with open(path, 'rb') as fh:
            return pickle.load(fh)


def parity_record(w, ref_vals, dev_vals):
    """What BASELINE.md section 4 promised beside the CPU timing: the device's values against the oracle's on the
    candidates the oracle was timed on -- max relative error of mu, s2 and the acquisition, and whether both pick
    the same candidate of the sample.  Relative errors use the stated tolerance ladder's floors:
    |d mu| / (|mu| + 1e-3 sqrt(rho)), |d s2| / (s2 + 1e-4 rho), acquisition where it is within 1e-9 of its maximum."""
    rho = w['rho']
    out = {'n_compared': int(len(ref_vals['acq'])), 'against': 'oracle/gp_ref.py on the cpu_baseline sample'}
    if 'mu' in ref_vals:
        mr, sr = ref_vals['mu'], ref_vals['s2']
        out['max_rel_mu'] = float(np.max(np.abs(dev_vals['mu'] - mr) / (np.abs(mr) + 1e-3 * np.sqrt(rho))))
        out['max_rel_s2'] = float(np.max(np.abs(dev_vals['s2'] - sr) / (sr + 1e-4 * rho)))
        out['moments_within_stated_tolerance'] = bool(
            np.all(np.abs(dev_vals['mu'] - mr) <= 1e-6 * np.abs(mr) + 1e-9 * np.sqrt(rho)) and
            np.all(np.abs(dev_vals['s2'] - sr) <= 1e-6 * sr + 1e-10 * rho))
    ar, ad = ref_vals['acq'], dev_vals['acq']
    if ar.ndim == 2:
        # Thompson: every local draw; f = bias + (random-feature term): the error is measured against the TERM (size
        # ~0.1 sqrt(rho)), not against |bias| -- a 1e-16 relative to a bias of 10 says nothing about a term of 0.2
        term = ar - w['bias']
        scale = np.sqrt(np.mean(term ** 2, axis=1, keepdims=True))
        out['draws_compared'] = int(ar.shape[0])
        out['max_rel_acq'] = float(np.max(np.abs(ad - ar) / (np.abs(term) + 1e-3 * scale)))
        out['max_abs_err_over_rms_term'] = float(np.max(np.abs(ad - ar) / scale))
        out['rms_term'] = float(np.sqrt(np.mean(term ** 2)))
        out['n_acq_compared'] = int(ar.size)
        bd = np.argmax(ad, axis=1)
        out['selected_index_matches'] = bool(np.all(bd == ref_vals['best']))
        out['selected_index'] = {'oracle': [int(v) for v in ref_vals['best'][:8]], 'device': [int(v) for v in bd[:8]],
                                 'draws_agreeing': int(np.sum(bd == ref_vals['best']))}
        return out
    live = np.abs(ar) > 1e-9 * np.max(np.abs(ar))
    out['max_rel_acq'] = float(np.max(np.abs(ad[live] - ar[live]) / np.abs(ar[live])))
    out['n_acq_compared'] = int(live.sum())
    out['selected_index_matches'] = bool(int(np.argmax(ad)) == int(ref_vals['best']))
    out['selected_index'] = {'oracle': int(ref_vals['best']), 'device': int(np.argmax(ad))}
    if 'target' in ref_vals:
        out['abs_err_target'] = float(abs(dev_vals['target'] - ref_vals['target']))
    if 'topk' in ref_vals and 'top_idx' in dev_vals:
        # The ORDER of the device's top-k over the whole grid, judged by the oracle (the winner alone can be
        # a trivial candidate).  Tolerance of one value: the stated moment tolerances propagated to first order,
        # |d acq| <= Phi(z) dmu + phi(z) ds2 / (2 s) for EI, dmu + sqrt(beta) ds2 / (2 s) for UCB; two neighbours may
        # swap only if the oracle's gap between them is within the sum of their tolerances.
        tk = ref_vals['topk']
        mu_, s2_ = tk['mu'], tk['s2']
        s_ = np.sqrt(s2_)
        dmu = 1e-6 * np.abs(mu_) + 1e-9 * np.sqrt(rho)
        ds2 = 1e-6 * s2_ + 1e-10 * rho
        if w['acq'] == 'ei':
            from oracle import gp_ref
            z_ = (mu_ - ref_vals['target']) / s_
            tol = gp_ref.norm_cdf(z_) * dmu + gp_ref.norm_pdf(z_) * ds2 / (2.0 * s_)
        else:
            tol = dmu + np.sqrt(ucb_beta(w['N'])) * ds2 / (2.0 * s_)
        ov = tk['acq']                                    # oracle values IN THE DEVICE'S ORDER
        gaps = ov[:-1] - ov[1:]
        out['topk_order_matches'] = bool(np.all(gaps >= -(tol[:-1] + tol[1:])))
        out['topk_compared'] = int(len(ov))
        out['topk_strictly_ordered_pairs'] = int(np.sum(gaps > (tol[:-1] + tol[1:])))     # pairs the check can actually tell apart
        out['topk_max_rel_val'] = float(np.max(np.abs(dev_vals['top_val'][:len(ov)] - ov) / np.maximum(np.abs(ov), 1e-300)))
        # no candidate of the oracle's sample outranks the device's k-th by more than the tolerance unless it IS in the top-k
        outside = ~np.isin(dev_vals['sample_global_index'], tk['index'])
        out['sample_outranks_topk'] = int(np.sum(ar[outside] > ov[-1] + tol[-1])) if len(ar) == len(outside) else None
        out['winner_is_trivial'] = bool(int(tk['index'][0]) < 2)      # Sobol' points 0 / 1: the origin / the centre of the box
    return out


def ensemble_hypers(w, n):
    """n hyper-parameter vectors around the workload's own (what an MCMC chain over (sn2, rho, ell, bias) holds): member 0 is
    the workload's point estimate.  Shared by bench.py --ensemble and tests/test_gpu_ensemble_grid.py."""
    rng = np.random.RandomState(77)
    out = [(w['sn2'], w['rho'], np.array(w['ell'], dtype=float), w['bias'])]
    for _ in range(n - 1):
        out.append((w['sn2'] * np.exp(0.5 * rng.randn()), w['rho'] * np.exp(0.2 * rng.randn()),
                    w['ell'] * np.exp(0.15 * rng.randn(w['d'])), w['bias'] + 0.1 * np.sqrt(w['rho']) * rng.randn()))
    return out


def ensemble_bench(args):
    """`--ensemble n`: the step with the reference's DEFAULT model, MCMC(gp, n=10) (pybo/bayesopt.py:115): n member GPs
    (one hyper-parameter vector each) are fitted and the acquisition is averaged over them on the whole grid by ONE
    gpx_ensemble_sweep_dev call; top-k of the average.  Candidates sharded contiguously over the ranks like the headline
    step (every rank fits all n members; the one exchange is the top-k all-gather): with n = 10 members on 8 ranks a split
    by members would leave six ranks with half the work of the other two."""
    import torch
    from pybo_amd._lib import Engine, DeviceGrid
    from pybo_amd import dist as pdist
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.share_device >= 0:
        local = args.share_device
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, **({'device_id': dev} if args.backend == 'nccl' else {}))
    n = args.ensemble
    w = make_workload(args.workload, args.candidates)
    if w['acq'] not in ('ei', 'ucb'):
        raise SystemExit('--ensemble needs an EI / UCB workload')
    N, d, M, k = w['N'], w['d'], w['M'], args.topk
    lo_i, hi_i = (M * rank) // world, (M * (rank + 1)) // world
    Ml = hi_i - lo_i
    dX = torch.from_numpy(w['X']).to(dev)
    dy = torch.from_numpy(w['y']).to(dev)
    grid = DeviceGrid('sobol', np.stack([w['lo'], w['hi']], axis=1), Ml, first=lo_i, device=local)   # = w['Xc'][lo_i:hi_i], in HBM
    hyp = ensemble_hypers(w, n)
    engines = [Engine(local) for _ in range(n)]
    target = float(np.max(w['y'])) if w['acq'] == 'ei' else ucb_beta(N)

    def step():
        for e, (sn2, rho, ell, bias) in zip(engines, hyp):
            e.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], ell, rho, sn2, bias)
        r = Engine.ensemble_sweep(engines, w['acq'], target, grid, k=k)
        tv, ti = r['top_val'], np.where(r['top_idx'] >= 0, r['top_idx'] + lo_i, r['top_idx'])
        if world > 1:
            return pdist.gather_topk(tv, ti, k)
        return tv, ti

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        best = step()
    for e in engines:
        e.timers(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tms = [e.timers(reset=True) for e in engines]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    sec = elapsed / args.steps
    tot = {kk: sum(t[kk] for t in tms) for kk in ('gram', 'cholesky', 'trtri', 'alpha', 'cross_gram', 'sweep_trmm', 'acq_topk',
                                                  'sweep_trmm_launches', 'sweep_trmm_flop')}
    ach = tot['sweep_trmm_flop'] / (tot['sweep_trmm'] * 1e-3) / 1e12
    sclk = [t['sweep_sclk_mhz'] for t in tms if t.get('sweep_sclk_mhz')]
    sclk = float(np.mean(sclk)) if sclk else None
    out = {'metric': 'BO-step wall-clock with the default model (ensemble of %d GPs: %d fits + acquisition averaged over the members '
                     'on the candidate grid) at N obs; steps/sec' % (n, n),
           'value': 1.0 / sec, 'unit': 'steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': sec * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic',
           'config': {'workload': w['desc'] + '; model = ensemble of %d hyper-parameter vectors (pybo/bayesopt.py:115)' % n, 'N': N,
                      'd': d, 'candidates': M, 'kernel': w['kernel'], 'acquisition': w['acq'], 'members': n, 'topk': k,
                      'parallelism': 'candidates sharded contiguously over %d rank(s), every rank fits all members' % world},
           'selected': {'index': int(best[1][0]), 'value': float(best[0][0])},
           'stage_ms_per_step_rank0_all_members': {kk: tot[kk] / args.steps for kk in ('gram', 'cholesky', 'trtri', 'alpha', 'cross_gram',
                                                                                   'sweep_trmm', 'acq_topk')},
           'roofline': {'kernel': SWEEP_KERNEL, 'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': ach / FP64_MFMA_PEAK_TFLOPS, 'traffic': None, 'launches': int(tot['sweep_trmm_launches']),
                        'avg_launch_ms': tot['sweep_trmm'] / max(tot['sweep_trmm_launches'], 1),
                        'flop_per_launch': tot['sweep_trmm_flop'] / max(tot['sweep_trmm_launches'], 1),
                        'work': 'members x N^2 flop per candidate (algorithmic)', 'sclk_mhz': sclk,
                        'frac_at_measured_clock': ach / (FP64_MFMA_PEAK_TFLOPS * sclk / 2400.0) if sclk else None}}
    if not args.no_cpu_baseline:
        # the oracle's ensemble on a bounded sample: ALL n member fits in full + the members' EI / UCB on nc candidates
        from oracle import gp_ref
        nc = min(args.cpu_candidates or max(4096, (1 << 17) // n), Ml)
        threads = len(os.sched_getaffinity(0))
        idx = np.concatenate(sample_indices(nc, Ml))
        Z = w['Xc'][lo_i + idx]
        t_fit = t_sw = 0.0
        acc = np.zeros(len(idx))
        mus, s2s = [], []
        for sn2, rho, ell, bias in hyp:
            t0 = time.perf_counter()
            ref = gp_ref.make_gp(sn2, rho, ell, bias, w['kernel'])
            ref.add_data(w['X'], w['y'])
            t_fit += time.perf_counter() - t0
            t0 = time.perf_counter()
            mu, s2 = ref.predict(Z)
            if w['acq'] == 'ei':
                s_ = np.sqrt(s2)
                z_ = (mu - target) / s_
                acc += (mu - target) * gp_ref.norm_cdf(z_) + s_ * gp_ref.norm_pdf(z_)
            else:
                mus.append(mu)
                s2s.append(s2)
            t_sw += time.perf_counter() - t0
        if w['acq'] == 'ei':
            want = acc / n
        else:
            mm = np.mean(mus, axis=0)
            want = mm + np.sqrt(target * np.maximum(np.mean(np.array(s2s) + np.array(mus) ** 2, axis=0) - mm ** 2, 0.0))
        step_cpu = t_fit + t_sw * (M / float(len(idx)))
        out['cpu_baseline'] = {'value': 1.0 / step_cpu, 'unit': 'steps/s', 'cores': int(threads), 'kind': 'port',
                               'sample': '%d member fits in full (N=%d: %.1f s) + the members\' sweeps on %d of %d candidates '
                                         '(%.1f s), extrapolated linearly; numpy/scipy' % (n, N, t_fit, len(idx), M, t_sw),
                               'sample_frac': len(idx) / float(M), 'extrapolated': True, 'seconds_per_step': step_cpu}
        # parity on the same candidates (outside the timed region)
        dsel = torch.from_numpy(np.ascontiguousarray(Z)).to(dev)
        buf = torch.empty(len(idx), dtype=torch.float64, device=dev)
        import ctypes as C
        lead = engines[0]
        handles = (C.c_void_p * n)(*[e._h for e in engines])
        prm = np.array([target])
        lead._check(lead._lib.gpx_ensemble_sweep_dev(handles, n, {'ei': 0, 'pi': 1, 'ucb': 2}[w['acq']], prm.ctypes.data_as(C.c_void_p), 1,
                                                     dsel.data_ptr(), len(idx), 0, None, None, buf.data_ptr(), None, None))
        got = buf.cpu().numpy()
        live = np.abs(want) > 1e-9 * np.max(np.abs(want))
        out['parity'] = {'n_compared': int(len(idx)), 'against': 'oracle/gp_ref.py: %d member models on the cpu_baseline sample' % n,
                         'max_rel_acq': float(np.max(np.abs(got[live] - want[live]) / np.abs(want[live]))),
                         'n_acq_compared': int(live.sum()),
                         'selected_index_matches': bool(int(np.argmax(got)) == int(np.argmax(want)))}
    print(json.dumps(out))


def plugin_step(w, nsteps, k):
    """One BO iteration THROUGH THE PLUGIN API at the workload size, end to end: `pybo_amd.solve_bayesopt` resumed
    from a checkpoint that holds the N observations (model + trace, the reference's own resume path,
    pybo/bayesopt.py:236-262), policy by name, solver 'lbfgs' over a grid resident in HBM, recommender 'latent',
    checkpoint written every iteration.  An iteration = policy (model.copy + target) + solver (grid stage + L-BFGS-B
    refinement of the best seed) + objective + model.add_data + recommender + checkpoint.  Timed from the objective
    calls: the span between two consecutive calls is one full iteration's work (the tail of one, the head of the
    next).  The first span (cold) contains the unpickled model's refit and the full N^2 M sweep; from then on the
    grid stage is a warm re-score."""
    import tempfile
    import pybo_amd
    from pybo_amd import models, inits
    from pybo_amd.bayesopt import safe_dump, Info
    N, d, M = w['N'], w['d'], w['M']
    bounds = np.stack([w['lo'], w['hi']], axis=1)
    gp = models.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], kernel=w['kernel'])
    gp._X, gp._Y = np.array(w['X']), np.array(w['y'])         # data only: the run below starts from the pickle
    grid = inits.DeviceGrid('sobol', bounds, M)                # the points of w['Xc'], generated in HBM
    rng = np.random.RandomState(11)
    stamps = []

    def objective(x):
        stamps.append(time.perf_counter())
        return float(w['f'](np.array(x, ndmin=2))[0] + 1e-3 * rng.randn())

    policy = 'ei' if w['acq'] == 'ei' else 'ucb'
    with tempfile.TemporaryDirectory() as tmp:
        log = os.path.join(tmp, 'bo.pkl')
        safe_dump(gp, Info(list(w['X']), list(w['y']), list(w['X'])), log)
        del gp
        t0 = time.perf_counter()
        xbest, model, info = pybo_amd.solve_bayesopt(objective, bounds, niter=N + nsteps, policy=policy,
                                                     solver=('lbfgs', {'xgrid': grid, 'nbest': k}),
                                                     recommender='latent', log=log)
        t1 = time.perf_counter()
        ckpt_bytes = os.path.getsize(log)
        # the same loop continued with a black box that takes 40 ms per evaluation (a sleep: the GPU stays free): what
        # an iteration costs BEYOND its objective call.  The query point is announced to the device before the
        # objective is called (model.anticipate -> gpx_append_begin), so the cache-correction pass runs meanwhile.
        slow_ms, nslow, marks = 40.0, min(4, nsteps), []

        def slow_objective(x):
            marks.append(time.perf_counter())
            time.sleep(slow_ms * 1e-3)
            marks.append(time.perf_counter())
            return float(w['f'](np.array(x, ndmin=2))[0] + 1e-3 * rng.randn())

        pybo_amd.solve_bayesopt(slow_objective, bounds, niter=N + nsteps + nslow, policy=policy,
                                solver=('lbfgs', {'xgrid': grid, 'nbest': k}), recommender='latent', log=log)
        between = [(marks[2 * i + 2] - marks[2 * i + 1]) * 1e3 for i in range(nslow - 1)]    # end of call i -> start of call i+1
    spans = np.diff(stamps)
    rec = {'what': 'pybo_amd.solve_bayesopt resumed at N observations, policy %r, solver lbfgs over a DeviceGrid of '
                   '%d Sobol points, recommender latent, checkpoint every iteration; spans between consecutive '
                   'objective calls' % (policy, M),
           'iterations': int(nsteps), 'N_start': int(N),
           'cold_ms': (stamps[0] - t0) * 1e3,
           'cold_what': 'checkpoint load + fit + policy + full sweep + refinement up to the first objective call',
           'tail_ms': (t1 - stamps[-1]) * 1e3,
           'checkpoint_bytes': int(ckpt_bytes), 'selected_last': [float(v) for v in info.x[-1]]}
    if len(spans):
        rec['warm_ms'] = float(np.mean(spans) * 1e3)
        rec['warm_ms_each'] = [float(s * 1e3) for s in spans]
        rec['warm_ms_median'] = float(np.median(spans) * 1e3)     # (the first span holds the factor's growth by a block: N sits on a 128-boundary)
    if between:
        rec['warm_ms_beside_a_%d_ms_objective' % int(slow_ms)] = float(np.mean(between))
        rec['beside_what'] = ('time from the return of one objective call to the start of the next (add_data + recommender + '
                              'checkpoint + policy + solver) when the objective itself takes %d ms: the N*M covariance '
                              'evaluations of the cache correction ran during the call (announced query point)' % int(slow_ms))
    return rec


def nontrivial_selection_record(eng, dev, M, k, cpu_candidates):
    """Workload 'ns2' (the north-star problem with its optimum moved off the grid centre) through the device path -- fit,
    EI sweep over ALL M candidates, top-k -- and the oracle's verdict on that top-k and on a sample of the grid
    (parity_record: order of the top-k, nothing in the sample outranks it, moments within the stated tolerance)."""
    import torch
    w2 = make_workload('ns2', M)
    dX2, dy2 = torch.from_numpy(w2['X']).to(dev), torch.from_numpy(w2['y']).to(dev)
    dZ = torch.from_numpy(w2['Xc']).to(dev)
    eng.fit_dev(dX2.data_ptr(), w2['N'], w2['d'], dy2.data_ptr(), w2['kernel'], w2['ell'], w2['rho'], w2['sn2'], w2['bias'])
    param = eng.mean_at_obs()[1]
    tv, ti = eng.sweep_dev('ei', param, dZ.data_ptr(), M, k)
    nc = min(cpu_candidates or 4096, 4096, M)
    _, ref_vals = cpu_baseline(w2, nc, M, [], [int(v) for v in np.asarray(ti).ravel() if v >= 0])
    sidx = ref_vals['index']
    dsel = torch.from_numpy(np.ascontiguousarray(w2['Xc'][sidx])).to(dev)
    buf = torch.empty(3, len(sidx), dtype=torch.float64, device=dev)
    eng.sweep_dev('ei', param, dsel.data_ptr(), len(sidx), 0, d_acq=buf[0].data_ptr(), d_mu=buf[1].data_ptr(),
                  d_s2=buf[2].data_ptr())
    eng.sync()
    hb = buf.cpu().numpy()
    rec = parity_record(w2, ref_vals, dict(acq=hb[0], mu=hb[1], s2=hb[2], target=float(param), top_val=np.asarray(tv).ravel(),
                                           top_idx=np.asarray(ti).ravel(), sample_global_index=sidx))
    rec['workload'] = w2['desc']
    rec['selected'] = {'index': int(np.asarray(ti).ravel()[0]), 'value': float(np.asarray(tv).ravel()[0])}
    rec['timed'] = False
    return rec


def bring_up_gpx_exchange(args, eng, rank, world, dist, dev):
    """The north-star exchange: libgpx's own RCCL communicator on the engine's stream (gpx_comm_init / gpx_topk_allgather).
    Returns (Comm, 'gpx') or -- with --exchange auto only -- a string 'torch (fallback: <why>)' after saying so on stderr.
    Every decision is taken by ALL ranks together (one MIN all-reduce of a status word per phase), and the binding is
    probed on every rank BEFORE the collective ncclCommInitRank, so that a rank that cannot load librccl does not leave
    the others blocked inside the rendezvous."""
    import torch
    from pybo_amd._lib import Comm

    def agree(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def give_up(why):
        msg = 'bench.py: rank %d: the gpx (device-side RCCL) exchange is NOT in use: %s' % (rank, why)
        if args.exchange == 'gpx':
            sys.stderr.write(msg + ' -- --exchange gpx was asked for: FATAL\n')
            sys.stderr.flush()
            os._exit(5)
        sys.stderr.write(msg + ' -- FALLING BACK to the torch.distributed transport\n')
        sys.stderr.flush()
        return 'torch (fallback: %s)' % why

    if args.share_device >= 0:
        return give_up('the ranks share device %d (dry run): RCCL refuses two ranks on one GPU' % args.share_device)
    why, uid = '', None
    try:
        uid = Comm.unique_id()                       # loads + binds librccl in THIS process; no communication
    except Exception as exc:                         # noqa: BLE001
        why = 'librccl could not be bound: %s' % exc
    if not agree(uid is not None):
        return give_up(why or 'another rank could not bind librccl')
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    comm = None
    try:
        comm = Comm(eng, rank, world, box[0])        # ncclCommInitRank: collective over all ranks
    except Exception as exc:                         # noqa: BLE001
        why = 'gpx_comm_init failed: %s' % exc
    if not agree(comm is not None):
        if comm is not None:
            comm.close()
        return give_up(why or 'gpx_comm_init failed on another rank')
    return comm, 'gpx'


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher (the driver's command shape): start N ranks of this very
    command through torch.distributed.run on a free local port, one GPU per rank; rank 0 prints the ONE JSON line
    on the inherited stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, stdin=subprocess.DEVNULL)


def main_sharded(args):
    """`--mode sharded`: ONE process drives --gpus devices through pybo_amd.models.ShardedGP (one gpx handle per device,
    host threads) and a ShardedDeviceGrid -- the multi-GPU drop-in under solve_bayesopt() (the objective is evaluated once,
    in this process; no torch.distributed, no RCCL: the exchange is a host merge of P x k pairs).  Same step as the SPMD
    mode: every replica refits (replicated, bitwise equal), sweeps its shard of the grid, top-k merged."""
    import torch
    from pybo_amd import models
    from pybo_amd.models import sharded as msh
    from pybo_amd._lib import ShardedDeviceGrid
    P = args.gpus
    ndev = torch.cuda.device_count()
    if args.share_device < 0 and P > ndev:
        sys.stderr.write('bench.py: FATAL: --mode sharded --gpus %d but only %d GPU(s) visible\n' % (P, ndev))
        raise SystemExit(3)
    devices = list(range(P)) if args.share_device < 0 else [args.share_device] * P
    w = make_workload(args.workload, args.candidates)
    if w['acq'] == 'thompson':
        raise SystemExit('--mode sharded times the EI / UCB workloads (ns, b, c); the Thompson workloads shard DRAWS, '
                         'which the SPMD mode measures')
    N, d, M, k = w['N'], w['d'], w['M'], args.topk
    bounds = np.stack([w['lo'], w['hi']], axis=1)
    gp = models.make_gp(w['sn2'], w['rho'], w['ell'], w['bias'], kernel=w['kernel'], devices=devices)
    gp.add_data(w['X'], w['y'])
    grid = ShardedDeviceGrid('sobol', bounds, M, devices)      # the points of w['Xc'], laid out over the devices
    for r in gp.replicas:
        for kv in args.opt:
            name, val = kv.split('=')
            r._engine().set_option(name, int(val))

    def step():
        def refit(r):
            r._fitted = False
            return r._engine()
        msh._run([lambda r=r: refit(r) for r in gp.replicas])
        param = float(np.max(gp.posterior_mean_at_data())) if w['acq'] == 'ei' else ucb_beta(N)
        return gp.acq_topk(w['acq'], param, grid, k)

    def sync():
        for r in gp.replicas:
            r._engine().sync()
    for _ in range(args.warmup):
        best = step()
    for r in gp.replicas:
        r._engine().timers(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
    sync()
    sec = (time.perf_counter() - t0) / args.steps
    tms = [r._engine().timers(reset=True) for r in gp.replicas]
    stages = ('gram', 'cholesky', 'trtri', 'alpha', 'cross_gram', 'sweep_trmm', 'acq_topk')
    per = {kk: {'min': min(t[kk] for t in tms) / args.steps, 'mean': float(np.mean([t[kk] for t in tms])) / args.steps,
                'max': max(t[kk] for t in tms) / args.steps} for kk in stages if tms[0][kk] > 0}
    tm = tms[0]
    ach = tm['sweep_trmm_flop'] / (tm['sweep_trmm'] * 1e-3) / 1e12
    out = {'metric': 'BO-step wall-clock (GP fit + 1e6-candidate acq sweep) at N obs; steps/sec',
           'value': 1.0 / sec, 'unit': 'steps/s', 'n_gpus': P, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': sec * 1e3, 'seconds_per_step': sec, 'higher_is_better': True, 'scaling': 'strong',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'mode': 'sharded',
           'config': {'workload': w['desc'], 'N': N, 'd': d, 'candidates': M, 'kernel': w['kernel'], 'acquisition': w['acq'],
                      'topk': k, 'devices': devices,
                      'parallelism': 'ONE process, %d device handles (pybo_amd.models.ShardedGP): fit replicated, grid resident '
                                     'and sharded contiguously (ShardedDeviceGrid), host merge of the top-k' % P},
           'selected': {'index': int(best[1][0]), 'value': float(best[0][0])},
           'stage_ms_per_step_all_replicas': per,
           'roofline': {'kernel': SWEEP_KERNEL, 'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / FP64_MFMA_PEAK_TFLOPS, 'traffic': None,
                        'launches': int(tm['sweep_trmm_launches']), 'of': 'replica 0'}}
    if not args.no_cpu_baseline:
        span = M // P
        nc = min(args.cpu_candidates or (M if N <= 2048 else (1 << 17)), span)
        out['cpu_baseline'], _ = cpu_baseline(w, nc, span, [])
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='ns', choices=['ns', 'ns2', 'b', 'c', 'd', 'e'])
    ap.add_argument('--candidates', type=int, default=1 << 20)
    ap.add_argument('--topk', type=int, default=10)
    ap.add_argument('--chunk', type=int, default=0)
    ap.add_argument('--tile-order', type=int, default=-1)
    ap.add_argument('--draws', type=int, default=0, help='Thompson draws (default: 64 for workload d, 8 for e)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-refine', action='store_true', help='skip the separately reported L-BFGS refinement')
    ap.add_argument('--cpu-candidates', type=int, default=0, help='candidates in the CPU sample (0 = auto)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend; gloo only for dry runs of the N>1 path')
    ap.add_argument('--share-device', type=int, default=-1,
                    help='dry run: every rank uses this one GPU (with --backend gloo)')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE',
                    help='engine option (gpx_set_option), e.g. --opt potrf=0; repeatable')
    ap.add_argument('--warm-batch', type=int, default=1,
                    help='observations appended per warm step (q of a batch-BO add_data(X, Y)); their cache '
                         'corrections share one pass over the candidates')
    ap.add_argument('--warm-steps', type=int, default=8,
                    help='also time this many WARM iterations (append one observation + re-score the cached '
                         'sweep sums); reported separately as warm_step, never as value; 0 = skip')
    ap.add_argument('--exchange', default='auto', choices=['auto', 'torch', 'gpx'],
                    help="transport of the top-k exchange: gpx = libgpx's own RCCL binding (gpx_topk_allgather: the pairs go "
                         "device -> xGMI -> device, only the k winners reach the host; needs one GPU per rank), torch = "
                         "torch.distributed (the pairs travel through host tensors).  auto (default) = gpx, falling through "
                         "to torch -- loudly, on stderr and in collective.exchange -- only if the RCCL binding cannot be "
                         "brought up on every rank (or the ranks share one device: dry runs)")
    ap.add_argument('--plugin-steps', type=int, default=12,
                    help='also time this many iterations of pybo_amd.solve_bayesopt THROUGH THE PLUGIN API at the '
                         'workload size (cold + warm; reported separately as plugin_step); 0 = skip')
    ap.add_argument('--ensemble', type=int, default=0,
                    help='n > 0: time the step with the reference\'s DEFAULT model instead -- an ensemble of n GPs (pybo/bayesopt.py:115: '
                         'MCMC(gp, n=10)): n fits + ONE gpx_ensemble_sweep over the grid; own roofline (n x N^2 x M flop) and cpu_baseline')
    ap.add_argument('--cpu-baseline-worker', default='', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-span', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-draws', default='', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-topk', default='', help=argparse.SUPPRESS)
    ap.add_argument('--mode', default='spmd', choices=['spmd', 'sharded'],
                    help="spmd (default): one process per GPU under torch.distributed, RCCL all-gather of the top-k; "
                         "sharded: ONE process drives --gpus devices through pybo_amd.models.ShardedGP + "
                         "ShardedDeviceGrid -- the drop-in under solve_bayesopt(), no process group")
    ap.add_argument('--init-timeout', type=int, default=180, help='seconds to wait for the process group / RCCL to come up')
    args = ap.parse_args()

    if args.cpu_baseline_worker:                      # child of cpu_baseline_unpinned: no GPU, no process group
        import pickle
        res = cpu_baseline(make_workload(args.workload, args.candidates), args.cpu_candidates, args.cpu_span,
                           [int(v) for v in args.cpu_draws.split(',') if v != ''],
                           [int(v) for v in args.cpu_topk.split(',') if v != ''])
        with open(args.cpu_baseline_worker, 'wb') as fh:
            pickle.dump(res, fh)
        return

    if args.mode == 'sharded':
        return main_sharded(args)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # invoked as `python bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit(self_launch(args.gpus))
    if args.ensemble > 0:
        return ensemble_bench(args)

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    if args.share_device >= 0:
        local = args.share_device
    ndev = torch.cuda.device_count()
    if args.share_device < 0 and world > ndev:
        masked = [v for v in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES') if os.environ.get(v)]
        if ndev >= 1 and masked:
            # a launcher that shows every rank its own device(s) only: this rank cannot see that the others hold different
            # GPUs -- go on with what is visible; if two ranks DO share a device RCCL refuses the communicator (exit code 4)
            sys.stderr.write('bench.py: rank %d sees %d device(s) of a %d-rank job under %s=%s: using local device %d\n'
                             % (rank, ndev, world, masked[0], os.environ[masked[0]], local % ndev))
            local = local % ndev
        else:
            # one rank per GPU is the layout: more ranks than visible devices would put two ranks on one GPU and RCCL
            # refuses that ("invalid usage") only after a long rendezvous -- fail at once, and loudly
            sys.stderr.write('bench.py: FATAL: WORLD_SIZE=%d but only %d GPU(s) visible to rank %d\n' % (world, ndev, rank))
            raise SystemExit(3)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    rccl_info = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=args.init_timeout)
        try:
            if args.backend == 'nccl':
                dist.init_process_group('nccl', device_id=dev, timeout=tmo)      # = RCCL over xGMI on MI355X
            else:
                # the gloo transport announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."); the contract
                # is ONE JSON line there, so fd 1 points at stderr while the group comes up
                sys.stdout.flush()
                keep = os.dup(1)
                os.dup2(2, 1)
                try:
                    dist.init_process_group('gloo', timeout=tmo)
                    dist.barrier()
                finally:
                    sys.stdout.flush()
                    os.dup2(keep, 1)
                    os.close(keep)
            # prove the communicator with one all-reduce before anything is timed: every rank contributes its rank + 1
            probe = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            t0p = time.perf_counter()
            dist.all_reduce(probe)
            if args.backend == 'nccl':
                torch.cuda.synchronize(dev)
            first_ms = (time.perf_counter() - t0p) * 1e3
            if abs(float(probe.item()) - world * (world + 1) / 2.0) > 1e-9:
                raise RuntimeError('all-reduce over %d ranks returned %r' % (world, float(probe.item())))
            ver = None
            if args.backend == 'nccl':
                try:
                    ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
                except Exception:
                    ver = None
            rccl_info = {'backend': args.backend, 'ranks': world, 'devices_visible': ndev, 'rccl_version': ver,
                         'first_allreduce_ms': first_ms, 'init_timeout_s': args.init_timeout}
        except BaseException as exc:      # noqa: NEVER fall back to another backend on our own: say what failed, exit non-zero
            sys.stderr.write('bench.py: FATAL: rank %d could not bring up the %s process group over %d ranks: %r\n'
                             % (rank, args.backend, world, exc))
            sys.stderr.flush()
            os._exit(4)

    from pybo_amd._lib import Engine
    from pybo_amd import dist as pdist
    w = make_workload(args.workload, args.candidates)
    N, d, M, k = w['N'], w['d'], w['M'], args.topk
    # contiguous candidate slice of this rank
    lo_i, hi_i = (M * rank) // world, (M * (rank + 1)) // world
    dX = torch.from_numpy(w['X']).to(dev)
    dy = torch.from_numpy(w['y']).to(dev)
    dXc = torch.from_numpy(np.ascontiguousarray(w['Xc'][lo_i:hi_i])).to(dev)
    stream = torch.cuda.current_stream(dev)
    eng = Engine(local, stream.cuda_stream)
    if args.chunk:
        eng.set_option('chunk', args.chunk)
    if args.tile_order >= 0:
        eng.set_option('tile_order', args.tile_order)
    for kv in args.opt:
        name, val = kv.split('=')
        eng.set_option(name, int(val))
    Ml = hi_i - lo_i
    comm = None
    exchange_used = 'torch' if world > 1 else None
    if world > 1 and args.exchange in ('auto', 'gpx'):
        exchange_used = bring_up_gpx_exchange(args, eng, rank, world, dist, dev)
        if isinstance(exchange_used, tuple):
            comm, exchange_used = exchange_used

    thompson = None
    if w['acq'] == 'thompson':
        S = args.draws or (8 if w['name'] == 'e' else 64)
        mine = [s for s in range(S) if s % world == rank]      # draws are the sharded unit here

    split = {'local': [], 'exchange': []}      # per step, this rank: seconds before / inside the exchange

    def step():
        t_a = time.perf_counter()
        if w['acq'] == 'thompson':
            # each rank owns S/world posterior draws and sweeps ALL candidates for them.  The host draws of the random
            # features (numpy, the caller's seeded streams: 6-15 ms for 64 draws at d = 32) do not depend on the fit: they
            # are made while the device factorises (the ctypes call releases the GIL; a worker thread carries it)
            import threading
            err = []

            def fit():
                try:
                    eng.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
                except BaseException as exc:      # noqa: re-raised on the caller's thread below
                    err.append(exc)
            th_fit = threading.Thread(target=fit)
            th_fit.start()
            Ws, bs, zs = [], [], []
            for s in mine:
                Wd, bd_, zd_ = thompson_draw(w, s)       # same draw order as GP.sample_f / the oracle
                Ws.append(Wd)
                bs.append(bd_)
                zs.append(zd_)
            th_fit.join()
            if err:
                raise err[0]
        else:
            eng.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
        if w['acq'] == 'thompson':
            Ws, bs = np.array(Ws), np.array(bs)
            sc = np.sqrt(2.0 * w['rho'] / 100)
            # feature Grams and the 100 x 100 weight posteriors of all local draws: one device call
            ths = eng.rff_posterior(Ws, bs, np.array(zs), sc)
            tv, ti = eng.rff_sweep_dev(Ws, bs, ths, w['bias'],
                                       dXc_full.data_ptr(), M, 1)
            tv, ti = tv[:, 0], ti[:, 0]
        else:
            if w['acq'] == 'ei':
                _, mx = eng.mean_at_obs()
                param = mx
            elif w['acq'] == 'ucb':
                param = ucb_beta(N)
            tv, ti = eng.sweep_dev(w['acq'], param, dXc.data_ptr(), Ml, k)
            ti = np.where(ti >= 0, ti + lo_i, ti)
        t_b = time.perf_counter()                   # (the top-k came back to the host: the local work is complete)
        if world > 1:
            if w['acq'] == 'thompson':
                # one (value, index) pair per draw, draws sharded over ranks: ONE all-gather, no merge
                # (the q recommendations are w['Xc'][indices])
                # (the device-side all-gather moves the same number of pairs from every rank: uneven draw counts go through torch)
                even = (S % world == 0)
                res = comm.topk_allgather(len(tv), 0, 0) if (comm is not None and even) else pdist.gather_pairs(tv, ti)
            elif comm is not None:                   # libgpx's own RCCL binding: device -> xGMI -> device merge
                res = comm.topk_allgather(k, lo_i, k)
            else:
                res = pdist.gather_topk(tv, ti, k)   # RCCL all-gather + deterministic merge
        else:
            res = (tv, ti)
        split['local'].append(t_b - t_a)
        split['exchange'].append(time.perf_counter() - t_b)     # includes the wait for the slowest rank
        return res

    if w['acq'] == 'thompson':
        dXc_full = torch.from_numpy(w['Xc']).to(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        best = step()
    eng.timers(reset=True)
    split['local'], split['exchange'] = [], []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    tm = eng.timers(reset=True)
    # every rank's own view of the timed region (N > 1: gathered so that ONE line tells where the time went on WHICH rank)
    mine_rec = {'rank': rank, 'device': local, 'local_ms': 1e3 * float(np.mean(split['local'])),
                'exchange_ms': 1e3 * float(np.mean(split['exchange'])),
                'exchange_ms_max': 1e3 * float(np.max(split['exchange'])),
                'stage_ms': {kk: tm[kk] / args.steps for kk in ('gram', 'cholesky', 'trtri', 'alpha', 'cross_gram',
                                                                'sweep_trmm', 'acq_topk', 'rff', 'rff_sweep') if tm[kk] > 0},
                'chol_fallbacks': int(tm.get('chol_fallbacks', 0))}
    all_recs = [mine_rec]
    if world > 1:
        all_recs = [None] * world
        dist.all_gather_object(all_recs, mine_rec)

    # ---- warm BO step (reported SEPARATELY; the headline stays the cold step above) ------------------------
    # Iteration t+1 of the loop with fixed hyper-parameters over the same resident grid: model.add_data(x, y)
    # -> policy -> solver's grid sweep (pybo/bayesopt.py:262-269).  The engine extends the factor by one row
    # (gpx_append), corrects the cached per-candidate sums of the last sweep for it (N*M covariance evaluations
    # instead of the N^2*M product) and re-scores the grid (gpx_sweep_update).  N grows by one per step.
    warm = None
    if args.warm_steps > 0 and w['acq'] != 'thompson':
        eng.set_option('sweep_cache', 1)
        step()                                          # one cold step fills the cache (untimed)
        eng.set_option('sweep_cache', 0)
        rngw = np.random.RandomState(7)
        wq = max(1, args.warm_batch)
        Xn = w['lo'] + (w['hi'] - w['lo']) * rngw.rand(args.warm_steps * wq, d)
        yn = w['f'](Xn) + 1e-3 * rngw.randn(args.warm_steps * wq)
        eng.timers(reset=True)
        fence()
        t0 = time.perf_counter()
        for i in range(args.warm_steps):
            for j in range(i * wq, (i + 1) * wq):
                eng.append(Xn[j], yn[j])
            param = eng.mean_at_obs()[1] if w['acq'] == 'ei' else ucb_beta(N + (i + 1) * wq)
            r = eng.sweep_update(w['acq'], param, k=k, want_all=False)
            tv, ti = r['top_val'], np.where(r['top_idx'] >= 0, r['top_idx'] + lo_i, r['top_idx'])
            if world > 1:
                wbest = comm.topk_allgather(k, lo_i, k) if comm is not None else pdist.gather_topk(tv, ti, k)
            else:
                wbest = (tv, ti)
        fence()
        welapsed = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([welapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            welapsed = float(te.item())
        tw = eng.timers(reset=True)
        wsec = welapsed / args.warm_steps
        warm = {'ms_per_step': wsec * 1e3, 'value': 1.0 / wsec, 'unit': 'steps/s', 'steps': args.warm_steps,
                'what': 'gpx_append (rank-1 extension of the factor + rank-1 correction of the cached sweep sums) '
                        '+ EI target / UCB beta + gpx_sweep_update over the same candidates; hyper-parameters '
                        'fixed, N grows by %d per step from %d' % (wq, N),
                'appends_per_step': wq,
                'stage_ms_per_step_rank0': {kk: tw[kk] / args.warm_steps for kk in ('append', 'rank1', 'acq_topk')},
                'selected': {'index': int(wbest[1][0]), 'value': float(wbest[0][0])},
                'speedup_vs_cold_step': (elapsed / args.steps) / wsec}

    # ---- every collective of the run is behind us: leave the process group NOW, together.  What follows is rank 0's alone
    # (the refinement, the CPU baseline: minutes of host time, the parity record, the plug-in loop); ranks waiting for it
    # inside an RCCL barrier would sit under the process group's collective timeout -- a baseline slower than that timeout
    # would have turned a finished measurement into a watchdog abort.  The other ranks exit 0 here.
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
        if rank != 0:
            return

    # ---- the solver's refinement (A7) and the recommender (A8), reported SEPARATELY (excluded from the
    # headline): L-BFGS-B from the sweep's seeds with device gradients, exactly what solve_lbfgs does after the
    # grid sweep (pybo/solvers/lbfgs.py:56-68) -- the reference's behaviour (only the best seed's refinement is
    # returned, so only that one is computed) and the all-seeds variant in lock-step.
    refine = None
    if rank == 0 and w['acq'] in ('ei', 'ucb') and not args.no_refine:
        import scipy.optimize
        from scipy.special import erfc
        from pybo_amd.solvers.lbfgs import _refine_lockstep
        eng.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
        param = eng.mean_at_obs()[1] if w['acq'] == 'ei' else ucb_beta(N)
        tv, ti = eng.sweep_dev(w['acq'], param, dXc.data_ptr(), Ml, k)
        seeds = w['Xc'][lo_i + ti]
        box = np.stack([w['lo'], w['hi']], axis=1)
        ncall = [0]

        def index(X, grad=True):
            ncall[0] += 1
            mu, s2, dmu, ds2 = eng.predict(np.array(X, ndmin=2, dtype=float), grad=True)
            if w['acq'] == 'ucb':
                return mu + np.sqrt(param * s2), dmu + 0.5 * np.sqrt(param / s2)[:, None] * ds2
            s = np.sqrt(s2)
            z = (mu - param) / s
            cdf = 0.5 * erfc(-z * 0.70710678118654752440)
            pdf = 0.39894228040143267794 * np.exp(-0.5 * z * z)
            return (mu - param) * cdf + s * pdf, cdf[:, None] * dmu + (0.5 * pdf / s)[:, None] * ds2

        def negated(x):
            fx, gx = index(x[None])
            return -fx[0], -gx[0]
        index(seeds[:1])                                 # untimed: the first call allocates the handle's gradient scratch
        ncall[0] = 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        x1, f1 = scipy.optimize.fmin_l_bfgs_b(negated, seeds[0], bounds=box)[:2]
        t_first, c_first = time.perf_counter() - t0, ncall[0]
        ncall[0] = 0
        t0 = time.perf_counter()
        res = _refine_lockstep(index, seeds, box)
        t_all, c_all = time.perf_counter() - t0, ncall[0]
        refine = {'what': 'L-BFGS-B refinement of the sweep seeds with device gradients (gpx_predict with grad); '
                          'not part of the headline step',
                  'first_seed': {'ms': t_first * 1e3, 'gradient_calls': c_first, 'value': float(-f1)},
                  'all_%d_seeds_lockstep' % len(seeds): {'ms': t_all * 1e3, 'batched_gradient_calls': c_all,
                                                         'best_value': float(-min(r[1] for r in res))}}

    if rank == 0:
        Np_ = (N + 127) // 128 * 128
        sec = elapsed / args.steps
        out = {
            'metric': 'BO-step wall-clock (GP fit + 1e6-candidate acq sweep) at N obs; steps/sec',
            'value': 1.0 / sec, 'unit': 'steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': sec * 1e3, 'seconds_per_step': sec,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': w['desc'], 'N': N, 'd': d, 'candidates': M, 'kernel': w['kernel'],
                       'acquisition': w['acq'], 'topk': k,
                       'parallelism': 'candidates sharded contiguously over %d rank(s), fit replicated, '
                                      'all-gather of (value,index) top-k' % world},
            'selected': {'index': int(best[1][0]), 'value': float(best[0][0])},
            'stage_ms_per_step_rank0': {kk: tm[kk] / args.steps for kk in
                                        ('gram', 'cholesky', 'trtri', 'alpha', 'cross_gram', 'sweep_trmm',
                                         'acq_topk', 'rff') if tm[kk] > 0},
        }
        # min / mean / max over ALL ranks, per stage (HIP events on each rank's stream) and for the two host-side spans of
        # a step: `local` (fit + sweep + top-k back on the host) and `exchange` (the all-gather + merge, INCLUDING the wait
        # for the slowest rank: a straggler shows up as a large exchange on the others)
        def _mmm(vals):
            return {'min': float(np.min(vals)), 'mean': float(np.mean(vals)), 'max': float(np.max(vals)),
                    'argmax_rank': int(np.argmax(vals))}
        per = {'local': _mmm([r['local_ms'] for r in all_recs]), 'exchange': _mmm([r['exchange_ms'] for r in all_recs]),
               'exchange_worst_step': _mmm([r['exchange_ms_max'] for r in all_recs])}
        for kk in sorted(set().union(*[set(r['stage_ms']) for r in all_recs])):
            per[kk] = _mmm([r['stage_ms'].get(kk, 0.0) for r in all_recs])
        out['stage_ms_per_step_all_ranks'] = per
        out['ranks'] = [{'rank': r['rank'], 'device': r['device'], 'local_ms': r['local_ms'], 'exchange_ms': r['exchange_ms']}
                        for r in all_recs]
        fb = sum(r['chol_fallbacks'] for r in all_recs)
        if fb:
            out['chol_taskgraph_fallbacks'] = int(fb)        # fits re-run on the stream schedule (should be 0)
        if rccl_info is not None:
            out['collective'] = dict(rccl_info, exchange=exchange_used, exchange_requested=args.exchange)
        # what the 1-GPU stage times predict for this rank count (the replicated fit is the serial term)
        try:
            one = json.load(open(os.path.join(ROOT, 'profiles', 'r06_bench_%s.json' % w['name'])))
            st1 = one['stage_ms_per_step_rank0']
            if w['acq'] == 'thompson':
                par = st1.get('rff', 0.0)
                ser = sum(v for kk, v in st1.items() if kk not in ('rff',))
                units = 'draws'
                share = float(np.ceil((args.draws or (8 if w['name'] == 'e' else 64)) / float(world))) / \
                    (args.draws or (8 if w['name'] == 'e' else 64))
            else:
                par = st1.get('cross_gram', 0.0) + st1.get('sweep_trmm', 0.0) + st1.get('acq_topk', 0.0)
                ser = sum(v for kk, v in st1.items() if kk not in ('cross_gram', 'sweep_trmm', 'acq_topk'))
                units = 'candidates'
                share = 1.0 / world
            out['scaling_model'] = {'formula': 'serial (replicated fit) + parallel (%s sharded) x share + exchange (~0.1 ms)' % units,
                                    'from': 'profiles/r06_bench_%s.json (1 GPU)' % w['name'], 'serial_ms': ser,
                                    'parallel_ms_1gpu': par, 'share': share, 'predicted_ms_per_step': ser + par * share + 0.1,
                                    'measured_ms_per_step': sec * 1e3}
            if w['name'] == 'd':
                # the fit is "replicas only" -- at N = 16384 it is most of the step and does not divide by P
                out['scaling_model']['note'] = (
                    'config D is fit-dominated and the fit is REPLICATED on every rank (SURVEY 8(e): replicas only): only the '
                    'Thompson stage (%.1f of %.1f ms) divides by the rank count -- expect about %.2fx at 8 ranks, not 8x'
                    % (par, ser + par, (ser + par) / (ser + par / 8.0 + 0.1)))
        except Exception:
            pass
        if tm['sweep_trmm'] > 0:
            launches = tm['sweep_trmm_launches']
            # HBM traffic per launch comes from the committed PMC passes (bench.py cannot collect
            # counters itself); only reported when this run's launch geometry is the profiled one
            traffic, traffic_file = None, None
            for fname in PMC_TRAFFIC_FILES:
                try:
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', fname)))
                    cols = tm['sweep_trmm_flop'] / max(launches, 1) / (float(N) * N)
                    if pmc['config']['Np'] == Np_ and abs(cols - pmc['config']['cols_per_launch']) < 1:
                        traffic, traffic_file = pmc['k_sweep_trmm']['traffic_bytes_per_launch'], fname
                        break
                except Exception:
                    continue
            ach = tm['sweep_trmm_flop'] / (tm['sweep_trmm'] * 1e-3) / 1e12
            out['roofline'] = {'kernel': SWEEP_KERNEL if Np_ >= 4096 else SWEEP_KERNEL_SHORT, 'bound': 'mfma', 'achieved': ach,
                               'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / FP64_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                               'traffic_measured': False,      # a profile-time constant (see traffic_source), never collected in this run
                               'traffic_unit': 'bytes/launch',
                               'traffic_source': 'profile: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE) of this '
                                                 'launch geometry, committed as profiles/%s -- a profile-time '
                                                 'constant, NOT collected during this run' % (traffic_file or '(none for this geometry)'),
                               'launches': int(launches),
                               # the shader clock the kernel's own workgroups measured during these launches (s_memtime over
                               # s_memrealtime ticks, accumulated by the kernel): the peak above is quoted at 2400 MHz
                               'sclk_mhz': tm.get('sweep_sclk_mhz') or None,
                               'frac_at_measured_clock': (ach / (FP64_MFMA_PEAK_TFLOPS * tm['sweep_sclk_mhz'] / 2400.0))
                               if tm.get('sweep_sclk_mhz') else None,
                               'avg_launch_ms': tm['sweep_trmm'] / max(launches, 1),
                               'flop_per_launch': tm['sweep_trmm_flop'] / max(launches, 1)}
            out['traffic_measured'] = False           # roofline.traffic is read from profiles/, see roofline.traffic_source
        # the fit's two MFMA stages against the same peak (algorithmic N^3/3 flop each), for EVERY workload:
        # the replicated fit is what bounds strong scaling
        fit = {}
        for stage, label in (('cholesky', 'cholesky (k_chol_tg: persistent task-graph kernel with its nine shadow workgroups; a single 128-block: k_potrf16; option chol_tg = 0: k_potrf16 + k_panel_solve16 + '
                                          'k_row_update64 + k_syrk_update)'),
                             ('trtri', 'triangular inverse (k_trtri_gemm1/2)')):
            if tm[stage] > 0:
                ach = (float(N) ** 3 / 3.0) * args.steps / (tm[stage] * 1e-3) / 1e12
                fit[stage] = {'kernel': label, 'bound': 'mfma', 'achieved': ach, 'peak': FP64_MFMA_PEAK_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': ach / FP64_MFMA_PEAK_TFLOPS,
                              'ms': tm[stage] / args.steps, 'flop': float(N) ** 3 / 3.0}
        if fit.get('trtri') and tm.get('trtri_ahead', 0) > 0:
            # the inversion's leading part ran on a side stream behind the factorisation (option trtri_ahead): the span is
            # what was left AFTER the factor was done -- not the duration of N^3/3 flop; the pair is priced together below
            fit['trtri']['ahead_fits'] = int(tm['trtri_ahead'])
            fit['trtri']['note'] = ('ms = the part of the inversion left after the factorisation; its leading part overlapped '
                                    'the factorisation (frac would overstate the kernels: omitted)')
            for kk in ('achieved', 'frac'):
                fit['trtri'].pop(kk, None)
        if fit.get('cholesky') and fit.get('trtri'):
            ms = fit['cholesky']['ms'] + fit['trtri']['ms']
            ach = 2.0 * (float(N) ** 3 / 3.0) / (ms * 1e-3) / 1e12
            fit['factor_and_inverse'] = {'ms': ms, 'flop': 2.0 * float(N) ** 3 / 3.0, 'achieved': ach, 'unit': 'TFLOP/s',
                                         'peak': FP64_MFMA_PEAK_TFLOPS, 'frac': ach / FP64_MFMA_PEAK_TFLOPS}
        if fit:
            out['roofline_fit'] = fit
        if tm.get('rff_sweep', 0) > 0:
            # the Thompson sweep kernel: bound by the double-precision lanes, which its matrix instructions (the projection)
            # and its vector instructions (the cosine epilogue) SHARE: 16 lane-operations per clock per SIMD either way
            # (v_mfma_f64_16x16x4 = 1024 FMA in 64 clocks; a 64-lane f64 VALU instruction = 4 clocks).  Algorithmic work per
            # (draw, feature, candidate): d multiply-adds + 20 instructions of cos_cw and the weighted sum (counted in the ISA)
            lane_peak = 256 * 4 * 16 * 2.4e9
            ach = tm['rff_sweep_ops'] / (tm['rff_sweep'] * 1e-3)
            out['roofline_rff'] = {'kernel': 'k_rff_mfma5', 'bound': 'fp64 lanes (MFMA projection + VALU cosine share them)',
                                   'achieved': ach / 1e12, 'peak': lane_peak / 1e12, 'unit': 'T lane-operations/s',
                                   'frac': ach / lane_peak, 'ms': tm['rff_sweep'] / args.steps,
                                   'lane_ops_per_step': tm['rff_sweep_ops'] / args.steps,
                                   'work': 'draws x features x (d + 20) x candidates, features NOT padded (n = 100)',
                                   # this kernel is power-bound: the shader clock its own workgroups measured (s_memtime over
                                   # s_memrealtime) against the 2400 MHz the peak is quoted at
                                   'sclk_mhz': tm.get('rff_sclk_mhz') or None,
                                   'frac_at_measured_clock': (ach / (lane_peak * tm['rff_sclk_mhz'] / 2400.0))
                                   if tm.get('rff_sclk_mhz') else None,
                                   'traffic': None}
        if refine is not None:
            out['refine'] = refine
        if warm is not None:
            # work of the cache correction: N covariance evaluations per candidate per PASS (up to 8 appended
            # points share a pass); the kernel is fp64-VALU-issue bound like the cross-Gram
            passes = -(-warm['appends_per_step'] // 8)
            warm['rank1_cov_evals_per_s'] = float(N) * Ml * passes / (warm['stage_ms_per_step_rank0']['rank1'] * 1e-3) \
                if warm['stage_ms_per_step_rank0']['rank1'] > 0 else None
            if warm['rank1_cov_evals_per_s'] and w['kernel'] == 'se' and d == 8 and warm['appends_per_step'] == 1:
                # what bounds it: 42.4 VALU instructions per evaluation (SQ_INSTS_VALU, profiles/history/r03_pmc_covariance_kernels.txt:
                # a profile-time constant for SE-ARD at d = 8, q = 1) against the chip's 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
                warm['roofline'] = {'kernel': 'k_sweep_rankq<1, se>', 'bound': 'valu-issue',
                                    'achieved': warm['rank1_cov_evals_per_s'] * 42.4 / 1e12, 'peak': 256 * 4 * 16 * 2.4e9 / 1e12,
                                    'unit': 'T lane-instructions/s',
                                    'frac': warm['rank1_cov_evals_per_s'] * 42.4 / (256 * 4 * 16 * 2.4e9),
                                    'instructions_per_evaluation': 42.4}
            out['warm_step'] = warm
        if 'roofline' not in out and 'cholesky' in fit:      # no sweep GEMM in this workload (Thompson)
            out['roofline'] = dict(fit['cholesky'], traffic=None)
        if not args.no_cpu_baseline:
            # sample: the WHOLE sweep for N <= 2048 (config B: ~1 min of host time), otherwise SURVEY 8(d)'s 2^17
            # candidates (~100 s at N = 8192 on 128 BLAS threads) in TWO disjoint runs timed separately (linearity of the
            # extrapolation), labelled `extrapolated`.  With N > 1 ranks rank 0 times it (on its own shard) after the
            # process group has been left (the other ranks have exited).
            span = Ml if w['acq'] != 'thompson' else M
            nc = min(args.cpu_candidates or (M if N <= 2048 else (1 << 17)), span)
            draws = mine if w['acq'] == 'thompson' else []
            # the candidates the timed step ranked best over the WHOLE grid (merged over the ranks): the oracle evaluates them
            # too (untimed) and the parity record checks their ORDER
            tk_idx = [int(v) for v in np.asarray(best[1]).ravel() if v >= 0] if w['acq'] in ('ei', 'ucb') else []
            if world > 1 and os.environ.get('OMP_NUM_THREADS') == '1':
                out['cpu_baseline'], ref_vals = cpu_baseline_unpinned(w['name'], M, nc, span, draws, tk_idx)
            else:
                out['cpu_baseline'], ref_vals = cpu_baseline(w, nc, span, draws, tk_idx)
            # the same candidates on the device, outside any timed region: every bench line is also a parity check
            eng.fit_dev(dX.data_ptr(), N, d, dy.data_ptr(), w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
            dev_vals = {}
            sidx = ref_vals['index']
            if w['acq'] == 'thompson':
                # EVERY local draw of the step, on the candidates the oracle evaluated
                tri = [thompson_draw(w, s_) for s_ in mine]
                Wa, ba, za = (np.array([t_[i] for t_ in tri]) for i in range(3))
                th = eng.rff_posterior(Wa, ba, za, np.sqrt(2.0 * w['rho'] / 100))
                dev_vals['acq'] = eng.rff_sweep(Wa, ba, th, w['bias'], w['Xc'][sidx], k=0)['vals']
            else:
                param = eng.mean_at_obs()[1] if w['acq'] == 'ei' else ucb_beta(N)
                dsel = torch.from_numpy(np.ascontiguousarray(w['Xc'][lo_i + sidx])).to(dev)
                buf = torch.empty(3, len(sidx), dtype=torch.float64, device=dev)
                eng.sweep_dev(w['acq'], param, dsel.data_ptr(), len(sidx), 0, d_acq=buf[0].data_ptr(),
                              d_mu=buf[1].data_ptr(), d_s2=buf[2].data_ptr())
                eng.sync()
                hb = buf.cpu().numpy()
                dev_vals.update(acq=hb[0], mu=hb[1], s2=hb[2], target=float(param))
            if w['acq'] in ('ei', 'ucb'):
                dev_vals['top_val'], dev_vals['top_idx'] = np.asarray(best[0]).ravel(), np.asarray(best[1]).ravel()
                dev_vals['sample_global_index'] = lo_i + sidx
            out['parity'] = parity_record(w, ref_vals, dev_vals)
            if w['name'] == 'ns':
                # The headline workload's winner is Sobol' point 1 (the box centre IS the optimum): trivially right.  The same
                # line therefore carries -- untimed -- the selection record of the variant 'ns2' (optimum off the centre): the
                # device's top-k over the whole grid against the oracle's order.  The headline and its timing stay 'ns'.
                out['parity']['nontrivial'] = nontrivial_selection_record(eng, dev, M, k, args.cpu_candidates)
        if args.plugin_steps > 0 and world == 1 and w['acq'] in ('ei', 'ucb'):
            out['plugin_step'] = plugin_step(w, args.plugin_steps, k)
        print(json.dumps(out))


if __name__ == '__main__':
    main()
