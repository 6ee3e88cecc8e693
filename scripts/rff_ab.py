"""A/B of the Thompson sweep kernels (option x_rff: 0 = double-buffered, 1 = round 3) on configs D and E: python scripts/rff_ab.py"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl in ('d', 'e'):
    for v in (1, 0, 1, 0):      # 1 = the round-3 kernel, 0 = the default (round 5)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', wl, '--steps', '4', '--warmup', '1',
                              '--no-cpu-baseline', '--opt', 'x_rff=%d' % v], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            r = d['roofline_rff']
            print('config %s x_rff=%d: step %.3f ms, k_rff_mfma %.3f ms/step, frac %.3f, cholesky %.3f ms, selected %s' % (
                wl, v, d['ms_per_step'], r['ms'], r['frac'], d['stage_ms_per_step_rank0']['cholesky'], d['selected']), flush=True)
        except Exception as exc:
            print('config %s x_rff=%d FAILED: %r %s' % (wl, v, exc, out.stderr[-400:]))
