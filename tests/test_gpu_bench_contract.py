"""The bench.py contract the driver relies on, end to end on the device at a reduced candidate count: ONE JSON line
with the metric / value / roofline / cpu_baseline fields (single rank), and the N > 1 launch path through
torch.distributed.run (two gloo ranks sharing the one GPU of the box: same selected candidate as the single rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--workload', 'b', '--candidates', '131072', '--steps', '2', '--warmup', '1', '--warm-steps', '2']


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(cmd):
    out = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                         stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    # (the gloo transport of the two-rank run prints its own "[Gloo] Rank ..." lines; bench.py prints ONE line)
    lines = [l for l in out.stdout.decode().splitlines() if l.strip() and not l.startswith('[Gloo]')]
    assert len(lines) == 1, lines                      # exactly one line on stdout, and it is JSON
    return json.loads(lines[0])


def test_single_rank_line_has_the_contract_fields():
    o = _line([sys.executable, 'bench.py'] + COMMON + ['--cpu-candidates', '8192'])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in o, key
    assert o['n_gpus'] == 1 and o['steps'] == 2 and o['warmup'] == 1 and o['dtype'] == 'f64' and o['unit'] == 'steps/s'
    assert abs(o['value'] * o['ms_per_step'] / 1e3 - 1.0) < 1e-9 and o['higher_is_better'] is True
    assert o['vs_baseline'] is None and 'workload' in o['config'] and 'model' not in o['config']
    r = o['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert 0.05 < r['frac'] < 1.0
    c = o['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert {'cholesky', 'trtri'} <= set(o['roofline_fit']) and o['warm_step']['ms_per_step'] > 0
    assert o['selected']['index'] >= 0


def test_two_ranks_through_torch_distributed_run_select_the_same_candidate():
    one = _line([sys.executable, 'bench.py'] + COMMON + ['--no-cpu-baseline', '--no-refine'])
    two = _line([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                 '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--backend', 'gloo',
                 '--share-device', '0'] + COMMON + ['--no-cpu-baseline', '--no-refine'])
    assert two['n_gpus'] == 2 and two['scaling'] == 'strong'
    assert two['selected'] == one['selected']          # bit-identical merged top-1 (value and global index)
    assert two['warm_step']['selected'] == one['warm_step']['selected']


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """The driver's command shape is literally `python3 bench.py --gpus N` (BENCH_r02.json.cmd): bench.py must start
    the N ranks itself.  Dry run on the one GPU of the box: gloo + --share-device 0.  One JSON line, n_gpus = 2, the
    single-rank selection, and the N > 1 line carries roofline, cpu_baseline and parity too."""
    one = _line([sys.executable, 'bench.py'] + COMMON + ['--no-cpu-baseline', '--no-refine', '--plugin-steps', '0'])
    two = _line([sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--share-device', '0'] + COMMON +
                ['--no-refine', '--cpu-candidates', '8192'])
    assert two['n_gpus'] == 2 and two['selected'] == one['selected']
    assert two['warm_step']['selected'] == one['warm_step']['selected']
    assert two['roofline']['frac'] > 0.05 and two['cpu_baseline']['value'] > 0
    assert two['parity']['selected_index_matches'] and two['parity']['moments_within_stated_tolerance']


def test_single_rank_line_reports_parity_and_the_plugin_level_step():
    o = _line([sys.executable, 'bench.py'] + COMMON + ['--cpu-candidates', '8192', '--no-refine', '--plugin-steps', '3'])
    p = o['parity']
    assert p['n_compared'] == 8192 and p['selected_index_matches'] and p['moments_within_stated_tolerance']
    assert p['max_rel_mu'] < 1e-6 and p['max_rel_s2'] < 1e-6 and p['max_rel_acq'] < 1e-6
    s = o['plugin_step']
    assert s['iterations'] == 3 and s['cold_ms'] > 0 and len(s['warm_ms_each']) == 2 and s['warm_ms'] > 0
