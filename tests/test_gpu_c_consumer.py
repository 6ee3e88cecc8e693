"""The drop-in boundary from C, with DEVICE work: tests/c/abi_sweep.c is compiled as strict C99
against include/gpx.h, linked with libgpx.so, run on a small problem, and its printed results are compared with the CPU
oracle -- once through gpx_create -> gpx_fit -> gpx_mean_at_obs -> gpx_sweep -> gpx_get_vectors, once through the sharded
listing of INTEGRATION.md section 3 with a one-rank RCCL communicator (gpx_comm_* + gpx_topk_allgather)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from oracle import gp_ref
from helpers import mu_tol, s2_tol  # noqa: F401  (the ladder lives in helpers)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build(tmp_path):
    gcc = shutil.which('gcc')
    assert gcc is not None, 'no C compiler on the GPU box'
    libdir = os.path.join(ROOT, 'pybo_amd', 'csrc')
    exe = tmp_path / 'abi_sweep'
    subprocess.check_call([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-O1',
                           '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'c', 'abi_sweep.c'),
                           '-o', str(exe), '-L', libdir, '-lgpx', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    return exe


def _problem(tmp_path, N=700, d=4, M=6000, k=12, seed=3):
    rng = np.random.RandomState(seed)
    X = rng.rand(N, d)
    y = np.sin(3 * X.sum(1)) + 0.05 * rng.randn(N)
    ell, rho, sn2, bias = 0.3 + 0.1 * rng.rand(d), 1.3, 1e-3, 0.2
    Xc = rng.rand(M, d)
    path = tmp_path / 'problem.bin'
    with open(path, 'wb') as fh:
        fh.write(struct.pack('<4q', N, d, M, k))
        for arr in (X, y, ell, np.array([rho, sn2, bias]), Xc):
            fh.write(np.ascontiguousarray(arr, dtype='<f8').tobytes())
    return path, (X, y, ell, rho, sn2, bias, Xc, k)


def _parse(text):
    out = {'top': [], 'acq': {}, 'alpha': {}}
    for line in text.splitlines():
        w = line.split()
        if w[0] == 'target':
            out['target'] = float(w[1])
        elif w[0] == 'top':
            out['top'].append((int(w[2]), float(w[3])))
        elif w[0] in ('acq', 'alpha'):
            out[w[0]][int(w[1])] = float(w[2])
    return out


@pytest.mark.parametrize('mode', ['plain', 'comm'])
def test_c_program_fits_sweeps_and_matches_the_oracle(tmp_path, mode):
    exe = _build(tmp_path)
    path, (X, y, ell, rho, sn2, bias, Xc, k) = _problem(tmp_path)
    args = [str(exe), str(path)] + (['comm'] if mode == 'comm' else [])
    res = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()
    got = _parse(res.stdout.decode())
    ref = gp_ref.make_gp(sn2, rho, ell, bias)
    ref.add_data(X, y)
    target = ref.mean_at_obs().max()
    assert abs(got['target'] - target) <= 1e-9 * np.sqrt(rho) + 1e-6 * abs(target)
    want = ref.get_improvement(got['target'], Xc)
    # every printed acquisition value, at the EI tolerance of the ladder where EI is live
    for j, v in got['acq'].items():
        assert abs(v - want[j]) <= 1e-6 * abs(want[j]) + 1e-12 * want.max(), (j, v, want[j])
    # the selection: same indices in the same order wherever the oracle's gap exceeds the tolerance, values at 1e-6
    order = gp_ref.topk_desc(want, k)
    idx = [i for i, _ in got['top']]
    assert idx[0] == order[0]
    gaps = np.abs(np.diff(want[order])) / np.abs(want[order[:-1]])
    if np.all(gaps > 1e-5):
        assert idx == list(order)
    for i, v in got['top']:
        assert abs(v - want[i]) <= 1e-6 * abs(want[i])
    # alpha = (K + sn2 I)^-1 (y - bias)
    alpha = ref.alpha()
    for j, v in got['alpha'].items():
        assert abs(v - alpha[j]) <= 1e-8 * np.abs(alpha).max(), (j, v, alpha[j])
