"""
ctypes binding of libgpx.so (include/gpx.h).  This is the ONLY way pybo_amd reaches the GP arithmetic:
there is no CPU fallback -- if the shared library is missing, or no HIP device is present, the first use
raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

# A handle owns up to four HIP streams (its own + three side streams: the factorisation's lookahead, and the correction
# pass of an ANNOUNCED observation that runs beside the recommender's calls).  ROCm multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues, 4 by default: with a second live handle (an ensemble, a second model, torch's own
# streams) two streams of one handle can land on the same queue and what was meant to overlap runs back to back --
# measured: the warm plug-in iteration 14.0 -> 16.3 ms at N = 8192 merely because another handle existed
# (profiles/history/).  Eight queues keep two handles apart.  The variable is read by the HIP runtime when it
# initialises, so it is the APPLICATION's to set (bench.py does, before importing torch): a
# library import does not change its host's environment -- see hw_queues_hint() below.


def hw_queues_hint():
    """None, or a one-line warning when this process runs several device handles on the runtime's default of 4 hardware
    queues (GPU_MAX_HW_QUEUES unset or < 8): streams meant to overlap may then share a queue."""
    try:
        q = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        q = 4
    if q >= 8:
        return None
    return ('pybo_amd: several device handles on GPU_MAX_HW_QUEUES=%d hardware queues; export GPU_MAX_HW_QUEUES=8 before '
            'the HIP runtime starts to keep their streams apart (see INTEGRATION.md)' % q)


_HERE = os.path.dirname(os.path.abspath(__file__))
# libgpx.so is the shipping library.  GPX_DIAGNOSTICS=1 selects libgpx_diag.so (the same objects, gpx_set_option also accepts the
# diagnostic knobs of csrc/gpx_diag.h: what the test-suite and the A/B scripts drive); GPX_LIB_PATH: any other build of the ABI.
LIB_PATH = os.environ.get('GPX_LIB_PATH') or os.path.join(
    _HERE, 'csrc', 'libgpx_diag.so' if os.environ.get('GPX_DIAGNOSTICS', '0') not in ('', '0') else 'libgpx.so')

# every symbol include/gpx.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int64)
_i64 = C.c_int64
_dbl = C.c_double
# every symbol pybo_amd/csrc/gpx_diag.h declares (exported by both builds)
DIAG_SYMBOLS = {
    'gpx_chol_trace': (_i64, [_P, _P, _i64]),
    'gpx_chol_tasks2': (_i64, [C.c_int, C.c_int, _P, _i64, _P]),
}
SYMBOLS = {
    'gpx_create': (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    'gpx_destroy': (C.c_int, [_P]),
    'gpx_last_error': (C.c_char_p, [_P]),
    'gpx_version': (C.c_int, []),
    'gpx_set_option': (C.c_int, [_P, C.c_char_p, _i64]),
    'gpx_fit': (C.c_int, [_P, _P, _i64, _i64, _P, C.c_int, _P, _dbl, _dbl, _dbl]),
    'gpx_fit_dev': (C.c_int, [_P, _P, _i64, _i64, _P, C.c_int, _P, _dbl, _dbl, _dbl]),
    'gpx_loglik': (C.c_int, [_P, _P]),
    'gpx_loglik_batch': (C.c_int, [_P, _i64, _P, _P]),
    'gpx_append': (C.c_int, [_P, _P, _dbl]),
    'gpx_append_begin': (C.c_int, [_P, _P]),
    'gpx_fail_pivot': (_i64, [_P]),
    'gpx_get_matrix': (C.c_int, [_P, C.c_int, _P]),
    'gpx_get_vectors': (C.c_int, [_P, _P, _P]),
    'gpx_fit_stage': (C.c_int, [_P, _P, _i64, _i64, _P, C.c_int, _P, _dbl, _dbl, _dbl, C.c_int]),
    'gpx_mean_at_obs': (C.c_int, [_P, _P, _P]),
    'gpx_var_at_obs': (C.c_int, [_P, _P]),
    'gpx_capacity': (_i64, [_P]),
    'gpx_predict': (C.c_int, [_P, _P, _i64, _P, _P, _P, _P]),
    'gpx_predict_mean': (C.c_int, [_P, _P, _i64, _P, _P]),
    'gpx_sweep': (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _i64, _i64, _P, _P, _P, _P, _P]),
    'gpx_sweep_dev': (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _i64, _i64, _P, _P, _P, _P, _P]),
    'gpx_sweep_update': (C.c_int, [_P, C.c_int, _P, C.c_int, _i64, _P, _P, _P, _P, _P]),
    'gpx_sweep_update_dev': (C.c_int, [_P, C.c_int, _P, C.c_int, _i64, _P, _P, _P, _P, _P]),
    'gpx_sweep_cache_size': (_i64, [_P]),
    'gpx_rff_sweep': (C.c_int, [_P, _P, _P, _P, _i64, _i64, _i64, _dbl, _P, _i64, _i64, _P, _P, _P]),
    'gpx_rff_sweep_dev': (C.c_int, [_P, _P, _P, _P, _i64, _i64, _i64, _dbl, _P, _i64, _i64, _P, _P, _P]),
    'gpx_rff_grad': (C.c_int, [_P, _P, _P, _P, _i64, _i64, _dbl, _P, _i64, _P, _P]),
    'gpx_rff_gram': (C.c_int, [_P, _P, _P, _i64, _P, _P]),
    'gpx_rff_gram_batch': (C.c_int, [_P, _P, _P, _i64, _i64, _P, _P]),
    'gpx_rff_posterior': (C.c_int, [_P, _P, _P, _P, _i64, _i64, _dbl, _P]),
    'gpx_ensemble_sweep': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, _i64, _i64, _P, _P, _P, _P, _P]),
    'gpx_ensemble_sweep_dev': (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, _P, _i64, _i64, _P, _P, _P, _P, _P]),
    'gpx_ensemble_predict': (C.c_int, [_P, C.c_int, _P, _i64, _P, _P, _P, _P]),
    'gpx_grid_create': (C.c_int, [C.c_int, C.c_int, _P, _i64, _i64, C.c_uint64, _i64, _P, C.c_int, C.POINTER(_P)]),
    'gpx_grid_data': (_P, [_P]),
    'gpx_grid_rows': (C.c_int, [_P, _P, _i64, _P]),
    'gpx_grid_destroy': (C.c_int, [_P]),
    'gpx_comm_unique_id': (C.c_int, [_P]),
    'gpx_comm_init': (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    'gpx_comm_destroy': (C.c_int, [_P]),
    'gpx_comm_size': (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'gpx_comm_last_error': (C.c_char_p, []),
    'gpx_topk_allgather': (C.c_int, [_P, _i64, _i64, _i64, _P, _P]),
    'gpx_timers': (C.c_int, [_P, _P, C.c_int, C.c_int]),
    'gpx_sync': (C.c_int, [_P]),
    'gpx_diagnostics': (C.c_int, []),
}

GPX_OK, GPX_EARG, GPX_ENOTPD, GPX_EHIP, GPX_EOOM, GPX_ESTATE, GPX_ERCCL = 0, -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
KERNELS = {'se': 0, 'matern5': 1, 'matern3': 2, 'matern1': 3}
ACQ = {'ei': 0, 'pi': 1, 'ucb': 2, 'mean': 3}
TIMER_NAMES = ['gram', 'cholesky', 'trtri', 'alpha', 'cross_gram', 'sweep_trmm', 'acq_topk', 'rff',
               'sweep_trmm_launches', 'sweep_trmm_flop', 'copies', 'append', 'rank1', 'rff_sweep', 'rff_sweep_ops',
               'chol_fallbacks', 'sweep_sclk_mhz', 'rff_sclk_mhz', 'trtri_ahead']
TOPK_MAX = 4096

_lib = None


def chol_tasks(nblocks, chunks=0):
    """The task lists of the task-graph factorisation (host only): two (n, 8) int16 arrays
    {type, I, J, k0, k1, ordinal, aux, reserved} -- the critical list and the workers' list (csrc/gpx_diag.h: gpx_chol_tasks2)."""
    lib = load()
    counts = np.zeros(2, dtype=np.int64)
    tot = lib.gpx_chol_tasks2(nblocks, chunks, None, 0, _ptr(counts))
    if tot < 0:
        raise ValueError('gpx_chol_tasks2: bad arguments')
    out = np.zeros((max(int(tot), 1), 8), dtype=np.int16)
    lib.gpx_chol_tasks2(nblocks, chunks, _ptr(out), int(tot), _ptr(counts))
    o = np.cumsum(np.r_[0, counts])
    return [out[o[q]:o[q + 1]] for q in range(2)]


class GpxError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'libgpx error %d: %s' % (code, msg))
        self.code = code


def load():
    """Load libgpx.so (once) and set the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('pybo_amd: %s not found -- build it with `python __graft_entry__.py` '
                              '(or pybo_amd/csrc/build.sh); there is no CPU fallback.' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(DIAG_SYMBOLS.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def sobol_direction_numbers(d, nbits):
    """Direction numbers (d, bits) uint32 of scipy's unscrambled Sobol' generator (Joe-Kuo), LSB-first in the
    Gray-code recurrence: point i = XOR_b [bit b of i^(i>>1)] * sv[:, b], x = point * 2^-bits.  Read from the
    generator object when it exposes them, otherwise derived through the public API (the point with Gray
    code 2^b is number 2^(b+1) - 1); checked against the generator's first points either way."""
    from scipy.stats import qmc
    eng = qmc.Sobol(d, scramble=False)
    bits = int(getattr(eng, 'bits', 30) or 30)
    sv = getattr(eng, '_sv', None)
    if sv is not None and np.shape(sv) == (d, bits):
        sv = np.ascontiguousarray(sv, dtype=np.uint32)
    else:
        sv = np.zeros((d, bits), dtype=np.uint32)
        pos = 0
        for b in range(min(bits, max(int(nbits), 4))):
            target = 2 ** (b + 1) - 1
            if target > pos:
                eng.fast_forward(target - pos)
            sv[:, b] = np.round(eng.random(1)[0] * 2.0 ** bits).astype(np.uint32)
            pos = target + 1
    ref = qmc.Sobol(d, scramble=False).random(16)
    acc = np.zeros(d, dtype=np.uint32)
    for i in range(16):
        g = i ^ (i >> 1)
        acc[:] = 0
        for b in range(5):
            if (g >> b) & 1:
                acc ^= sv[:, b]
        if not np.array_equal(acc * 2.0 ** -bits, ref[i]):
            raise RuntimeError('scipy Sobol generator does not follow the expected Gray-code recurrence')
    return sv, bits


class DeviceGrid(object):
    """A candidate grid generated and kept in HBM (gpx_grid_*): `kind` 'sobol' (points first..first+n-1 of
    scipy's unscrambled sequence, scaled to the box) or 'uniform' (Philox4x32-10 keyed by `seed`).  Behaves
    like a read-only (n, d) array for the solver: len(), .shape, grid[idx] (rows copied to the host),
    np.asarray(grid) (the whole grid -- avoid on the hot path)."""

    def __init__(self, kind, bounds, n, seed=0, first=0, device=0):
        self._lib = load()
        bounds = _f64(np.array(bounds, dtype=float, ndmin=2))
        n = int(n)
        d = len(bounds)
        g = _P()
        if kind == 'sobol':
            sv, bits = sobol_direction_numbers(d, int(first + n).bit_length())
            rc = self._lib.gpx_grid_create(int(device), 1, _ptr(bounds), n, d, 0, int(first),
                                           sv.ctypes.data_as(_P), bits, C.byref(g))
        elif kind == 'uniform':
            rc = self._lib.gpx_grid_create(int(device), 0, _ptr(bounds), n, d, int(seed), int(first), None, 0,
                                           C.byref(g))
        else:
            raise ValueError("grid kind must be 'sobol' or 'uniform'")
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_last_error(None) or b'').decode())
        self._g = g
        self.kind, self.bounds, self.device = kind, bounds, int(device)
        self.shape = (n, d)
        self.ptr = self._lib.gpx_grid_data(g)

    def __len__(self):
        return self.shape[0]

    def rows(self, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        out = np.empty((len(idx), self.shape[1]))
        rc = self._lib.gpx_grid_rows(self._g, _ptr(idx), len(idx), _ptr(out))
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_last_error(None) or b'').decode())
        return out

    def view(self, lo, hi):
        """Rows [lo, hi) as a grid of their own, without a copy; the same object for the same range (a model keys
        its warm sweep cache on the grid object it swept)."""
        lo, hi = int(lo), int(hi)
        if not 0 <= lo <= hi <= len(self):
            raise IndexError('grid view out of range')
        views = self.__dict__.setdefault('_views', {})
        if (lo, hi) not in views:
            views[(lo, hi)] = DeviceGridView(self, lo, hi)
        return views[(lo, hi)]

    def __getitem__(self, idx):
        if isinstance(idx, slice):                  # grid[lo:hi] (pybo_amd.dist shards a grid this way)
            lo, hi, step = idx.indices(len(self))
            if step == 1:
                return self.view(lo, max(lo, hi))   # contiguous: stays on the device
            return self.rows(np.arange(lo, hi, step, dtype=np.int64))
        if np.isscalar(idx):
            return self.rows([idx])[0]
        return self.rows(idx)

    def __array__(self, dtype=None, copy=None):
        out = np.empty(self.shape)
        rc = self._lib.gpx_grid_rows(self._g, None, 0, _ptr(out))
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_last_error(None) or b'').decode())
        return out if dtype is None else out.astype(dtype)

    def close(self):
        if getattr(self, '_g', None):
            self._lib.gpx_grid_destroy(self._g)
            self._g = None
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceGridView(DeviceGrid):
    """Rows [lo, hi) of a DeviceGrid as a grid of their own (no copy: a pointer into the parent's HBM array) -- what
    a rank of the multi-process layout sweeps when every rank holds the whole grid (pybo_amd.dist.sharded_topk)."""

    def __init__(self, parent, lo, hi):
        self._lib = parent._lib
        self._parent, self._lo = parent, int(lo)
        self._g = None
        self.kind, self.bounds, self.device = parent.kind, parent.bounds, parent.device
        self.shape = (int(hi) - int(lo), parent.shape[1])
        self.ptr = parent.ptr + int(lo) * parent.shape[1] * 8

    def rows(self, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        if len(idx) and (idx.min() < 0 or idx.max() >= len(self)):
            raise IndexError('grid row out of range')
        return self._parent.rows(idx + self._lo)

    def __array__(self, dtype=None, copy=None):
        out = self._parent.rows(np.arange(self._lo, self._lo + len(self), dtype=np.int64))
        return out if dtype is None else out.astype(dtype)

    def close(self):
        self.ptr = None


class ShardedDeviceGrid(object):
    """ONE logical (n, d) grid laid out over several devices for pybo_amd.models.ShardedGP: shard p = rows
    shard_bounds(n, p, P) generated and kept on devices[p] (Sobol': the same points of the sequence; uniform: the same
    Philox stream -- the concatenation of the shards IS the single-device grid, bit for bit).  Behaves like a
    read-only (n, d) array for the solver: len(), .shape, grid[idx] (rows fetched from their owners)."""

    def __init__(self, kind, bounds, n, devices, seed=0, first=0):
        n, devices = int(n), [int(dv) for dv in devices]
        P = len(devices)
        self.kind, self.devices = kind, devices
        self.shards = []
        for p, dv in enumerate(devices):
            lo, hi = (n * p) // P, (n * (p + 1)) // P
            g = DeviceGrid(kind, bounds, hi - lo, seed=seed, first=int(first) + lo, device=dv) if hi > lo else None
            self.shards.append((lo, hi, g))
        self.bounds = np.array(bounds, dtype=float, ndmin=2)
        self.shape = (n, len(self.bounds))

    def __len__(self):
        return self.shape[0]

    def rows(self, idx):
        idx = np.ascontiguousarray(np.atleast_1d(idx), dtype=np.int64)
        out = np.empty((len(idx), self.shape[1]))
        if len(idx) and (idx.min() < 0 or idx.max() >= len(self)):
            raise IndexError('grid row out of range')
        for lo, hi, g in self.shards:
            mine = np.flatnonzero((idx >= lo) & (idx < hi))
            if len(mine):
                out[mine] = g.rows(idx[mine] - lo)
        return out

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return self.rows(np.arange(*idx.indices(len(self)), dtype=np.int64))
        if np.isscalar(idx):
            return self.rows([idx])[0]
        return self.rows(idx)

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate([np.asarray(g) for _, _, g in self.shards if g is not None])
        return out if dtype is None else out.astype(dtype)

    def close(self):
        for _, _, g in self.shards:
            if g is not None:
                g.close()


class Comm(object):
    """RCCL communicator bound to an Engine's device and stream (gpx_comm_*): the exchange step of the sharded
    sweep without torch.distributed.  Rank 0 creates the id with `Comm.unique_id()` and hands the 128 bytes to
    the other ranks over any side channel; every rank then constructs `Comm(engine, rank, nranks, id)`."""

    @staticmethod
    def unique_id():
        lib = load()
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        rc = lib.gpx_comm_unique_id(C.cast(buf, _P))
        if rc != GPX_OK:
            raise GpxError(rc, (lib.gpx_comm_last_error() or b'').decode())
        return bytes(buf)

    def __init__(self, engine, rank, nranks, uid):
        self._lib = load()
        if len(uid) != COMM_ID_BYTES:
            raise ValueError('the communicator id has %d bytes' % COMM_ID_BYTES)
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        c = _P()
        rc = self._lib.gpx_comm_init(engine._h, int(rank), int(nranks), C.cast(buf, _P), C.byref(c))
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_comm_last_error() or b'').decode())
        self._c, self._engine = c, engine            # the engine must outlive the communicator
        self.rank, self.nranks = int(rank), int(nranks)

    def topk_allgather(self, n, index_offset, k):
        """Gather the engine's last n device (value, index) pairs from every rank; k > 0: merged k best
        (identical on every rank), k = 0: all nranks*n pairs in rank order."""
        m = int(k) if k > 0 else int(n) * self.nranks
        tv = np.empty(m)
        ti = np.empty(m, dtype=np.int64)
        rc = self._lib.gpx_topk_allgather(self._c, int(n), int(index_offset), int(k), _ptr(tv), _ptr(ti))
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_comm_last_error() or b'').decode())
        return tv, ti

    def close(self):
        if getattr(self, '_c', None):
            self._lib.gpx_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine(object):
    """One gpx handle = one (device, model).  Thin, stateless-in-Python wrapper over the C-ABI."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        h = _P()
        rc = self._lib.gpx_create(int(device), _P(stream) if stream else None, C.byref(h))
        if rc != GPX_OK:
            raise GpxError(rc, (self._lib.gpx_last_error(None) or b'').decode())
        self._h = h
        self.device = int(device)
        self.N = 0
        self.d = 0
        Engine._live += 1
        if Engine._live > 1 and not Engine._hinted:
            hint = hw_queues_hint()
            Engine._hinted = True
            if hint:
                import warnings
                warnings.warn(hint, RuntimeWarning, stacklevel=2)

    _live = 0            # handles alive in this process
    _hinted = False      # the GPU_MAX_HW_QUEUES hint is given once

    def close(self):
        if getattr(self, '_h', None):
            self._lib.gpx_destroy(self._h)
            self._h = None
            Engine._live -= 1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != GPX_OK:
            msg = (self._lib.gpx_last_error(self._h) or b'').decode()
            if rc == GPX_ENOTPD:
                raise np.linalg.LinAlgError(msg)
            raise GpxError(rc, msg)

    def set_option(self, name, value):
        self._check(self._lib.gpx_set_option(self._h, name.encode(), int(value)))

    # -- fit -----------------------------------------------------------------------------------
    def fit(self, X, y, kernel, ell, rho, sn2, bias, stage=3):
        X = _f64(X)
        if X.ndim != 2:
            raise ValueError('X must be (N, d)')
        y = _f64(y).reshape(-1)
        N, d = X.shape
        if len(y) != N:
            raise ValueError('X and y disagree on N')
        ell = _f64(np.broadcast_to(np.asarray(ell, dtype=float), (d,)))
        kid = KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        self.N, self.d = N, d
        if stage == 3:
            rc = self._lib.gpx_fit(self._h, _ptr(X), N, d, _ptr(y), kid, _ptr(ell), rho, sn2, bias)
        else:
            rc = self._lib.gpx_fit_stage(self._h, _ptr(X), N, d, _ptr(y), kid, _ptr(ell), rho, sn2, bias,
                                         int(stage))
        self._check(rc)
        self.N, self.d = N, d

    def fit_dev(self, dX_ptr, N, d, dy_ptr, kernel, ell, rho, sn2, bias):
        ell = _f64(np.broadcast_to(np.asarray(ell, dtype=float), (d,)))
        kid = KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        self._check(self._lib.gpx_fit_dev(self._h, _P(dX_ptr), N, d, _P(dy_ptr), kid, _ptr(ell), rho, sn2,
                                          bias))
        self.N, self.d = N, d

    def loglik(self):
        out = C.c_double()
        self._check(self._lib.gpx_loglik(self._h, C.byref(out)))
        return out.value

    def loglik_batch(self, hypers):
        """log p(y | X, theta_b) for B hyper-parameter vectors [sn2, rho, ell_1..d, bias] on the resident data: one
        batched launch chain, one synchronisation; the engine's own fit is untouched."""
        hypers = _f64(hypers).reshape(-1, self.d + 3)
        out = np.empty(len(hypers))
        self._check(self._lib.gpx_loglik_batch(self._h, len(hypers), _ptr(hypers), _ptr(out)))
        return out

    def append(self, x, y):
        """Rank-1 extension by one observation; returns False when a refit is needed (block boundary)."""
        x = _f64(x).reshape(-1)
        if len(x) != self.d:
            raise ValueError('x must have %d coordinates' % self.d)
        rc = self._lib.gpx_append(self._h, _ptr(x), float(y))
        if rc == GPX_ESTATE:
            return False
        self._check(rc)
        self.N += 1
        return True

    def append_begin(self, x):
        """Announce the next observation's location (gpx_append_begin): the value-independent work of the coming
        append runs ahead.  Returns False when there is nothing to run ahead (no live sweep cache, block boundary)."""
        x = _f64(x).reshape(-1)
        if len(x) != self.d:
            raise ValueError('x must have %d coordinates' % self.d)
        rc = self._lib.gpx_append_begin(self._h, _ptr(x))
        if rc == GPX_ESTATE:
            return False
        self._check(rc)
        return True

    def fail_pivot(self):
        return int(self._lib.gpx_fail_pivot(self._h))

    def get_matrix(self, which):
        out = np.empty((self.N, self.N))
        self._check(self._lib.gpx_get_matrix(self._h, {'L': 0, 'T': 1, 'K': 2}[which], _ptr(out)))
        return out

    def get_vectors(self):
        a = np.empty(self.N)
        alpha = np.empty(self.N)
        self._check(self._lib.gpx_get_vectors(self._h, _ptr(a), _ptr(alpha)))
        return a, alpha

    def mean_at_obs(self):
        mu = np.empty(self.N)
        mx = C.c_double()
        self._check(self._lib.gpx_mean_at_obs(self._h, _ptr(mu), C.byref(mx)))
        return mu, mx.value

    def var_at_obs(self):
        s2 = np.empty(self.N)
        self._check(self._lib.gpx_var_at_obs(self._h, _ptr(s2)))
        return s2

    def capacity(self):
        """Rows the handle's factor buffers are allocated for (they stay allocated until the handle is closed)."""
        return int(self._lib.gpx_capacity(self._h))

    # -- posterior / sweep -----------------------------------------------------------------------
    def predict(self, Xc, grad=False):
        Xc = _f64(Xc).reshape(-1, self.d)
        M = len(Xc)
        mu = np.empty(M)
        s2 = np.empty(M)
        if grad:
            dmu = np.empty((M, self.d))
            ds2 = np.empty((M, self.d))
            self._check(self._lib.gpx_predict(self._h, _ptr(Xc), M, _ptr(mu), _ptr(s2), _ptr(dmu), _ptr(ds2)))
            return mu, s2, dmu, ds2
        self._check(self._lib.gpx_predict(self._h, _ptr(Xc), M, _ptr(mu), _ptr(s2), None, None))
        return mu, s2

    def predict_mean(self, Xc, grad=False):
        """Posterior mean [and its gradient] only, k(x, X).alpha: no pass over the factor's inverse per call."""
        Xc = _f64(Xc).reshape(-1, self.d)
        M = len(Xc)
        mu = np.empty(M)
        dmu = np.empty((M, self.d)) if grad else None
        self._check(self._lib.gpx_predict_mean(self._h, _ptr(Xc), M, _ptr(mu), _ptr(dmu) if grad else None))
        return (mu, dmu) if grad else mu

    def sweep(self, acq, param, Xc, k=0, want_all=True, want_moments=False):
        """Host-buffer sweep.  Returns dict(top_val, top_idx, acq, mu, s2)."""
        Xc = _f64(Xc)
        Xc = Xc.reshape(-1, self.d) if self.d else np.atleast_2d(Xc)
        M = len(Xc)
        aid = ACQ[acq] if isinstance(acq, str) else int(acq)
        params = _f64([0.0 if param is None else param])
        tv = np.empty(k)
        ti = np.empty(k, dtype=np.int64)
        out = np.empty(M) if want_all else None
        mu = np.empty(M) if want_moments else None
        s2 = np.empty(M) if want_moments else None
        self._check(self._lib.gpx_sweep(self._h, aid, _ptr(params), 1, _ptr(Xc), M, k,
                                        _ptr(tv) if k else None, _ptr(ti) if k else None, _ptr(out),
                                        _ptr(mu), _ptr(s2)))
        return dict(top_val=tv, top_idx=ti, acq=out, mu=mu, s2=s2)

    def sweep_update(self, acq, param, k=0, want_all=True, want_moments=False):
        """Re-score the cached candidate set (option sweep_cache, kept current by append): O(M)."""
        M = self.sweep_cache_size()
        aid = ACQ[acq] if isinstance(acq, str) else int(acq)
        params = _f64([0.0 if param is None else param])
        tv = np.empty(k)
        ti = np.empty(k, dtype=np.int64)
        out = np.empty(M) if want_all else None
        mu = np.empty(M) if want_moments else None
        s2 = np.empty(M) if want_moments else None
        self._check(self._lib.gpx_sweep_update(self._h, aid, _ptr(params), 1, k, _ptr(tv) if k else None,
                                               _ptr(ti) if k else None, _ptr(out), _ptr(mu), _ptr(s2)))
        return dict(top_val=tv, top_idx=ti, acq=out, mu=mu, s2=s2)

    def sweep_cache_size(self):
        return int(self._lib.gpx_sweep_cache_size(self._h))

    def sweep_dev(self, acq, param, dXc_ptr, M, k, d_acq=None, d_mu=None, d_s2=None):
        aid = ACQ[acq] if isinstance(acq, str) else int(acq)
        params = _f64([0.0 if param is None else param])
        tv = np.empty(k)
        ti = np.empty(k, dtype=np.int64)
        self._check(self._lib.gpx_sweep_dev(self._h, aid, _ptr(params), 1, _P(dXc_ptr), M, k,
                                            _ptr(tv) if k else None, _ptr(ti) if k else None,
                                            _P(d_acq) if d_acq else None, _P(d_mu) if d_mu else None,
                                            _P(d_s2) if d_s2 else None))
        return tv, ti

    @staticmethod
    def ensemble_sweep(engines, acq, param, Xc, k=0, want_all=True, want_moments=False):
        """Hyper-parameter ensemble sweep (gpx_ensemble_sweep[_dev]): the member sweeps and their average stay
        on the device.  `Xc`: host array (M, d) or a DeviceGrid.  Returns dict(top_val, top_idx, acq, mu, s2)."""
        lead = engines[0]
        handles = (_P * len(engines))(*[e._h for e in engines])
        aid = ACQ[acq] if isinstance(acq, str) else int(acq)
        params = _f64([0.0 if param is None else param])
        tv = np.empty(k)
        ti = np.empty(k, dtype=np.int64)
        if isinstance(Xc, DeviceGrid):
            M = len(Xc)
            lead._check(lead._lib.gpx_ensemble_sweep_dev(handles, len(engines), aid, _ptr(params), 1, Xc.ptr, M,
                                                         k, _ptr(tv) if k else None, _ptr(ti) if k else None,
                                                         None, None, None))
            return dict(top_val=tv, top_idx=ti, acq=None, mu=None, s2=None)
        Xc = _f64(Xc).reshape(-1, lead.d)
        M = len(Xc)
        out = np.empty(M) if want_all else None
        mu = np.empty(M) if want_moments else None
        s2 = np.empty(M) if want_moments else None
        lead._check(lead._lib.gpx_ensemble_sweep(handles, len(engines), aid, _ptr(params), 1, _ptr(Xc), M, k,
                                                 _ptr(tv) if k else None, _ptr(ti) if k else None, _ptr(out),
                                                 _ptr(mu), _ptr(s2)))
        return dict(top_val=tv, top_idx=ti, acq=out, mu=mu, s2=s2)

    @staticmethod
    def ensemble_predict(engines, Xc):
        """Per-member moments and gradients at the rows of Xc in ONE call (gpx_ensemble_predict):
        mu, s2 (n, M); dmu, ds2 (n, M, d)."""
        lead = engines[0]
        handles = (_P * len(engines))(*[e._h for e in engines])
        Xc = _f64(Xc).reshape(-1, lead.d)
        n, M, d = len(engines), len(Xc), lead.d
        mu, s2 = np.empty((n, M)), np.empty((n, M))
        dmu, ds2 = np.empty((n, M, d)), np.empty((n, M, d))
        lead._check(lead._lib.gpx_ensemble_predict(handles, n, _ptr(Xc), M, _ptr(mu), _ptr(s2), _ptr(dmu), _ptr(ds2)))
        return mu, s2, dmu, ds2

    # -- Thompson --------------------------------------------------------------------------------
    def rff_gram(self, W, b):
        W = _f64(W)
        b = _f64(b)
        n = len(b)
        A = np.empty((n, n))
        v = np.empty(n)
        self._check(self._lib.gpx_rff_gram(self._h, _ptr(W), _ptr(b), n, _ptr(A), _ptr(v)))
        return A, v

    def rff_gram_batch(self, W, b):
        W = _f64(W)
        S, n, d = W.shape
        b = _f64(b).reshape(S, n)
        A = np.empty((S, n, n))
        v = np.empty((S, n))
        self._check(self._lib.gpx_rff_gram_batch(self._h, _ptr(W), _ptr(b), S, n, _ptr(A), _ptr(v)))
        return A, v

    def rff_posterior(self, W, b, z, sc):
        """Weights theta (S, n) of S posterior draws: feature Grams and the n x n solves on the device."""
        W = _f64(W)
        S, n, d = W.shape
        b = _f64(b).reshape(S, n)
        z = _f64(z).reshape(S, n)
        theta = np.empty((S, n))
        self._check(self._lib.gpx_rff_posterior(self._h, _ptr(W), _ptr(b), _ptr(z), S, n, float(sc), _ptr(theta)))
        return theta

    def rff_sweep(self, W, b, theta, bias, Xc, k=0, want_all=True):
        W = _f64(W)
        S, n, d = W.shape
        b = _f64(b).reshape(S, n)
        theta = _f64(theta).reshape(S, n)
        Xc = _f64(Xc).reshape(-1, d)
        M = len(Xc)
        tv = np.empty((S, k))
        ti = np.empty((S, k), dtype=np.int64)
        out = np.empty((S, M)) if want_all else None
        self._check(self._lib.gpx_rff_sweep(self._h, _ptr(W), _ptr(b), _ptr(theta), S, n, d, bias, _ptr(Xc), M,
                                            k, _ptr(tv) if k else None, _ptr(ti) if k else None, _ptr(out)))
        return dict(top_val=tv, top_idx=ti, vals=out)

    def rff_eval_grad(self, W, b, theta, bias, Xc):
        W = _f64(W)
        n, d = W.shape
        b = _f64(b)
        theta = _f64(theta)
        Xc = _f64(Xc).reshape(-1, d)
        M = len(Xc)
        f = np.empty(M)
        g = np.empty((M, d))
        self._check(self._lib.gpx_rff_grad(self._h, _ptr(W), _ptr(b), _ptr(theta), n, d, bias, _ptr(Xc), M,
                                           _ptr(f), _ptr(g)))
        return f, g

    def rff_sweep_dev(self, W, b, theta, bias, dXc_ptr, M, k):
        W = _f64(W)
        S, n, d = W.shape
        b = _f64(b).reshape(S, n)
        theta = _f64(theta).reshape(S, n)
        tv = np.empty((S, k))
        ti = np.empty((S, k), dtype=np.int64)
        self._check(self._lib.gpx_rff_sweep_dev(self._h, _ptr(W), _ptr(b), _ptr(theta), S, n, d, bias,
                                                _P(dXc_ptr), M, k, _ptr(tv), _ptr(ti), None))
        return tv, ti

    # -- measurement -----------------------------------------------------------------------------
    def timers(self, reset=False):
        out = np.zeros(len(TIMER_NAMES))
        self._lib.gpx_timers(self._h, _ptr(out), len(out), 1 if reset else 0)
        return dict(zip(TIMER_NAMES, out.tolist()))

    def chol_trace(self, nblocks, full_log=False):
        """Diagnostic (option chol_tg_trace = 1): the task-graph factorisation's own stamps of the last fit, microseconds
        from the first one: (diag (nblocks, 3) = wait / start / end per diagonal block, crit (nblocks, 8, 2) = the stamps of the
        workgroups that follow block row p -- slots 0, 1: S1 (waiting / right-hand sides loaded, .. / last rows stored), 2, 3: S2,
        4: U (waiting / earlier chunks in), 5: U (tile loaded / stored) -- see gpx_chol_trace in csrc/gpx_diag.h); None without a trace."""
        out = np.zeros(20 * nblocks + 8 * 1024 + 16 + (4 << 20 if full_log else 0), dtype=np.int64)
        n = self._lib.gpx_chol_trace(self._h, _ptr(out), out.size)
        if n < 20 * nblocks:
            return None
        self.last_chol_profile = out[20 * nblocks:20 * nblocks + 8192].reshape(1024, 8).copy()   # per workgroup (role index): see tg_trace.py
        self.last_chol_tasklog = None
        if full_log and n >= out.size:     # (option chol_tg_trace = 2) per workgroup 1024 records {task as 8 int16, start, end}
            self.last_chol_tasklog = out[20 * nblocks + 8192 + 16:].reshape(1024, 1024, 4).copy()
        out = out[:20 * nblocks]
        t0 = out[0]
        self.last_chol_trace_origin = int(t0)      # raw ticks of the first stamp (the task log's records are raw)
        diag = (out[:4 * nblocks].reshape(nblocks, 4)[:, :3] - t0) / 100.0
        crit = out[4 * nblocks:].reshape(nblocks, 8, 2).astype(np.float64)
        crit = np.where(crit > 0, (crit - t0) / 100.0, np.nan)
        return diag, crit

    def sync(self):
        self._check(self._lib.gpx_sync(self._h))
