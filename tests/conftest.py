import os
import sys

import pytest

# the application's choice, made before the HIP runtime starts (pybo_amd itself no longer touches the environment): eight
# hardware queues keep the streams of several live handles apart
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
# The test-suite drives the DIAGNOSTICS build of the library (pybo_amd/csrc/libgpx_diag.so: the same objects as libgpx.so, but
# gpx_set_option also accepts the diagnostic knobs of csrc/gpx_diag.h -- chunk lists, grids, traces -- which the bit-identity and
# fuzz tests use as witnesses).  tests/test_abi.py loads the SHIPPING library as well and checks that it refuses them.
os.environ.setdefault('GPX_DIAGNOSTICS', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _have_gpu():
    try:
        from pybo_amd import _lib
        import ctypes as C
        lib = _lib.load()
        h = C.c_void_p()
        if lib.gpx_create(0, None, C.byref(h)) == 0:
            lib.gpx_destroy(h)
            return True
    except Exception:
        pass
    return False


@pytest.fixture(scope='session')
def gpu_available():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU (or without the built library) must FAIL loudly, not skip:
    # a silent skip would read as "parity green".  So nothing is auto-skipped here.
    pass
