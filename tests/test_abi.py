"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/gpx.h declares, and fails LOUDLY when there is no GPU (no silent fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'gpx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gpx_[a-z_0-9]+)\s*\(', src)))


def _declared_diag():
    src = open(os.path.join(ROOT, 'pybo_amd', 'csrc', 'gpx_diag.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gpx_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from pybo_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'libgpx.so does not export %s' % n
    # and the binding table covers the header exactly
    assert sorted(_lib.SYMBOLS) == names
    assert sorted(_lib.DIAG_SYMBOLS) == _declared_diag()


def test_shipping_and_diagnostics_builds_export_the_same_abi():
    """build.sh links two libraries from the same objects (api.hip compiled without / with -DGPX_DIAGNOSTICS): both export every
    symbol of include/gpx.h and csrc/gpx_diag.h; gpx_diagnostics() tells them apart; the header stays within its size budget."""
    from pybo_amd import _lib
    libdir = os.path.join(ROOT, 'pybo_amd', 'csrc')
    ship, diag = C.CDLL(os.path.join(libdir, 'libgpx.so')), C.CDLL(os.path.join(libdir, 'libgpx_diag.so'))
    for n in list(_lib.SYMBOLS) + list(_lib.DIAG_SYMBOLS):
        assert hasattr(ship, n) and hasattr(diag, n), n
    assert ship.gpx_diagnostics() == 0 and diag.gpx_diagnostics() == 1
    assert ship.gpx_version() == diag.gpx_version() >= 600
    assert os.path.basename(_lib.LIB_PATH) == 'libgpx_diag.so'          # what this test session drives (conftest.py)
    lines = open(os.path.join(ROOT, 'include', 'gpx.h')).read().splitlines()
    assert len(lines) <= 300 and max(len(ln) for ln in lines) <= 160
    # no diagnostic option is documented as part of the shipping ABI's option list
    public = re.sub(r'Diagnostic knobs.*?GPX_EARG\.', '', open(os.path.join(ROOT, 'include', 'gpx.h')).read(), flags=re.S)
    for name in ('x_skip', 'x_bg', 'chol_tg_chunks', 'chol_tg_nap', 'chol_tg_grid', 'chol_tg_isolate', 'chol_tg_trace', 'grad_rb_cs'):
        assert '"%s"' % name not in public, name


@pytest.mark.gpu
def test_shipping_library_refuses_the_diagnostic_options():
    """A handle of the SHIPPING library answers every diagnostic knob with GPX_EARG (and keeps working); the diagnostics build
    accepts them.  Both libraries live in this process side by side."""
    from pybo_amd import _lib
    ship = C.CDLL(os.path.join(ROOT, 'pybo_amd', 'csrc', 'libgpx.so'))
    ship.gpx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    ship.gpx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    ship.gpx_last_error.restype = C.c_char_p
    ship.gpx_last_error.argtypes = [C.c_void_p]
    ship.gpx_destroy.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert ship.gpx_create(0, None, C.byref(h)) == 0
    for name, val in (('x_skip', 1), ('x_bg', 4), ('x_bg_lds', 8), ('x_bg_iters', 10), ('chol_tg_chunks', 1248), ('chol_tg_nap', 8),
                      ('chol_tg_grid', 64), ('chol_tg_isolate', 0), ('chol_tg_trace', 1), ('grad_rb_cs', 128), ('x_rff', 1)):
        assert ship.gpx_set_option(h, name.encode(), val) == _lib.GPX_EARG, name
        assert b'GPX_DIAGNOSTICS' in ship.gpx_last_error(h)
    for name, val in (('chunk', 256), ('tile_order', 27), ('chol_tg', 0), ('chol_tg_min', 3), ('eager_inverse', 1)):
        assert ship.gpx_set_option(h, name.encode(), val) == 0, name
    assert ship.gpx_destroy(h) == 0
    e = _lib.Engine(0)                                   # the diagnostics build (this session's library)
    for name, val in (('chol_tg_chunks', 1248), ('chol_tg_nap', 8), ('chol_tg_trace', 1), ('x_skip', 0)):
        e.set_option(name, val)
    e.close()


def test_version_and_loud_failure_without_gpu(gpu_available):
    from pybo_amd import _lib
    lib = _lib.load()
    assert lib.gpx_version() >= 100
    if gpu_available:
        pytest.skip('a GPU is present; the no-GPU failure path is checked on CPU-only hosts')
    h = C.c_void_p()
    rc = lib.gpx_create(0, None, C.byref(h))
    assert rc == _lib.GPX_EHIP and not h.value
    assert b'no HIP device' in lib.gpx_last_error(None)
    with pytest.raises(_lib.GpxError):
        _lib.Engine(0)
    # the model object has no CPU path either
    from pybo_amd import models
    gp = models.make_gp(1e-3, 1.0, [0.3, 0.3], 0.0)
    with pytest.raises(_lib.GpxError):
        gp.add_data(np.random.rand(4, 2), np.random.rand(4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'pybo_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/gpx.h compiles as strict C99 and a C program linked against
    libgpx.so can call it (here only gpx_version / gpx_last_error: no device work on a CPU-only host)."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no C compiler on this host')
    libdir = os.path.join(ROOT, 'pybo_amd', 'csrc')
    src = tmp_path / 'abi_probe.c'
    src.write_text('#include <stdio.h>\n#include "gpx.h"\n'
                   'int main(void) {\n'
                   '    gpx_handle *h = 0; gpx_grid *g = 0;\n'
                   '    const char *msg = gpx_last_error(0);\n'
                   '    printf("%d %d\\n", gpx_version(), (int)(msg != 0));\n'
                   '    return (h == 0 && g == 0 && GPX_OK == 0 && GPX_GRID_SOBOL == 1) ? 0 : 1;\n'
                   '}\n')
    exe = tmp_path / 'abi_probe'
    subprocess.check_call([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic',
                           '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
                           '-L', libdir, '-lgpx', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) >= 100 and out[1] == '1'


def test_every_reference_citation_of_the_headers_points_at_an_existing_line():
    """include/gpx.h and the host modules cite the pybo call site each entry point replaces as `pybo/<file>.py:<line>[-<line>]`.
    Where the reference is present (the build container) every cited file exists and every cited line is inside it."""
    ref = '/root/reference'
    if not os.path.isdir(os.path.join(ref, 'pybo')):
        pytest.skip('the reference is not on this host (GPU box)')
    cited = set()
    files = [os.path.join(ROOT, 'include', 'gpx.h'), os.path.join(ROOT, 'DESIGN.md'), os.path.join(ROOT, 'INTEGRATION.md')]
    for dp, _, fs in os.walk(os.path.join(ROOT, 'pybo_amd')):
        files += [os.path.join(dp, f) for f in fs if f.endswith(('.py', '.hip', '.h'))]
    for path in files:
        txt = open(path, encoding='utf-8', errors='replace').read()
        for m in re.finditer(r'pybo/([a-z_/]+\.py):(\d+)(?:-(\d+))?((?:,\d+(?:-\d+)?)*)', txt):
            nums = [int(m.group(2))] + ([int(m.group(3))] if m.group(3) else [])
            nums += [int(v) for v in re.findall(r'\d+', m.group(4) or '')]
            cited.add((m.group(1), max(nums)))
    assert len(cited) >= 20
    for rel, last in sorted(cited):
        path = os.path.join(ref, 'pybo', rel)
        assert os.path.exists(path), rel
        n = len(open(path, encoding='utf-8', errors='replace').read().splitlines())
        assert last <= n, '%s has %d lines, cited line %d' % (rel, n, last)
