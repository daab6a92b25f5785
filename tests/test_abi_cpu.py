"""CPU-side checks of the C ABI: the shared library loads, exports every symbol that
include/vwgpu.h declares, and the pure-host entry points behave.  No GPU needed."""
import ctypes as C
import os
import re

import pytest

from vectorwave_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "vwgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vw_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = N.load()
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), "libvwgpu.so does not export %s" % s
        assert s in N.SIGNATURES, "python binding lacks a prototype for %s" % s
    assert sorted(N.SIGNATURES) == syms


def test_max_levels_matches_reference_rule():
    lib = N.load()
    # modwt/MultiLevelMODWTTransformTest.java:254-291
    assert lib.vw_max_levels(10000, 2) == 9
    assert lib.vw_max_levels(8, 8) == 0
    assert lib.vw_max_levels(9, 8) == 1
    assert lib.vw_max_levels(480, 8) == 7
    assert lib.vw_max_levels(16384, 8) == 9
    from oracle import oracle as o
    for n in (1, 2, 3, 17, 64, 100, 777, 4096, 777777):
        for l in (2, 4, 8, 9, 20, 102):
            assert lib.vw_max_levels(n, l) == o.max_levels(n, l)


def test_null_context_and_workspace_query():
    lib = N.load()
    assert lib.vw_destroy(None) == N.VW_E_NULL
    assert lib.vw_last_error(None) == b"null context"
    assert lib.vw_swt_denoise_workspace_bytes(8192, 480, 4) >= 8192 * 5 * 480 * 8 + 8192 * 8
    assert lib.vw_swt_denoise_workspace_bytes(0, 480, 4) == 0


def test_no_gpu_fails_loudly():
    lib = N.load()
    if lib.vw_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.vw_init(0, C.byref(h)) == N.VW_E_NO_DEVICE
    with pytest.raises(N.NativeLibraryError):
        N.Context(0)


def test_one_hip_runtime_whatever_the_import_order():
    """libvwgpu.so loaded before torch must not leave the process with two HIP/HSA runtimes (the second
    one finds no device).  Fresh interpreter: load the library, then import torch, then list the mapped
    runtime libraries."""
    import subprocess
    import sys
    code = (
        "from vectorwave_amd import _native as N\n"
        "N.load()\n"
        "import torch\n"
        "libs = set()\n"
        "for line in open('/proc/self/maps'):\n"
        "    p = line.split()[-1]\n"
        "    if 'libamdhip64' in p or 'libhsa-runtime64' in p:\n"
        "        libs.add(p)\n"
        "names = sorted(libs)\n"
        "assert len([p for p in names if 'libamdhip64' in p]) == 1, names\n"
        "assert len([p for p in names if 'libhsa-runtime64' in p]) == 1, names\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
