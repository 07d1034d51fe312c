"""C-ABI surface (no GPU needed): the library builds/loads here, exports every function include/fpd_amd.h declares,
the ctypes mirrors have the library's struct sizes, and argument validation fails loudly without touching a device."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope='module')
def runtime():
    from fpd_amd import runtime as R
    if not os.path.exists(R.LIB_PATH):
        subprocess.check_call(['bash', os.path.join(os.path.dirname(R.LIB_PATH), 'build.sh')])
    R.lib()
    return R


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'fpd_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(fpd_[a-z0-9_]+)\s*\(', src)) - {'fpd_stream_t'})


def test_every_declared_symbol_is_exported_and_bound(runtime):
    names = declared_functions()
    assert len(names) >= 25
    lib = runtime.lib()
    for n in names:
        assert hasattr(lib, n), 'libfpd_amd.so does not export ' + n
        assert n in runtime.SYMBOLS, 'runtime.py has no ctypes prototype for ' + n
    assert sorted(runtime.SYMBOLS) == names


def test_struct_sizes_match(runtime):
    import ctypes
    for name, st in runtime._STRUCTS.items():
        assert runtime.lib().fpd_abi_sizeof(name.encode()) == ctypes.sizeof(st), name
    assert runtime.lib().fpd_abi_sizeof(b'nope') == -1
    assert runtime.lib().fpd_abi_version() == 1


def test_validation_errors_without_device(runtime):
    lib = runtime.lib()
    a = runtime.ConvT()
    assert lib.fpd_conv_forward(a, None) != 0 and b'null' in lib.fpd_last_error()
    a.x = a.w = a.y = 1
    a.N, a.H, a.W, a.C, a.K, a.R, a.S, a.stride, a.pad, a.P, a.Q = 1, 8, 8, 16, 16, 3, 3, 1, 1, 7, 8
    assert lib.fpd_conv_forward(a, None) != 0 and b'inconsistent' in lib.fpd_last_error()
    p = lib.fpd_plan_create()
    assert lib.fpd_plan_add(p, 99, runtime.ctypes_byref(a) if hasattr(runtime, 'ctypes_byref') else None, 0) < 0
    lib.fpd_plan_destroy(p)
    with pytest.raises(runtime.FpdError):
        runtime.check(-2, 'x')


def test_missing_library_is_an_error_not_a_fallback(runtime, monkeypatch):
    monkeypatch.setattr(runtime, '_lib', None)
    monkeypatch.setattr(runtime, 'LIB_PATH', '/nonexistent/libfpd_amd.so')
    with pytest.raises(runtime.FpdError):
        runtime.lib()
