"""The C-ABI library loads on a machine without a GPU and exports every entry
point declared in include/sdp_hip.h; argument errors are reported through the
status code + sdp_last_error() convention.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from stodynprog_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'sdp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sdp_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = nat.lib()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), 'libsdp_hip.so does not export ' + name
    # and the Python binding declares a prototype for each of them
    assert sorted(nat.EXPORTS) == names


def test_error_convention_without_gpu():
    lib = nat.lib()
    h = C.c_void_p()
    assert lib.sdp_problem_create(None, C.byref(h)) == -1
    assert b'NULL' in lib.sdp_last_error()
    d = nat.sdp_problem_desc()
    d.dtype, d.d, d.nu = 0, 7, 1
    assert lib.sdp_problem_create(C.byref(d), C.byref(h)) == -2          # SDP_EDIM
    assert b'dimension' in lib.sdp_last_error()
    one = np.zeros(8)
    orders = np.full(5, 2, dtype=np.int64)
    rc = lib.sdp_mlinterp_f64(5, nat.ptr(one), nat.ptr(one), nat.ptr(orders), nat.ptr(one), 1,
                              nat.ptr(one), 1, nat.ptr(one))
    assert rc == -2
    # same text as the reference raises (multilinear_cython.pyx:47)
    assert lib.sdp_last_error() == b"Can't interpolate in dimension strictly greater than 5"
    with pytest.raises(Exception) as e:
        nat.check(rc)
    assert "strictly greater than 5" in str(e.value)
    assert lib.sdp_problem_vi_sweep(None, 0.0, 0, 0, None) == -1
    assert lib.sdp_comm_create(3, 2, b'x' * 128, C.byref(h)) == -1


def test_descriptor_layout_matches_the_header(tmp_path):
    """sizeof / offsetof of every field of sdp_problem_desc as gcc lays the struct of
    include/sdp_hip.h out, against the ctypes mirror in _native.py"""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    fields = [name for name, _ in nat.sdp_problem_desc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sdp_hip.h"\nint main(void) {\n'
                   + '  printf("%zu\\n", sizeof(sdp_problem_desc));\n'
                   + ''.join('  printf("%zu\\n", offsetof(sdp_problem_desc, {}));\n'.format(f) for f in fields)
                   + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(src)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == C.sizeof(nat.sdp_problem_desc)
    assert nums[1:] == [getattr(nat.sdp_problem_desc, f).offset for f in fields]
    # 4*int32 + 4*int64 + 4 ptr + 2 ptr + 4*int32 + 3 ptr + 2*int64 + ptr + 4*int32 + 2*int32
    assert nums[0] == 16 + 32 + 32 + 16 + 16 + 24 + 16 + 8 + 16 + 8


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(nat, '_lib', None)
    monkeypatch.setattr(nat, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(nat.NativeError) as e:
        nat.lib()
    assert 'no CPU fallback' in str(e.value)
