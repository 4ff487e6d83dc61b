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


def test_the_product_library_has_no_collective_override(tmp_path):
    """SDP_RCCL_LIBRARY (the stand-in for the collective library used by the multi-rank tests) is
    compiled into the TEST build only (-DSDP_TEST_HOOKS): the product library neither reports test
    hooks nor contains the variable's name; the test build, made from the same source, does."""
    lib = nat.lib()
    assert lib.sdp_test_hooks() == 0
    assert b'SDP_RCCL_LIBRARY' not in open(nat.LIB_PATH, 'rb').read()
    if not os.path.exists(nat.HIPCC):
        pytest.skip('hipcc not available to build the test variant')
    hooks = nat.build_library(test_hooks_to=str(tmp_path / 'libsdp_hip_testhooks.so'))
    assert b'SDP_RCCL_LIBRARY' in open(hooks, 'rb').read()
    h = C.CDLL(hooks)
    h.sdp_test_hooks.restype = C.c_int
    assert h.sdp_test_hooks() == 1


def test_every_generated_unit_declares_what_it_was_built_for():
    """`sdp_meta` (csrc/sdp_kernel_args.h): column units, staged units and node-order units all
    define it, with the table sizes of the plan they were generated from"""
    from stodynprog_amd import models
    for make, kernel in ((lambda: models.synthetic3d(N=20), 'auto'), (lambda: models.synthetic3d(N=20), 'staged'),
                         (lambda: models.inventory(), 'auto'), (lambda: models.storage_ar1(), 'generic')):
        _, s = make()
        s.kernel = kernel
        src = s._kernel_plan()['source']
        headers = ''.join(open(os.path.join(nat.CSRC, h)).read() for h in
                          ('sdp_sweep_kernel.h', 'sdp_column_kernel.h', 'sdp_staged_kernel.h'))
        assert 'sdp_meta[SDP_META_WORDS]' in headers
        assert '#include "sdp_' in src
