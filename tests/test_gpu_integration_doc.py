"""INTEGRATION.md section B.1 executed: the documented reference-side ctypes binding
of `multilinear_interpolation` is cut out of the document VERBATIM, run in a
fresh interpreter (library found through LD_LIBRARY_PATH, as a maintainer of the
reference would install it) and fed the golden interpolation vectors of the
compiled reference -- so the documented binding cannot rot."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys
import numpy as np
g = np.load({golden!r})
n = 0
for c in range(int(g['n_cases'])):
    p = 'c{{:02d}}_'.format(c)
    out = multilinear_interpolation(g[p + 'smin'], g[p + 'smax'], g[p + 'orders'],
                                    np.ascontiguousarray(g[p + 'values']),
                                    np.ascontiguousarray(g[p + 's']))
    ref = g[p + 'out']
    assert out.dtype == ref.dtype and out.shape == ref.shape, c
    assert np.array_equal(out, ref, equal_nan=True), 'case {{}}'.format(c)
    n += 1
# the reference's error for d >= 5 (multilinear_cython.pyx:46-47), same text
try:
    multilinear_interpolation(np.zeros(5), np.ones(5), np.full(5, 2), np.zeros((1, 32)), np.zeros((5, 1)))
    raise SystemExit('no exception for d = 5')
except Exception as e:
    assert str(e) == "Can't interpolate in dimension strictly greater than 5", str(e)
print('binding of INTEGRATION.md B.1: {{}} golden cases bit-exact'.format(n))
'''


def documented_binding():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    section = text[text.index('### B.1'):text.index('### B.2')]
    blocks = re.findall(r'```python\n(.*?)```', section, flags=re.S)
    assert len(blocks) == 1, 'section B.1 should hold exactly one python block'
    return blocks[0]


def test_documented_binding_is_what_the_header_declares():
    """(no GPU needed) the block names entry points that include/sdp_hip.h declares"""
    block = documented_binding()
    header = open(os.path.join(ROOT, 'include', 'sdp_hip.h')).read()
    for name in set(re.findall(r'_lib\.(sdp_\w+)', block)):
        assert re.search(r'\b{}\s*\('.format(name), header), name
    assert "C.CDLL('libsdp_hip.so')" in block


@pytest.mark.gpu
def test_documented_binding_runs_the_golden_vectors(gpu, tmp_path):
    script = tmp_path / 'binding.py'
    script.write_text(documented_binding() + DRIVER.format(
        golden=os.path.join(ROOT, 'tests', 'golden', 'g1_interp.npz')))
    csrc = os.path.join(ROOT, 'stodynprog_amd', 'csrc')
    env = dict(os.environ, LD_LIBRARY_PATH=csrc + os.pathsep + os.environ.get('LD_LIBRARY_PATH', ''))
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=280, cwd=str(tmp_path))
    assert out.returncode == 0 and b'golden cases bit-exact' in out.stdout, out.stdout.decode()
