import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# parity bars (BASELINE.json north_star): fp64 |dV|/|V| < 1e-10, policy indices
# exact.  The reference's expectation is np.inner (BLAS summation order), ours
# is sequential in w, so J agrees to ~1e-15 and the argmin can only differ
# where two controls tie to that level (SURVEY section 7 "hard parts").
RTOL_J = 1e-10
TIE_RTOL = 1e-12


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box)')
    # a fresh checkout has no libsdp_hip.so yet: build it once (hipcc cross-compiles
    # gfx950 without a GPU); the tests themselves never fall back to anything else
    from stodynprog_amd import _native as nat
    if not os.path.exists(nat.LIB_PATH) and os.path.exists(nat.HIPCC):
        nat.build_library()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def assert_sweep_parity(J, idx, J_ref, idx_ref, margin_ref, what='', rtol=RTOL_J,
                        tie_rtol=TIE_RTOL):
    """J within rtol of the reference everywhere; argmin index identical
    wherever the reference's best/second-best margin is above the tie level."""
    J, J_ref = np.asarray(J, dtype=float), np.asarray(J_ref, dtype=float)
    scale = max(1.0, float(np.abs(J_ref).max()))
    err = np.abs(J - J_ref).max() / scale
    assert err < rtol, '{}: max |dJ|/|J| = {:.3e}'.format(what, err)
    idx, idx_ref = np.asarray(idx).astype(np.int64), np.asarray(idx_ref).astype(np.int64)
    diff = idx != idx_ref
    if diff.any():
        tie = np.asarray(margin_ref) <= tie_rtol * np.maximum(1.0, np.abs(J_ref))
        bad = diff & ~tie
        assert not bad.any(), '{}: {} argmin mismatches outside near-ties (of {} nodes)'.format(
            what, int(bad.sum()), diff.size)
    return err, int(diff.sum())


def have_gpu():
    try:
        from stodynprog_amd import _native as nat
        return nat.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu():
    from stodynprog_amd import _native as nat
    nat.require_gpu()          # loud failure: GPU tests never fall back to anything
    nat.check(nat.lib().sdp_set_device(0))
    return nat
