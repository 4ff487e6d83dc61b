"""Column kernel with a ROW WINDOW (csrc/sdp_column_kernel.h, SDP_COL_ROWS): grids
whose W x N0 table exceeds the LDS of a CU tabulate, per segment of a column,
only the rows the segment's next states reach.  Bit-identical to the direct
kernel, also when the predicted window misses rows (controls outside it are
recomputed from global memory)."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _long_lead(N0=1024, n1=12, n2=10, dtype=np.float64):
    """the benchmark model on a grid with a 1024-point leading axis: 32 x 1024 x 8 B
    = 256 KiB of table, more than a CU's LDS"""
    sysd, s = models.synthetic3d(N=16)
    s.dtype = np.dtype(dtype)
    s.discretize_state(0, 1, N0, 0, 1, n1, 0, 1, n2)
    return sysd, s


def _storage(n_E=2048, n_P=9, n_w=11, p_max=0.35):
    """a stock with a state-dependent control box and a next stock level that depends
    on the perturbation (located per lattice cell)"""
    sysd = SysDescription((2, 1, 1), name='long storage')
    sysd.dyn = lambda E, P, u, w: (E + u - 0.05 * abs(u) + 0.02 * w, 0.8 * P + w)
    sysd.cost = lambda E, P, u, w: (P - u) ** 2 + 0.01 * E
    sysd.control_box = lambda E, P: ((np.max((-E, -p_max)), np.min((10 - E, p_max))),)
    sysd.perturb_laws = [NormalLaw(0, 0.5)]
    s = DPSolver(sysd)
    s.discretize_state(0, 10, n_E, -2, 2, n_P)
    s.discretize_perturb(-1.5, 1.5, n_w)
    s.control_steps = (0.05,)
    return sysd, s


def _pair(make, **attrs):
    out = []
    for kernel in ('generic', 'column'):       # ('auto' prefers the reduced-array sweep where it applies: see below)
        _, s = make()
        s.kernel = kernel
        for k, v in attrs.items():
            setattr(s, k, v)
        V = np.random.default_rng(4).standard_normal(s._state_grid_shape).astype(s.dtype)
        J, pol = s.value_iteration(V, report_time=False)
        out.append((J, pol, s.last_policy_index, s, V))
    return out


def _check(a, b):
    assert b[3].backend_info['kernel'] == 'column' and b[3].backend_info['row_window'], b[3].backend_info
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_table_larger_than_lds_runs_the_windowed_column_kernel(gpu):
    a, b = _pair(_long_lead)
    _check(a, b)
    win = b[3].backend_info['row_window']
    assert win['rows'] < 1024 and win['segment_nodes'] >= 64


def test_auto_takes_the_reduced_array_sweep_for_one_long_stock(gpu):
    """8-byte reals, a stock the perturbation does not reach: 'auto' runs csrc/sdp_lead_kernel.h with one
    controlled axis (5.9 ms against 12.7 ms at 1024 x 128 x 128, tools/window_vs_lead.py) -- same bits"""
    a, b = _pair(_long_lead)
    _, s = _long_lead()
    J, pol = s.value_iteration(a[4], report_time=False)
    assert s.backend_info['kernel'] == 'lead' and s.backend_info['controlled_axes'] == 1
    assert np.array_equal(J, a[0]) and np.array_equal(pol, a[1]) and np.array_equal(s.last_policy_index, a[2])
    assert np.array_equal(J, b[0])


def test_window_with_per_node_boxes_and_a_perturbed_stock(gpu):
    a, b = _pair(_storage)
    _check(a, b)


def test_window_float32(gpu):
    a, b = _pair(lambda: _long_lead(N0=2048, dtype=np.float32))
    _check(a, b)


def test_controls_outside_the_predicted_window_are_recomputed(gpu, monkeypatch):
    """a reach estimate that is far too small: segments as long as the window, most
    controls lead outside it -- same bits"""
    monkeypatch.setattr(DPSolver, '_lead_reach_rows', lambda self, model, bp, box_t=None: 1)
    a, b = _pair(_storage)
    _check(a, b)


def test_window_eval_policy_and_relative_dp(gpu):
    (Ja, pa, ia, one, V), (Jb, pb, ib, two, _) = _pair(_storage)
    Ea, fa = one.eval_policy(pa, 4, rel_dp=True, report_time=False, J_ref_full=True)
    Eb, fb = two.eval_policy(pb, 4, rel_dp=True, report_time=False, J_ref_full=True)
    assert two.backend_info['row_window']
    assert np.array_equal(Ea, Eb) and np.array_equal(fa, fb)
    ref = one._state_ref_ind
    Vd = V - V[ref]
    (Ka, ra), _ = one.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    (Kb, rb), _ = two.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    assert np.array_equal(Ka, Kb) and ra == rb


def test_window_with_a_ragged_leading_axis(gpu):
    """1000 rows: not a multiple of the segment or of the wavefront"""
    a, b = _pair(lambda: _long_lead(N0=1000, n1=7, n2=9))
    _check(a, b)
