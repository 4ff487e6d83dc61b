"""Column kernel with a table PER CONTROL (csrc/sdp_column_kernel.h, SDP_TRAIL_HAS_U):
models whose trailing next states depend on the control but not on the leading
state variable.  The nodes of a column share the partial interpolation over the
trailing axes control by control -- provided they share their control values --
so the workgroup rebuilds its LDS table for every control.  Bit-identical to
the direct kernel, the staged kernel and the numpy oracle."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _run(make, kernel, V=None, seed=8):
    _, s = make()
    s.kernel = kernel
    if V is None:
        V = np.random.default_rng(seed).standard_normal(s._state_grid_shape)
    J, pol = s.value_iteration(V, report_time=False)
    return J, pol, s.last_policy_index, s, V


def _price_maker(n_E=600, n_P=7, n_w=5, box_on_price=False, box_on_stock=False):
    """a stock whose use moves the price process (x1' depends on u), two controls"""
    sysd = SysDescription((2, 2, 1), name='price maker')
    sysd.dyn = lambda E, P, u, v, w: (E + u - 0.1 * v, 0.7 * P + w + 0.05 * u - 0.02 * v)
    sysd.cost = lambda E, P, u, v, w: P * u + 0.1 * u * u + 0.3 * (v - 0.2) ** 2 + 0.01 * E + w * v

    def box(E, P):
        hi = 1.0 + (0.1 * P if box_on_price else 0.0) + (0.01 * E if box_on_stock else 0.0)
        return ((-1.0, hi), (0.0, 0.5))
    sysd.control_box = box
    sysd.perturb_laws = [NormalLaw(0, 0.3)]
    s = DPSolver(sysd)
    s.discretize_state(0, 10, n_E, -2, 2, n_P)
    s.discretize_perturb(-0.6, 0.6, n_w)
    s.control_steps = (0.25, 0.25)
    return sysd, s


def _check_same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_control_coupled_benchmark_runs_the_per_control_table(gpu):
    make = lambda: models.synthetic3d_coupled(N=24)
    auto = _run(make, 'column')
    assert auto[3].backend_info['kernel'] == 'column' and auto[3].backend_info['table_per_control']
    _check_same(auto, _run(make, 'generic'))
    _check_same(auto, _run(make, 'staged'))


def test_against_the_numpy_oracle(gpu):
    from oracle import vi_numpy
    J, pol, idx, s, V = _run(lambda: models.synthetic3d_coupled(N=20), 'column',
                             V=models.synthetic3d_V0(models.synthetic3d_coupled(N=20)[1].state_grid))
    assert s.backend_info['table_per_control']
    nodes = np.arange(0, V.size, 11)
    Jo, polo, idxo, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(s), V, nodes=nodes)
    assert np.array_equal(J.ravel()[nodes], Jo) and np.array_equal(idx.ravel()[nodes], idxo)
    assert np.array_equal(pol.reshape(-1, 1)[nodes], polo)


def test_more_nodes_per_column_than_threads_two_controls(gpu):
    """600 nodes along axis 0 > 512 threads: the column is split over workgroups"""
    auto = _run(_price_maker, 'column')
    assert auto[3].backend_info['table_per_control'], auto[3].backend_info
    _check_same(auto, _run(_price_maker, 'generic'))


def test_box_may_depend_on_the_trailing_state_but_not_on_the_stock(gpu):
    make = lambda: _price_maker(n_E=40, box_on_price=True)
    auto = _run(make, 'column')
    assert auto[3].backend_info['table_per_control'] and auto[3].backend_info['box_per_node']
    _check_same(auto, _run(make, 'generic'))
    # a box that depends on the stock: the nodes of a column no longer share their controls
    make = lambda: _price_maker(n_E=40, box_on_stock=True)
    other = _run(make, 'auto')
    assert other[3].backend_info['kernel'] == 'generic'          # (a small grid: the direct kernel since round 5 ..
    staged = _run(make, 'staged')
    assert staged[3].backend_info['kernel'] == 'staged'          #  .. the staged tiles on request: same bits)
    _check_same(other, staged)


def test_eval_policy_and_relative_dp(gpu):
    make = lambda: _price_maker(n_E=48)
    Ja, pa, ia, one, V = _run(make, 'generic')
    Jb, pb, ib, two, _ = _run(make, 'column')
    assert two.backend_info['table_per_control']
    Ea, fa = one.eval_policy(pa, 4, rel_dp=True, report_time=False, J_ref_full=True)
    Eb, fb = two.eval_policy(pb, 4, rel_dp=True, report_time=False, J_ref_full=True)
    assert np.array_equal(Ea, Eb) and np.array_equal(fa, fb)
    ref = one._state_ref_ind
    Vd = V - V[ref]
    (Ka, ra), _ = one.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    (Kb, rb), _ = two.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    assert np.array_equal(Ka, Kb) and ra == rb


def test_deterministic_and_float32_per_control_tables(gpu):
    def make_det():
        sysd = SysDescription((2, 1, 0), name='deterministic price maker')
        sysd.dyn = lambda E, P, u: (E + u, 0.8 * P + 0.1 * u)
        sysd.cost = lambda E, P, u: P * u + 0.2 * u * u + 0.01 * E
        sysd.control_box = lambda E, P: ((-1., 1.),)
        s = DPSolver(sysd)
        s.discretize_state(0, 10, 70, -2, 2, 9)
        s.control_steps = (0.125,)
        return sysd, s
    auto = _run(make_det, 'column')
    assert auto[3].backend_info['table_per_control']
    _check_same(auto, _run(make_det, 'generic'))

    def make32():
        sysd, s = models.synthetic3d_coupled(N=20)
        s.dtype = np.dtype(np.float32)
        return sysd, s
    V = models.synthetic3d_V0(make32()[1].state_grid, np.float32)
    auto = _run(make32, 'column', V=V)
    assert auto[3].backend_info['table_per_control'] and auto[0].dtype == np.float32
    _check_same(auto, _run(make32, 'generic', V=V))
