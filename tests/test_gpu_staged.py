"""LDS-staged tile kernel (csrc/sdp_staged_kernel.h) against the direct-gather
kernel and against the numpy oracle.  The staged box is only a prediction, so
the tests include dynamics that are NOT monotone over a chunk (cells outside the
box must fall back to global reads) and boxes larger than the LDS budget."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _both(make, V0=None, seed=3, **attrs):
    """one sweep with the direct and with the staged kernel; returns both solvers' results"""
    out = []
    for kernel in ('generic', 'staged'):
        _, s = make()
        s.kernel = kernel
        for k, v in attrs.items():
            setattr(s, k, v)
        V = np.random.default_rng(seed).standard_normal(s._state_grid_shape) if V0 is None else V0
        J, pol = s.value_iteration(V, report_time=False)
        assert s.backend_info['kernel'] == kernel, s.backend_info
        out.append((J, pol, s.last_policy_index, s))
    return out


def _same(a, b):
    assert np.array_equal(a[0], b[0], equal_nan=True)
    assert np.array_equal(a[1], b[1], equal_nan=True)
    assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize('name,kw', [
    ('inventory', {}),                                   # d = 1, discrete law
    ('nas_demo', {}),                                    # d = 2, per-node boxes
    ('storage_ar1', dict(n_E=21, n_P=31, steps=(0.05, 0.1))),   # two controls
    ('searev', dict(n_E=11, n_S=13, n_A=9, step=0.05)),
    ('synthetic3d', dict(N=20)),
    ('synthetic3d_coupled', dict(N=24)),
    ('synthetic3d_coupled', dict(N=21, cross=0.3)),      # fully coupled, ragged tiles
])
def test_staged_kernel_equals_the_direct_kernel(gpu, name, kw):
    a, b = _both(lambda: getattr(models, name)(**kw))
    _same(a, b)
    assert b[3].backend_info['staged']['threads'] == 512


def test_staged_kernel_matches_the_numpy_oracle_on_the_coupled_model(gpu):
    """the benchmark's control-coupled variant against oracle/vi_numpy.py (the
    reference's per-node loop with the same Python callables) on sampled nodes"""
    from oracle import vi_numpy
    _, s = models.synthetic3d_coupled(N=32)
    s.kernel = 'staged'
    V0 = models.synthetic3d_V0(s.state_grid)
    J, pol = s.value_iteration(V0, report_time=False)
    assert s.backend_info['kernel'] == 'staged'
    nodes = np.sort(np.random.default_rng(0).choice(V0.size, 400, replace=False))
    Jo, polo, idxo, mar = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(s), V0, nodes=nodes)
    assert np.array_equal(J.ravel()[nodes], Jo)
    assert np.array_equal(s.last_policy_index.ravel()[nodes], idxo)
    assert np.array_equal(pol.reshape(-1, 1)[nodes], polo)


def _wavy(N=24, n_w=7, amp=0.35):
    """dynamics that fold back inside a chunk of controls (x0' is NOT monotone in
    u) and jump with the perturbation: the corner prediction misses cells"""
    sysd = SysDescription((2, 1, 1), name='wavy')

    def dyn(x0, x1, u, w):
        return (x0 + amp * u * (1 - u) * 4 - 0.2, 0.5 * x1 + 0.4 * np.abs(w) * 8 + 0.1 * u * x0)
    sysd.dyn = dyn
    sysd.cost = lambda x0, x1, u, w: (x0 - 0.4) ** 2 + 0.3 * (u - x1) ** 2 + w * u
    sysd.control_box = lambda x0, x1: ((0., 1.),)
    sysd.perturb_laws = [NormalLaw(0, 0.05)]
    s = DPSolver(sysd)
    s.discretize_state(0, 1, N, 0, 1, N + 5)
    s.discretize_perturb(-0.12, 0.12, n_w)
    s.control_steps = (1. / 40,)
    return sysd, s


def test_cells_outside_the_predicted_box_fall_back_to_global_reads(gpu):
    a, b = _both(_wavy)
    _same(a, b)


def test_extrapolation_and_nan_through_the_staged_kernel(gpu):
    """next states far outside the grid (linear extrapolation, clamped cell) and
    NaN costs behave as in the direct kernel (first NaN wins the argmin)"""
    def make():
        sysd = SysDescription((2, 1, 1), name='wild')
        sysd.dyn = lambda x0, x1, u, w: (x0 + 5 * u + w, x1 * 3 - 1 + w * u)
        sysd.cost = lambda x0, x1, u, w: np.where(u > 0.9, np.nan, u * u + x0) + 0 * w
        sysd.control_box = lambda x0, x1: ((-1., 1.),)
        sysd.perturb_laws = [NormalLaw(0, 1.)]
        s = DPSolver(sysd)
        s.discretize_state(0, 1, 9, 0, 1, 14)
        s.discretize_perturb(-2, 2, 5)
        s.control_steps = (0.1,)
        return sysd, s
    a, b = _both(make)
    _same(a, b)


def test_four_state_variables_and_no_perturbation(gpu):
    def make():
        sysd = SysDescription((4, 1, 0), name='4-d deterministic')
        sysd.dyn = lambda a, b, c, e, u: (0.9 * a + 0.1 * u, 0.8 * b + 0.1 * a, 0.5 * c + 0.2 * b * u, 0.7 * e + 0.1)
        sysd.cost = lambda a, b, c, e, u: (a - 0.5) ** 2 + u * u + b * c + e
        sysd.control_box = lambda a, b, c, e: ((-1., 1.),)
        s = DPSolver(sysd)
        s.discretize_state(0, 1, 6, 0, 1, 5, 0, 1, 7, 0, 1, 9)
        s.control_steps = (0.25,)
        return sysd, s
    a, b = _both(make)
    _same(a, b)


def test_box_larger_than_the_lds_budget_is_cut_not_wrong(gpu, monkeypatch):
    """a chunk whose reach exceeds the budget: the box is cut and the rest of the
    cells read global memory -- forced here with a tiny budget"""
    from stodynprog_amd import codegen
    monkeypatch.setattr(codegen, 'STAGED_LDS_BYTES', 2048)
    a, b = _both(lambda: models.synthetic3d_coupled(N=24))
    _same(a, b)
    assert b[3].backend_info['staged']['cap'] == 256


def test_staged_relative_dp_and_eval_policy(gpu):
    _, one = models.synthetic3d_coupled(N=16)
    _, two = models.synthetic3d_coupled(N=16)
    one.kernel, two.kernel = 'generic', 'staged'
    V0 = models.synthetic3d_V0(one.state_grid)
    ref = one._state_ref_ind
    Vd = V0 - V0[ref]
    (Ja, ra), pa = one.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    (Jb, rb), pb = two.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    assert np.array_equal(Ja, Jb) and ra == rb and np.array_equal(pa, pb)
    Ea = one.eval_policy(pa, 3, report_time=False)
    Eb = two.eval_policy(pb, 3, report_time=False)
    assert np.array_equal(Ea, Eb)


def test_staged_float32_and_large_grid_through_the_host_path(gpu):
    """float32 staging (odd row stride rule) and a grid large enough for the host-array
    path to run the backup in phases (downloads under the next phase's kernel): node
    sub-ranges of a tile-walking kernel"""
    for dtype, N in ((np.float32, 24), (np.float64, 112)):
        res = []
        for kernel in ('generic', 'staged'):
            _, s = models.synthetic3d_coupled(N=N, cross=0.2)
            s.dtype = np.dtype(dtype)
            s.kernel = kernel
            V = models.synthetic3d_V0(s.state_grid, dtype)
            J, pol = s.value_iteration(V, report_time=False)
            assert s.backend_info['kernel'] == kernel and J.dtype == dtype
            res.append((J, pol, s.last_policy_index))
        _same(res[0], res[1])


def test_staged_minimal_axes(gpu):
    """axes of two points (one cell), fewer nodes than a tile"""
    def make():
        sysd = SysDescription((3, 1, 1), name='tiny')
        sysd.dyn = lambda a, b, c, u, w: (0.5 * a + u * b, b + w, 0.9 * c - 0.1 * u * a)
        sysd.cost = lambda a, b, c, u, w: a * a + u * u + c * w
        sysd.control_box = lambda a, b, c: ((0., 1.),)
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(0, 1, 2, -1, 1, 3, 0, 2, 2)
        s.discretize_perturb(-0.4, 0.4, 3)
        s.control_steps = (0.5,)
        return sysd, s
    a, b = _both(make)
    _same(a, b)
