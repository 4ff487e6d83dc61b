"""Several controlled state variables next to an exogenous process (csrc/sdp_lead_kernel.h): the
node-order sweep with the certified expectation-first filter on an array reduced over w.  The
reference's API admits any `dims` (stodynprog.py:57-81) and multi-control lattices (:655-660); the
claim under test is the one of test_gpu_filter.py: J, policy and policy index are BIT-IDENTICAL to
the direct kernel (every control, every perturbation point, 2^d vertex loads) and to the numpy
oracle -- smooth, random and special cost-to-go arrays, stocks that leave the grid, per-node
boxes, any larger radius; and a radius far too small is noticed."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _sweep(make, kernel, V, sweeps=1, dtype=np.float64):
    _, s = make()
    s.kernel = kernel
    s.dtype = np.dtype(dtype)
    J = np.asarray(V, dtype=dtype)
    with np.errstate(all='ignore'):
        for _ in range(sweeps):
            J, pol = s.value_iteration(J, report_time=False)
    return J, pol, s.last_policy_index, s


def _same(a, b):
    assert np.array_equal(a[0], b[0], equal_nan=True), 'J differs'
    assert np.array_equal(a[2], b[2]), 'policy index differs'
    assert np.array_equal(a[1], b[1], equal_nan=True), 'policy differs'


def _small(box_on_state=False, n=(14, 12, 9)):
    make = lambda: models.two_reservoirs(n_a=n[0], n_b=n[1], n_y=n[2], n_w=5, steps=(0.25, 0.25))
    if not box_on_state:
        return make

    def make2():
        sysd, s = make()
        sysd.control_box = lambda a, b, y: ((0., 0.5 + 0.25 * a), (0., 0.6 + 0.2 * b + 0.1 * y))
        s._cache.clear()
        return sysd, s
    return make2


def _smooth(s):
    a, b, y = [np.asarray(g) for g in s.state_grid]
    return ((a[:, None, None] - 1.0) ** 2 + 0.5 * (b[None, :, None] - 0.7) ** 2
            + 0.3 * np.cos(3 * y)[None, None, :] * (1 + 0.1 * a[:, None, None]))


def test_the_plan(gpu):
    _, s = models.two_reservoirs()
    assert s._traced().controlled_axes() == 2
    plan = s._kernel_plan()
    assert plan['lead_axes'] == 2 and plan['lanes'] == 1 and plan['filtered'] and not plan['column']
    src = plan['source']
    assert '#define SDP_LEAD_AXES 2' in src and 'sdp_model_leads' in src and 'sdp_model_trails' in src
    s.dtype = np.dtype('float32')                 # 4-byte reals: another kernel (round 5's form for them was 3 x slower than
    assert not s._kernel_plan()['lead_axes']      # every control the long way and went in round 6)
    s.kernel = 'lead'
    with pytest.raises(ValueError):
        s._kernel_plan()


@pytest.mark.parametrize('box_on_state', [False, True])
def test_two_reservoirs_same_bits_as_the_direct_kernel(gpu, box_on_state):
    make = _small(box_on_state)
    V = _smooth(make()[1])
    lead, gen = _sweep(make, 'auto', V), _sweep(make, 'generic', V)
    assert lead[3].backend_info['kernel'] == 'lead' and lead[3].backend_info['filter_form'] == 'reduced array'
    assert lead[3].backend_info['controlled_axes'] == 2 and gen[3].backend_info['kernel'] == 'generic'
    _same(lead, gen)
    V = np.random.default_rng(3).standard_normal(V.shape)
    _same(_sweep(make, 'auto', V), _sweep(make, 'generic', V))
    _same(_sweep(make, 'auto', V, sweeps=3), _sweep(make, 'generic', V, sweeps=3))


def test_against_the_numpy_oracle(gpu):
    from oracle import vi_numpy
    make = _small()
    _, s = make()
    V = _smooth(s)
    lead = _sweep(make, 'auto', V, sweeps=2)
    spec = vi_numpy.Spec.from_solver(s)
    J = V
    for _ in range(2):
        J, pol, idx, _ = vi_numpy.value_iteration(spec, J)
    assert np.array_equal(lead[0], J) and np.array_equal(lead[2], idx)


@pytest.mark.parametrize('case', ['nan', 'inf', 'huge', 'subnormal', 'constant'])
def test_special_values(gpu, case):
    make = _small()
    V = np.random.default_rng(5).standard_normal(make()[1]._state_grid_shape)
    if case == 'nan':
        V[3:5, 2:4, 1:3] = np.nan
    elif case == 'inf':
        V[:2] = np.inf
    elif case == 'huge':
        V *= 1e302
    elif case == 'subnormal':
        V *= 1e-310
    else:
        V[:] = 2.5
    _same(_sweep(make, 'auto', V), _sweep(make, 'generic', V))


def test_one_exogenous_axis_less_and_one_more(gpu):
    """d = 2 with both state variables controlled (no exogenous axis: the perturbation only reaches the
    cost-free part, i.e. nothing) is not this family; d = 4 with two exogenous axes is"""
    def make():
        sysd = SysDescription((4, 2, 1), name='two stocks, two exogenous')
        sysd.dyn = lambda a, b, y, z, u, v, w: (a + 0.5 * y - u, b + u - v, 0.7 * y + w, 0.5 * z - 0.3 * w + 0.1 * y)
        sysd.cost = lambda a, b, y, z, u, v, w: (v - 0.5 - 0.2 * z) * (v - 0.5 - 0.2 * z) + 0.1 * u * u + 0.2 * (a - 1.0) * (a - 1.0)
        sysd.control_box = lambda a, b, y, z: ((0., 1.), (0., 1.))
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(0, 2, 9, 0, 2, 8, -1, 1, 7, -1, 1, 6)
        s.discretize_perturb(-0.5, 0.5, 5)
        s.control_steps = (0.25, 0.25)
        return sysd, s
    V = np.random.default_rng(9).standard_normal(make()[1]._state_grid_shape)
    lead, gen = _sweep(make, 'auto', V), _sweep(make, 'generic', V)
    assert lead[3].backend_info['kernel'] == 'lead'
    _same(lead, gen)


def test_a_cost_that_sees_the_perturbation(gpu):
    """the first pass accumulates the cost's expectation with the reference's own g_w"""
    def make():
        sysd, s = _small()()
        sysd.cost = lambda a, b, y, u, v, w: ((v - 0.8 - w) * (v - 0.8 - w) + 0.05 * (u - v) * (u - v)
                                              + 0.3 * (a - 1.0) * (a - 1.0) * (1.0 + w) + 0.2 * b * u)
        s._cache.clear()
        return sysd, s
    src = make()[1]._kernel_plan()['source']
    assert '#define SDP_LEAD_COST_HAS_W 1' in src
    for V in (_smooth(make()[1]), np.random.default_rng(17).standard_normal(make()[1]._state_grid_shape)):
        lead, gen = _sweep(make, 'auto', V, sweeps=2), _sweep(make, 'generic', V, sweeps=2)
        assert lead[3].backend_info['kernel'] == 'lead'
        _same(lead, gen)
    V = np.random.default_rng(18).standard_normal(make()[1]._state_grid_shape)
    V[:2] = np.inf
    V[5, 3, 2] = np.nan
    _same(_sweep(make, 'auto', V), _sweep(make, 'generic', V))

    def tiny():                                   # weights of 1e-30 and a cost of 1e307: the raw magnitude decides
        sysd, s = make()
        sysd.cost = lambda a, b, y, u, v, w: 1e307 * ((v - 0.8 - w) * (v - 0.8 - w) + (a - 1.0) * (a - 1.0))
        s.perturb_proba = [np.asarray(s.perturb_proba[0]) * 1e-30]
        s._cache.clear()
        return sysd, s
    V = _smooth(tiny()[1]) * 1e306
    _same(_sweep(tiny, 'auto', V), _sweep(tiny, 'generic', V))


@pytest.mark.parametrize('scale', ['1e4', '1e12', '1e18'])
def test_any_larger_radius_gives_the_same_bits(gpu, debug_defines, scale):
    make = _small()
    V = _smooth(make()[1])
    ref = _sweep(make, 'generic', V)
    debug_defines.set(SDP_LEAD_FILTER_SCALE=scale)
    on = _sweep(make, 'auto', V)
    assert 'SDP_LEAD_FILTER_SCALE' in on[3]._kernel_plan()['source']
    _same(on, ref)


def test_near_ties_and_a_radius_far_too_small(gpu, debug_defines):
    """an objective that is flat in the controls: V linear in both stocks, the cost cancels the slope --
    the reference's argmin hangs on the last bits of its W x (3d + 3) roundings.  Same bits at the proven
    radius; with the radius cut by 1e6 the first pass picks its own minimum and differs somewhere."""
    def make():
        sysd = SysDescription((3, 2, 1), name='flat')
        sysd.dyn = lambda a, b, y, u, v, w: (a + 0.37 * u, b + 0.29 * v, 0.8 * y + w)
        sysd.cost = lambda a, b, y, u, v, w: (-1.3 * 0.37) * u + (-0.7 * 0.29) * v
        sysd.control_box = lambda a, b, y: ((-1., 1.), (-1., 1.))
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(0, 3, 24, 0, 3, 20, -1, 1, 6)
        s.discretize_perturb(-0.5, 0.5, 7)
        s.control_steps = (2.0 / 11, 2.0 / 9)
        return sysd, s
    g = make()[1].state_grid
    V = (1.3 * np.asarray(g[0])[:, None, None] + 0.7 * np.asarray(g[1])[None, :, None]
         + np.cos(3 * np.asarray(g[2]))[None, None, :])
    ref = _sweep(make, 'generic', V)
    assert len(np.unique(ref[2])) > 5
    _same(_sweep(make, 'auto', V), ref)
    debug_defines.set(SDP_LEAD_FILTER_SCALE='0.5')            # half the proven radius: still the same bits
    _same(_sweep(make, 'auto', V), ref)
    debug_defines.set(SDP_LEAD_FILTER_SCALE='1e-6')
    assert (_sweep(make, 'auto', V)[2] != ref[2]).sum() > 0


def test_policy_evaluation_and_iteration_run_on_the_same_problem(gpu):
    make = _small()
    _, s = make()
    V = _smooth(s)
    J, pol = s.value_iteration(V, report_time=False)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        E1 = s.eval_policy(pol, 3, False, V)
    _, g = make()
    g.kernel = 'generic'
    with contextlib.redirect_stdout(io.StringIO()):
        E2 = g.eval_policy(pol, 3, False, V)
    assert np.array_equal(E1, E2)


# ---------------------------------------------------------------------------
# Round 4: the stocks need not be listed first.  The reference takes the order of the state variables from
# dyn's signature (stodynprog.py:119-131) and its lerp nest follows that order (multilinear_cython.pyx:177-208);
# the filter works on a permuted view of the axes (stocks first), the second pass evaluates the reference's
# nest in the reference's own order.
# ---------------------------------------------------------------------------
def _stock_not_first(three=False, cost_w=False):
    if three:
        s = SysDescription((3, 1, 1), name='stock in the middle')
        s.dyn = lambda a, y, b, u, w: (0.8 * a + w, y + 0.7 * u, 0.5 * b - 0.3 * w + 0.1 * a)
        if cost_w:
            s.cost = lambda a, y, b, u, w: (a - 0.3 + 0.4 * w) * u + 0.2 * u * u + 0.05 * y + 0.1 * b * b
        else:
            s.cost = lambda a, y, b, u, w: (a - 0.3) * u + 0.2 * u * u + 0.05 * y + 0.1 * b * b
        s.control_box = lambda a, y, b: ((-1.0, 1.0 + 0.2 * a * a),)
    else:
        s = SysDescription((2, 1, 1), name='stock listed last')
        s.dyn = lambda a, y, u, w: (0.8 * a + w, y + 0.7 * u)
        s.cost = lambda a, y, u, w: (a - 0.3) * u + 0.2 * u * u + 0.05 * y
        s.control_box = lambda a, y: ((-1.0, 1.0),)
    s.perturb_laws = [NormalLaw(0, 0.2)]
    sol = DPSolver(s)
    if three:
        sol.discretize_state(-1, 1, 9, 0, 3, 40, -1, 1, 7)
    else:
        sol.discretize_state(-1, 1, 9, 0, 3, 40)
    sol.discretize_perturb(-0.5, 0.5, 7)
    sol.control_steps = (0.0625,)
    return s, sol


@pytest.mark.parametrize('three,cost_w', [(False, False), (True, False), (True, True)])
def test_a_stock_that_is_not_listed_first(gpu, three, cost_w, recwarn):
    from oracle import vi_numpy
    make = lambda: _stock_not_first(three, cost_w)
    _, a = make()
    shape = a._state_grid_shape
    V = np.random.default_rng(12).standard_normal(shape)
    Ja, pa = a.value_iterations(V, 3, report_time=False)
    assert a.backend_info['kernel'] == 'lead' and a.backend_info['certified_filter']
    assert a.backend_info['controlled_order'] == ([1, 0, 2] if three else [1, 0])
    assert not [w for w in recwarn.list if 'listing it' in str(w.message)]          # no "list it FIRST" advice any more
    src = a._kernel_plan()['source']
    assert '#define SDP_LEAD_PERM {1, 0, 2, 3}' in src and '#define SDP_LEAD_AXES 1' in src
    _, b = make()
    b.kernel = 'generic'
    Jb, pb = b.value_iterations(V, 3, report_time=False)
    assert np.array_equal(Ja, Jb) and np.array_equal(pa, pb)
    assert np.array_equal(a.last_policy_index, b.last_policy_index)
    # and the numpy oracle (the reference's nest order, its own callables) on one sweep
    J1, _ = a.value_iteration(V, report_time=False)
    Jo, _, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(a), V)
    assert np.array_equal(J1, Jo) and np.array_equal(a.last_policy_index, io)
    # special values take the long way on V itself
    V2 = V.copy()
    V2[(2,) * V.ndim] = np.nan
    V2[(0,) + (5,) * (V.ndim - 1)] = np.inf
    with np.errstate(all='ignore'):
        Jn, _ = a.value_iteration(V2, report_time=False)
        Jg, _ = b.value_iteration(V2, report_time=False)
    assert np.array_equal(Jn, Jg, equal_nan=True) and np.array_equal(a.last_policy_index, b.last_policy_index)
