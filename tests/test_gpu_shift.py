"""Certified filter on the SHIFTED lattice (csrc/sdp_colfilter_kernel.h, SDP_COL_SHIFT): a perturbation
that reaches the stock through a final sum, x0' = a(x, u) +- b(x_1.., w) -- the reference's inventory
example `x + u - w` (doc/example_inventory.py:31-33) next to an exogenous axis.  The first pass reads
G(s) = sum_w p_w T_w(s + shift_w), tabulated at the whole positions of a lattice moved by the
perturbation points, with a certified bound on what the interpolation between them leaves out; the
survivors take the reference's operations.  The claim under test is the one of test_gpu_filter.py:
J, policy and policy index are BIT-IDENTICAL to the kernel that evaluates every control the long way
and to the numpy oracle -- smooth and rough cost-to-go arrays, stocks that leave the grid, shifts wider
than the lattice held in LDS (the column then takes the long way), special values, any larger radius."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _sweep(make, filt, V, kernel='auto', sweeps=1):
    _, s = make()
    s.kernel = kernel
    s.certified_filter = filt
    J = np.asarray(V, dtype=float)
    with np.errstate(all='ignore'):
        for _ in range(sweeps):
            J, pol = s.value_iteration(J, report_time=False)
    return J, pol, s.last_policy_index, s


def _same(a, b):
    assert np.array_equal(a[0], b[0], equal_nan=True), 'J differs'
    assert np.array_equal(a[2], b[2]), 'policy index differs'
    assert np.array_equal(a[1], b[1], equal_nan=True), 'policy differs'


def _shop(n_x=96, n_d=12, n_w=7, w_gain=1.0, x_range=(-8., 24.), box_on_state=False, sign=-1, cost_w=False,
          order_step=0.25):
    """inventory next to a demand level: x' = (x + u) -+ (d + w_gain w)"""
    sysd = SysDescription((2, 1, 1), name='shop')
    if sign < 0:
        sysd.dyn = lambda x, d, u, w: (x + u - (d + w_gain * w), 2.0 + 0.7 * (d - 2.0) + 0.5 * w)
    else:
        sysd.dyn = lambda x, d, u, w: ((0.25 * d + w_gain * w) + (x + u), 2.0 + 0.7 * (d - 2.0) + 0.5 * w)
    if cost_w:
        sysd.cost = lambda x, d, u, w: np.where(x > 0, x * 0.5, -x * 3.0) + u * (1.0 + 0.1 * w)
    else:
        sysd.cost = lambda x, d, u, w: np.where(x > 0, x * 0.5, -x * 3.0) + u * 1.0
    if box_on_state:
        sysd.control_box = lambda x, d: ((0., 6. + 0.1 * x + d),)
    else:
        sysd.control_box = lambda x, d: ((0., 10.),)
    sysd.perturb_laws = [NormalLaw(0, 0.6)]
    s = DPSolver(sysd)
    s.discretize_state(x_range[0], x_range[1], n_x, 0., 4., n_d)
    s.discretize_perturb(-1.8, 1.8, n_w)
    s.control_steps = (order_step,)
    return sysd, s


def _smooth(s):
    x = np.asarray(s.state_grid[0])[:, None]
    d = np.asarray(s.state_grid[1])[None, :]
    return 0.05 * (x - 3.0) ** 2 + np.where(x > 0, 0.5 * x, -3.0 * x) + 0.3 * np.cos(d) * (1 + 0.01 * x)


def test_the_plan_and_the_generated_halves(gpu):
    _, s = _shop()
    plan = s._kernel_plan()
    assert plan['column'] and plan['filtered']
    src = plan['source']
    assert '#define SDP_COL_SHIFT 1' in src and '#define SDP_COL_SHIFT_TERMS 1' in src
    assert 'sdp_model_lead_a' in src and 'sdp_model_lead_b' in src
    s.dtype = np.dtype('float32')                 # 4-byte reals: every control the long way
    assert not s._kernel_plan()['filtered']


@pytest.mark.parametrize('sign', [-1, 1])
@pytest.mark.parametrize('box_on_state', [False, True])
def test_inventory_with_markov_demand_same_bits(gpu, sign, box_on_state):
    make = lambda: _shop(sign=sign, box_on_state=box_on_state)
    V = _smooth(make()[1])
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    assert on[3].backend_info['filter_form'] == 'shifted lattice' and not off[3].backend_info['certified_filter']
    assert on[3].backend_info['kernel'] == off[3].backend_info['kernel'] == 'column'
    _same(on, off)
    _same(on, _sweep(make, True, V, kernel='generic'))
    # a rough cost-to-go: large second differences, many survivors
    V = np.random.default_rng(5).standard_normal(V.shape)
    _same(_sweep(make, True, V), _sweep(make, False, V))


def test_against_the_numpy_oracle_over_a_chain_of_sweeps(gpu):
    from oracle import vi_numpy
    make = lambda: models.inventory_markov(n_x=64, n_d=10, n_w=7)
    _, s = make()
    V = np.zeros(s._state_grid_shape)
    on = _sweep(make, True, V, sweeps=4)
    assert on[3].backend_info['filter_form'] == 'shifted lattice'
    J = V
    spec = vi_numpy.Spec.from_solver(s)
    for _ in range(4):
        J, pol, idx, _ = vi_numpy.value_iteration(spec, J)
    assert np.array_equal(on[0], J) and np.array_equal(on[2], idx)


@pytest.mark.parametrize('case', ['leaves_the_grid', 'wide_shifts', 'shifts_beyond_the_lattice', 'coarse_axis'])
def test_edges_of_the_lattice(gpu, case):
    if case == 'leaves_the_grid':                # orders push the stock far above the axis, demand below it
        make = lambda: _shop(x_range=(-2., 6.))
    elif case == 'wide_shifts':                  # shifts spread over ~60 % of the axis
        make = lambda: _shop(w_gain=5.0)
    elif case == 'shifts_beyond_the_lattice':    # spread > the axis: more rows than LDS holds -> the long way
        make = lambda: _shop(w_gain=12.0)
    else:                                        # three rows: one inner row
        make = lambda: _shop(n_x=3)
    V = _smooth(make()[1])
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    assert on[3].backend_info['filter_form'] == 'shifted lattice'
    _same(on, off)
    V = np.random.default_rng(9).standard_normal(V.shape)
    _same(_sweep(make, True, V), _sweep(make, False, V))


@pytest.mark.parametrize('case', ['nan', 'inf', '-inf', 'huge', 'subnormal', 'mixed_scales', 'constant'])
def test_special_values(gpu, case):
    make = lambda: _shop()
    shape = make()[1]._state_grid_shape
    V = np.random.default_rng(5).standard_normal(shape)
    if case == 'nan':
        V[10:14, 2:5] = np.nan
    elif case == 'inf':
        V[:8, :] = np.inf                        # a forbidden region (deep backlog)
    elif case == '-inf':
        V[::7, 3] = -np.inf
    elif case == 'huge':
        V *= 1e302
    elif case == 'subnormal':
        V *= 1e-310
    elif case == 'mixed_scales':
        V[30:50] *= 1e12
    else:
        V[:] = 2.5                               # exact ties wherever the cost does not decide
    _same(_sweep(make, True, V), _sweep(make, False, V))


def test_split_columns_and_several_lanes_per_node(gpu):
    """few long columns: a column is shared by two workgroups (each builds the lattice, takes half of the nodes);
    a long control lattice: the lanes of a node share it and merge their bounds (b_max among them)"""
    make = lambda: models.inventory_markov()                      # 128 x 32 nodes: two units per column
    V = np.random.default_rng(23).standard_normal(make()[1]._state_grid_shape)
    on, off = _sweep(make, True, V, sweeps=2), _sweep(make, False, V, sweeps=2)
    assert on[3].backend_info['filter_form'] == 'shifted lattice'
    _same(on, off)
    make = lambda: _shop(order_step=0.02)                         # 501 controls per node
    for V in (_smooth(make()[1]), np.random.default_rng(24).standard_normal(make()[1]._state_grid_shape)):
        on, off = _sweep(make, True, V), _sweep(make, False, V)
        assert '#define SDP_COL_THREADS 384' in on[3]._kernel_plan()['source'] or \
            '#define SDP_COL_THREADS 512' in on[3]._kernel_plan()['source']
        _same(on, off)


def test_a_chain_of_sums(gpu):
    """x' = x + u - 0.5 w - 0.1 y: two terms without the stock or the control after a = x + u, one of them
    without the perturbation; and a cost that sees the perturbation"""
    def make():
        sysd = SysDescription((2, 1, 1), name='reservoir')
        sysd.dyn = lambda x, y, u, w: (x + u - 0.5 * w - 0.1 * y, 0.8 * y + w)
        sysd.cost = lambda x, y, u, w: (x - 0.3) * (x - 0.3) + 0.1 * u * u + 0.05 * w * u
        sysd.control_box = lambda x, y: ((-1., 1.),)
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(-1, 1, 65, -1, 1, 17)
        s.discretize_perturb(-0.5, 0.5, 7)
        s.control_steps = (0.05,)
        return sysd, s
    src = make()[1]._kernel_plan()['source']
    assert '#define SDP_COL_SHIFT_TERMS 2' in src and '#define SDP_COST_HAS_W 1' in src
    for V in (np.random.default_rng(21).standard_normal((65, 17)),
              np.add.outer(np.linspace(-1, 1, 65) ** 2, np.cos(np.linspace(-1, 1, 17)))):
        on, off = _sweep(make, True, V, sweeps=2), _sweep(make, False, V, sweeps=2)
        assert on[3].backend_info['filter_form'] == 'shifted lattice'
        _same(on, off)
        _same(on, _sweep(make, True, V, kernel='generic', sweeps=2))


NESTINGS = {
    'x+(w-u)': lambda x, y, u, w: (x + ((0.5 * w + 0.1 * y) - u), 0.8 * y + w),
    '(x-w)+u': lambda x, y, u, w: ((x - (0.5 * w + 0.1 * y)) + u, 0.8 * y + w),
    '(x-0.1y)-(u-w)': lambda x, y, u, w: ((x - 0.1 * y) - (u - 0.5 * w), 0.8 * y + w),
    'w-(u-x)': lambda x, y, u, w: (0.5 * w - (u - x), 0.8 * y + w),
    'four leaves': lambda x, y, u, w: ((x + 0.5 * w) + (0.3 * u - (0.1 * y + 0.2 * u)), 0.8 * y + w),
}


def _nested(nesting, x_range=(-1., 1.), dtype=np.float64):
    def make():
        sysd = SysDescription((2, 1, 1), name='reservoir, ' + nesting)
        sysd.dyn = NESTINGS[nesting]
        sysd.cost = lambda x, y, u, w: (x - 0.3) * (x - 0.3) + 0.1 * u * u
        sysd.control_box = lambda x, y: ((-1., 1.),)
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(x_range[0], x_range[1], 65, -1, 1, 17)
        s.discretize_perturb(-0.5, 0.5, 7)
        s.control_steps = (0.05,)
        s.dtype = np.dtype(dtype)
        return sysd, s
    return make


@pytest.mark.parametrize('nesting', sorted(NESTINGS))
def test_sums_in_another_nesting_are_regrouped_and_keep_the_bits(gpu, nesting):
    """x + (w - u) and its relatives (round 5; VERDICT r04 "shape cliffs"): the chain's leaves regrouped into a sum of
    the w-free ones and a sum of the others (TracedModel.lead_split, SDP_COL_SHIFT_CHAIN), the first pass on the
    shifted lattice with the sum of the leaves' magnitudes in its bound -- same bits as every control the long way, as
    the generic kernel and as the numpy oracle; smooth and rough cost-to-go, a grid far from the origin (leaves of
    magnitude 1e3 whose sum is the position)."""
    from oracle import vi_numpy
    make = _nested(nesting)
    plan = make()[1]._kernel_plan()
    assert '#define SDP_COL_SHIFT 1' in plan['source'] and plan['filtered']
    if nesting != 'w-(u-x)':                                             # (one w-free leaf there: an exact negation, nothing regrouped)
        assert '#define SDP_COL_SHIFT_CHAIN 0' not in plan['source'] and 'sdp_model_lead_aabs' in plan['source']
    for V in (np.random.default_rng(23).standard_normal((65, 17)),
              np.add.outer(np.linspace(-1, 1, 65) ** 2, np.cos(np.linspace(-1, 1, 17)))):
        on, off = _sweep(make, True, V, sweeps=2), _sweep(make, False, V, sweeps=2)
        assert on[3].backend_info['filter_form'] == 'shifted lattice'
        _same(on, off)
        _same(on, _sweep(make, True, V, kernel='generic', sweeps=2))
    s = make()[1]
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(s), V)
    J, u = s.value_iteration(V, report_time=False)
    assert np.array_equal(J, Jo) and np.array_equal(s.last_policy_index, io)
    far = _nested(nesting, x_range=(1000., 1002.))
    Vf = np.random.default_rng(24).standard_normal((65, 17))
    _same(_sweep(far, True, Vf), _sweep(far, False, Vf))


def test_a_cost_that_sees_the_perturbation_too(gpu):
    make = lambda: _shop(cost_w=True)
    V = _smooth(make()[1])
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    assert on[3].backend_info['filter_form'] == 'shifted lattice'
    assert 'SDP_COST_HAS_W 1' in on[3]._kernel_plan()['source']
    _same(on, off)


def test_weights_that_do_not_sum_to_one_and_tiny_weights(gpu):
    for factor in (3.7, 1e-30):
        def make():
            sysd, s = _shop()
            s.perturb_proba = [np.asarray(s.perturb_proba[0]) * factor]
            return sysd, s
        V = _smooth(make()[1]) * (1e300 if factor < 1 else 1.0)
        _same(_sweep(make, True, V), _sweep(make, False, V))


@pytest.mark.parametrize('scale', ['1e3', '1e9', '1e18'])
def test_any_larger_radius_gives_the_same_bits(gpu, debug_defines, scale):
    make = lambda: _shop()
    V = _smooth(make()[1])
    ref = _sweep(make, False, V)
    debug_defines.set(SDP_COL_FILTER_SCALE=scale)
    on = _sweep(make, True, V)
    assert 'SDP_COL_FILTER_SCALE' in on[3]._kernel_plan()['source']
    _same(on, ref)


def test_a_radius_far_too_small_is_noticed(gpu, debug_defines):
    """the interpolation bound B' is what decides here: without it (radius x 1e-6) the first pass
    trusts the chord where the kinks of the cost-to-go matter, and picks other controls"""
    make = lambda: _shop(order_step=0.05)
    V = np.random.default_rng(13).standard_normal(make()[1]._state_grid_shape)
    ref = _sweep(make, False, V)
    debug_defines.set(SDP_COL_FILTER_SCALE='1e-6')
    on = _sweep(make, True, V)
    assert (on[2] != ref[2]).sum() > 0


def test_the_benchmark_model_with_noise_in_the_stock(gpu):
    make = lambda: models.synthetic3d(N=24, n_w=8, stock_noise=0.07)
    V = models.synthetic3d_V0(make()[1].state_grid)
    on, off = _sweep(make, True, V, sweeps=3), _sweep(make, False, V, sweeps=3)
    assert on[3].backend_info['filter_form'] == 'shifted lattice'
    _same(on, off)


# ---------------------------------------------------------------------------
# Two-sided evidence for the ROUNDING part of the radius: with a cost-to-go that is linear in the stock the
# function G of a column is linear, the interpolation bound B' vanishes (up to the rounding of the second
# differences) and the objective is flat in the control -- the reference's argmin hangs on the last bits of
# its W x 6 roundings and of the positions, which the factor H of the bound has to cover on its own.
# ---------------------------------------------------------------------------
def _flat_shop(tilt, box_on_state=False):
    s_, b_ = 1.37, 0.0731
    sysd = SysDescription((2, 1, 1), name='flat objective, noise in the stock')
    sysd.dyn = lambda x, y, u, w: (x + b_ * u - 0.11 * w, 0.8 * y + w)
    sysd.cost = lambda x, y, u, w: (-s_ * b_) * u + tilt * (u * u)
    if box_on_state:
        sysd.control_box = lambda x, y: ((-1.0, 1.0 + 0.01 * x),)
    else:
        sysd.control_box = lambda x, y: ((-1.0, 1.0),)
    sysd.perturb_laws = [NormalLaw(0, 0.2)]
    s = DPSolver(sysd)
    s.discretize_state(0, 3, 192, -1, 1, 6)
    s.discretize_perturb(-0.5, 0.5, 7)
    s.control_steps = (2.0 / 47,)
    V = s_ * np.asarray(s.state_grid[0])[:, None] + np.cos(3 * np.asarray(s.state_grid[1]))[None, :]
    return sysd, s, V


@pytest.mark.parametrize('box_on_state', [False, True])
@pytest.mark.parametrize('scale', [None, '0.5'])
def test_near_ties_keep_the_bits_at_the_proven_radius_and_at_half_of_it(gpu, debug_defines, box_on_state, scale):
    for tilt in (0.0, 1e-16, 1e-15, 4e-15, 3e-14, 1e-12):
        make = lambda: _flat_shop(tilt, box_on_state)[:2]
        V = _flat_shop(tilt)[2]
        off = _sweep(make, False, V)
        if scale:
            debug_defines.set(SDP_COL_FILTER_SCALE=scale)
        on = _sweep(make, True, V)
        if scale:
            debug_defines.unset('SDP_COL_FILTER_SCALE')
        assert on[3].backend_info['filter_form'] == 'shifted lattice'
        _same(on, off)
        if tilt == 0.0:                          # (the case is what it claims to be: no clear winner)
            assert len(np.unique(off[2])) > 3


# ---------------------------------------------------------------------------
# Round 4: the shifted lattice in the resident-chunk kernel (csrc/sdp_colres_kernel.h): the table holds half of the
# perturbation points at a time, the lattice is accumulated in two pieces, the second pass locates its cell per
# perturbation point and carries up to TWO survivors through the rebuild of the tail.  Forced here on small
# problems (the planner picks it where the whole table leaves fewer than three workgroups per CU).
# ---------------------------------------------------------------------------
def _chunked(debug_defines, make, V, k, sweeps=1):
    debug_defines.set(SDP_COL_WRES=str(k))
    try:
        out = _sweep(make, True, V, sweeps=sweeps)
        src = out[3]._kernel_plan()['source']
        assert '#define SDP_COL_WRES {}'.format(k) in src and '#define SDP_COL_SHIFT 1' in src
    finally:
        debug_defines.unset('SDP_COL_WRES')
    return out


@pytest.mark.parametrize('case', ['smooth', 'rough', 'box_on_state', 'plus', 'leaves_the_grid', 'wide_shifts',
                                  'shifts_beyond_the_lattice', 'nan', 'inf', 'huge', 'constant', 'benchmark'])
def test_resident_chunks_on_the_shifted_lattice_give_the_same_bits(gpu, debug_defines, case):
    sweeps, k = 1, 4                                     # 7 points: 4 resident, 3 built twice
    if case == 'benchmark':
        make, k, sweeps = (lambda: models.synthetic3d(N=24, stock_noise=0.07)), 16, 2
        V = models.synthetic3d_V0(make()[1].state_grid)
    else:
        kw = {}
        if case == 'box_on_state':
            kw = dict(box_on_state=True)
        elif case == 'plus':
            kw = dict(sign=1)
        elif case == 'leaves_the_grid':
            kw = dict(x_range=(0., 8.))
        elif case == 'wide_shifts':
            kw = dict(w_gain=8.0)
        elif case == 'shifts_beyond_the_lattice':
            kw = dict(w_gain=40.0)                         # more rows than LDS holds: the column takes the long way
        make = lambda: _shop(**kw)
        s = make()[1]
        V = _smooth(s)
        rng = np.random.default_rng(3)
        if case == 'rough':
            V = rng.standard_normal(s._state_grid_shape)
        elif case == 'nan':
            V[10:14, 2:5] = np.nan
        elif case == 'inf':
            V[40:, :] = np.inf
        elif case == 'huge':
            V = V * 1e302
        elif case == 'constant':
            V = np.full(s._state_grid_shape, 2.5)
    plain, off = _sweep(make, True, V, sweeps=sweeps), _sweep(make, False, V, sweeps=sweeps)
    res = _chunked(debug_defines, make, V, k, sweeps)
    assert res[3].backend_info['filter_form'] == 'shifted lattice'
    _same(res, plain)
    _same(res, off)


@pytest.mark.parametrize('scale', [None, '0.5', '1e9'])
def test_resident_chunks_carry_two_survivors(gpu, debug_defines, scale):
    """an objective flat in the control (near-ties: nodes with exactly two survivors, and with many) at the proven
    radius, at half of it and at a radius that leaves every control standing"""
    for tilt in (0.0, 1e-15, 1e-13):
        make = lambda: _flat_shop(tilt)[:2]
        V = _flat_shop(tilt)[2]
        off = _sweep(make, False, V)
        if scale:
            debug_defines.set(SDP_COL_FILTER_SCALE=scale)
        try:
            res = _chunked(debug_defines, make, V, 4)
        finally:
            if scale:
                debug_defines.unset('SDP_COL_FILTER_SCALE')
        _same(res, off)


@pytest.mark.parametrize('controls', [41, 251])
def test_the_branch_and_bound_on_the_shifted_lattice_is_what_runs_and_keeps_the_bits(gpu, debug_defines, controls):
    """round 6: with final sums and the additive shape the resident-chunk kernel's first pass on the shifted lattice is the
    short one, as a branch and bound over blocks of controls (6 blocks of 8, 32 blocks of 8) -- the same bits as that pass
    without the pruning, as the lean pass of round 3, and as every control the long way; smooth and rough cost-to-go"""
    make = lambda: _shop(order_step=10.0 / (controls - 1))
    s0 = make()[1]

    def run(V, **knobs):
        debug_defines.set(SDP_COL_WRES='4', **knobs)
        try:
            out = _sweep(make, True, V, sweeps=2)
            return out, out[3]._kernel_plan()['source']
        finally:
            debug_defines.unset('SDP_COL_WRES', *knobs)
    for V in (_smooth(s0), np.random.default_rng(11).standard_normal(s0._state_grid_shape)):
        off = _sweep(make, False, V, sweeps=2)
        res, src = run(V)
        assert '#define SDP_COL_SHIFT 1' in src and '#define SDP_COL_WRES 4' in src
        assert '#define SDP_COL_LEAN2 1' in src and '#define SDP_COL_BNB 1' in src and '#define SDP_COL_UNROLL_W 1' in src
        assert res[3].backend_info['max_controls'] == controls
        _same(res, off)
        res, src = run(V, SDP_COL_BNB='0')
        assert '#define SDP_COL_LEAN2 1' in src and 'SDP_COL_BNB' not in src
        _same(res, off)
        res, src = run(V, SDP_COL_LEAN2='0')
        assert 'SDP_COL_LEAN2' not in src
        _same(res, off)
