"""Certified expectation-first filter of the column kernel (csrc/sdp_colfilter_kernel.h,
SdpColFilter).  When the perturbation reaches neither x0' nor the cost, the first pass
decides all but the near-minimal controls of a node on a table reduced over w, with a
rigorous error radius, and only the survivors are evaluated with the reference's
operations.  The claim under test: J, policy and policy index are BIT-IDENTICAL to the
kernel that evaluates every control the long way -- for ordinary inputs (one survivor
per node), for exact ties, for NaN / infinite / huge / subnormal values (the node then
takes the long way), and for any error radius >= the proven one (SDP_COL_FILTER_SCALE
blows it up until every control survives)."""
import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _sweep(make, filt, V, dtype=np.float64, kernel='auto', rel=False):
    _, s = make()
    s.dtype = np.dtype(dtype)
    s.kernel = kernel
    s.certified_filter = filt
    with np.errstate(all='ignore'):
        J, pol = s.value_iteration(np.asarray(V, dtype=float), report_time=False)
    return J, pol, s.last_policy_index, s


def _same(a, b):
    assert np.array_equal(a[0], b[0], equal_nan=True), 'J differs'
    assert np.array_equal(a[2], b[2]), 'policy index differs'
    assert np.array_equal(a[1], b[1], equal_nan=True), 'policy differs'


def _stock(n_x=96, n_y=9, n_w=7, cost_has_u=True, box_on_state=False):
    """stock x0 driven by the control, exogenous AR(1) y, perturbation on y only"""
    sysd = SysDescription((2, 1, 1), name='stock')
    sysd.dyn = lambda x, y, u, w: (x + 0.7 * u, 0.8 * y + w)
    if cost_has_u:
        sysd.cost = lambda x, y, u, w: (y - 0.3) * u + 0.2 * u * u + 0.05 * x
    else:
        sysd.cost = lambda x, y, u, w: 0.05 * x + y * y + 0.0 * u
    if box_on_state:
        sysd.control_box = lambda x, y: ((-1.0, 1.0 + 0.5 * y * y + 0.01 * x),)
    else:
        sysd.control_box = lambda x, y: ((-1.0, 1.0),)
    sysd.perturb_laws = [NormalLaw(0, 0.2)]
    s = DPSolver(sysd)
    s.discretize_state(0, 3, n_x, -1, 1, n_y)
    s.discretize_perturb(-0.5, 0.5, n_w)
    s.control_steps = (0.0625,)
    return sysd, s


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('N', [20, 33])
def test_benchmark_model_same_bits_with_and_without(gpu, N, dtype):
    make = lambda: models.synthetic3d(N=N)
    V = models.synthetic3d_V0(make()[1].state_grid)
    on, off = _sweep(make, True, V, dtype), _sweep(make, False, V, dtype)
    assert on[3].backend_info['certified_filter'] and not off[3].backend_info['certified_filter']
    assert on[3].backend_info['kernel'] == off[3].backend_info['kernel'] == 'column'
    _same(on, off)
    _same(on, _sweep(make, True, V, dtype, kernel='generic'))      # and the direct kernel


def test_against_the_numpy_oracle(gpu):
    from oracle import vi_numpy
    make = lambda: models.synthetic3d(N=20)
    V = np.random.default_rng(3).standard_normal((20, 20, 20))
    J, pol, idx, s = _sweep(make, True, V)
    assert s.backend_info['certified_filter']
    nodes = np.arange(0, V.size, 7)
    Jo, polo, idxo, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(s), V, nodes=nodes)
    assert np.array_equal(J.ravel()[nodes], Jo) and np.array_equal(idx.ravel()[nodes], idxo)


@pytest.mark.parametrize('box_on_state', [False, True])
def test_two_dimensional_stock_per_node_boxes_and_ragged_columns(gpu, box_on_state):
    make = lambda: _stock(box_on_state=box_on_state)
    V = np.random.default_rng(11).standard_normal(make()[1]._state_grid_shape)
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    assert on[3].backend_info['certified_filter']
    _same(on, off)


def test_exact_ties_keep_the_first_control(gpu):
    """a constant cost-to-go and a cost without the control: the controls of a node have the
    same expected cost up to the rounding of (1 - lam) v + lam v -- all of them survive the
    first pass, and the second decides like numpy's argmin (mostly the lowest index)"""
    make = lambda: _stock(cost_has_u=False)
    V = np.full(make()[1]._state_grid_shape, 2.5)
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    _same(on, off)
    assert (on[2] == 0).mean() > 0.5


@pytest.mark.parametrize('case', ['nan', 'inf', '-inf', 'huge', 'subnormal', 'mixed_scales'])
def test_special_values_take_the_long_way(gpu, case):
    make = lambda: _stock()
    shape = make()[1]._state_grid_shape
    V = np.random.default_rng(5).standard_normal(shape)
    if case == 'nan':
        V[10:14, 2:5] = np.nan
    elif case == 'inf':
        V[40:, :] = np.inf                       # a forbidden region
    elif case == '-inf':
        V[::7, 3] = -np.inf
    elif case == 'huge':
        V *= 1e302                               # the long path overflows here and there
    elif case == 'subnormal':
        V *= 1e-310
    else:
        V[30:50] *= 1e12                         # a penalty region next to ordinary values
    on, off = _sweep(make, True, V), _sweep(make, False, V)
    assert on[3].backend_info['certified_filter']
    _same(on, off)


@pytest.mark.parametrize('dtype,scale_v', [(np.float32, 2e38), (np.float64, 1e307)])
def test_tiny_weights_do_not_hide_an_overflow(gpu, dtype, scale_v):
    """weights of 1e-30: the weighted bounds of the first pass are small although the lerp of the
    reference's path overflows on extrapolated cells -- the raw magnitude of the table decides"""
    def make():
        sysd, s = _stock()
        s.perturb_proba = [np.asarray(s.perturb_proba[0]) * 1e-30]
        return sysd, s
    V = np.random.default_rng(8).standard_normal(make()[1]._state_grid_shape)
    V = V / np.abs(V).max() * scale_v
    on, off = _sweep(make, True, V, dtype), _sweep(make, False, V, dtype)
    assert on[3].backend_info['certified_filter']
    _same(on, off)


def test_weights_that_do_not_sum_to_one(gpu):
    def make():
        sysd, s = _stock()
        s.perturb_proba = [np.asarray(s.perturb_proba[0]) * 3.7]
        return sysd, s
    V = np.random.default_rng(2).standard_normal(make()[1]._state_grid_shape)
    _same(_sweep(make, True, V), _sweep(make, False, V))


@pytest.mark.parametrize('scale', ['1e6', '1e11', '1e18'])
def test_any_larger_radius_gives_the_same_bits(gpu, debug_defines, scale):
    """SDP_COL_FILTER_SCALE multiplies the error radius: more and more controls survive the
    first pass (all of them at 1e18) and go through the second; the result cannot change"""
    make = lambda: models.synthetic3d(N=24)
    V = models.synthetic3d_V0(make()[1].state_grid)
    ref = _sweep(make, False, V)
    debug_defines.set(SDP_COL_FILTER_SCALE=scale)
    on = _sweep(make, True, V)
    assert 'SDP_COL_FILTER_SCALE' in on[3]._kernel_plan()['source']
    _same(on, ref)


@pytest.mark.parametrize('n_x', [96, 600])
def test_two_survivors_take_the_short_cut(gpu, debug_defines, n_x):
    """SDP_COL_FILTER_TOP2 (default for 4-byte reals): the first pass also keeps the second
    best control, and a node with exactly two survivors evaluates those two -- one per lane
    of the node, or both on a lane that has the node to itself (600 nodes per column)"""
    make = lambda: _stock(n_x=n_x)
    V = np.random.default_rng(4).standard_normal(make()[1]._state_grid_shape)
    on, off = _sweep(make, True, V, np.float32), _sweep(make, False, V, np.float32)
    assert on[3].backend_info['certified_filter']
    _same(on, off)
    # the same path in 8-byte reals, with a radius wide enough to leave pairs
    debug_defines.set(SDP_COL_FILTER_TOP2='1')
    debug_defines.set(SDP_COL_FILTER_SCALE='3e10')
    on = _sweep(make, True, V)
    debug_defines.unset('SDP_COL_FILTER_TOP2')
    debug_defines.unset('SDP_COL_FILTER_SCALE')
    _same(on, _sweep(make, False, V))


def test_chained_sweeps_and_relative_dp(gpu):
    make = lambda: models.synthetic3d(N=24)
    _, a = make()
    _, b = make()
    b.certified_filter = False
    V = models.synthetic3d_V0(a.state_grid)
    V = V - V[a._state_ref_ind]
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        Ja, pa = a.value_iterations((V, 0.), 6, rel_dp=True)
        Jb, pb = b.value_iterations((V, 0.), 6, rel_dp=True)
    assert np.array_equal(Ja[0], Jb[0]) and Ja[1] == Jb[1] and np.array_equal(pa, pb)
    assert np.array_equal(a.last_policy_index, b.last_policy_index)


def _stock_cost_w(box_on_state=False, tilt=None):
    """the stock of _stock with a cost that sees the perturbation (x0' still does not)"""
    sysd, s = _stock(box_on_state=box_on_state)
    if tilt is None:
        sysd.cost = lambda x, y, u, w: (y + w) * u + 0.2 * u * u + 0.05 * x * (1.0 + w)
    else:                                        # flat in the control up to tilt u^2 (cf. _flat)
        sysd.cost = lambda x, y, u, w: (-1.37 * 0.7) * u * (1.0 + 0.0 * w) + w * (0.3 * y) + tilt * (u * u)
    s._cache.clear()
    return sysd, s


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('box_on_state', [False, True])
def test_a_cost_that_depends_on_the_perturbation_is_filtered_with_the_same_bits(gpu, dtype, box_on_state):
    """the first pass accumulates the cost's expectation with the reference's own g_w (sdp_col_cost_expect)"""
    make = lambda: _stock_cost_w(box_on_state)
    V = np.random.default_rng(21).standard_normal(make()[1]._state_grid_shape)
    on, off = _sweep(make, True, V, dtype), _sweep(make, False, V, dtype)
    assert on[3].backend_info['certified_filter'] and not off[3].backend_info['certified_filter']
    assert 'SDP_COST_HAS_W 1' in on[3]._kernel_plan()['source']
    _same(on, off)
    _same(on, _sweep(make, True, V, dtype, kernel='generic'))
    # special values: a forbidden region, a NaN patch
    V2 = V.copy()
    V2[40:, :] = np.inf
    V2[10:14, 2:5] = np.nan
    _same(_sweep(make, True, V2, dtype), _sweep(make, False, V2, dtype))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('scale', [None, '0.5'])
def test_near_ties_with_the_perturbation_in_the_cost(gpu, debug_defines, dtype, scale):
    s_ = 1.37
    for tilt in ([0.0, 1e-15, 1e-13] if dtype == np.float64 else [0.0, 1e-7, 1e-5]):
        make = lambda: _stock_cost_w(tilt=tilt)
        g = make()[1].state_grid
        V = s_ * np.asarray(g[0])[:, None] + np.cos(3 * np.asarray(g[1]))[None, :]
        off = _sweep(make, False, V, dtype)
        if scale:
            debug_defines.set(SDP_COL_FILTER_SCALE=scale)
        on = _sweep(make, True, V, dtype)
        if scale:
            debug_defines.unset('SDP_COL_FILTER_SCALE')
        _same(on, off)
        if tilt == 0.0:
            assert len(np.unique(off[2])) > 3


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_two_controls_on_one_lattice_go_through_the_control_table(gpu, dtype):
    """a 2-D control lattice shared by all nodes (flat index -> both control values in the table), the
    second control only in the cost; three tabulated values per control"""
    def make():
        sysd = SysDescription((2, 2, 1), name='stock with two controls')
        sysd.dyn = lambda x, y, u, v, w: (x + 0.7 * u, 0.8 * y + w)
        sysd.cost = lambda x, y, u, v, w: (y - 0.3) * u + 0.2 * u * u + 0.05 * x * (1.0 + v) + (v - 0.25 * y) * (v - 0.25 * y)
        sysd.control_box = lambda x, y: ((-1.0, 1.0), (0.0, 0.5))
        sysd.perturb_laws = [NormalLaw(0, 0.2)]
        s = DPSolver(sysd)
        s.discretize_state(0, 3, 80, -1, 1, 9)
        s.discretize_perturb(-0.5, 0.5, 7)
        s.control_steps = (0.125, 0.1)
        return sysd, s
    V = np.random.default_rng(31).standard_normal(make()[1]._state_grid_shape)
    on, off = _sweep(make, True, V, dtype), _sweep(make, False, V, dtype)
    assert on[3].backend_info['certified_filter']
    assert 'SDP_COL_UTAB' in on[3]._kernel_plan()['source']
    _same(on, off)
    _same(on, _sweep(make, True, V, dtype, kernel='generic'))


def test_where_the_filter_does_not_apply(gpu):
    """an x0' that sees the perturbation other than through a final sum (that form: test_gpu_shift.py), every control the long way"""
    sysd, s = _stock()
    sysd.dyn = lambda x, y, u, w: ((x + 0.7 * u) * (1.0 + 0.1 * w), 0.8 * y + w)
    s._cache.clear()
    s.value_iteration(np.zeros(s._state_grid_shape), report_time=False)
    assert s.backend_info['kernel'] == 'column' and not s.backend_info['certified_filter']


# ---------------------------------------------------------------------------
# Two-sided evidence for the error radius.  The tests above make the radius LARGER and see nothing
# change; these build inputs on which the reference's own choice hangs on the last bits of its
# W x 6 roundings, check that the filter still reproduces it at the proven radius and at HALF of
# it, and that a radius far too small IS noticed (so the checks are sensitive to the radius).
# ---------------------------------------------------------------------------
def _flat(tilt, n_x=192, n_y=6, n_w=7, box_on_state=False):
    """An objective that is flat in the control: cost-to-go V = s x (+ a function of y), cost
    -s b u + tilt u^2.  In real arithmetic every control of a node has the expected cost
    s x + E h(y') + tilt u^2: with tilt = 0 they all tie and the reference's argmin is decided by
    rounding noise alone; a tiny tilt grades the differences from 0 to a few ulp."""
    s_, b_ = 1.37, 0.0731
    sysd = SysDescription((2, 1, 1), name='flat objective')
    sysd.dyn = lambda x, y, u, w: (x + b_ * u, 0.8 * y + w)
    sysd.cost = lambda x, y, u, w: (-s_ * b_) * u + tilt * (u * u)
    if box_on_state:
        sysd.control_box = lambda x, y: ((-1.0, 1.0 + 0.01 * x),)
    else:
        sysd.control_box = lambda x, y: ((-1.0, 1.0),)
    sysd.perturb_laws = [NormalLaw(0, 0.2)]
    s = DPSolver(sysd)
    s.discretize_state(0, 3, n_x, -1, 1, n_y)
    s.discretize_perturb(-0.5, 0.5, n_w)
    s.control_steps = (2.0 / 47,)
    V = s_ * np.asarray(s.state_grid[0])[:, None] + np.cos(3 * np.asarray(s.state_grid[1]))[None, :]
    return sysd, s, V


NEAR_TIE_TILTS = {np.float64: [0.0, 1e-16, 1e-15, 4e-15, 3e-14, 1e-12],
                  np.float32: [0.0, 1e-8, 1e-7, 6e-7, 1e-5]}


@pytest.mark.parametrize('box_on_state', [False, True])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('scale', [None, '0.5'])
def test_near_ties_keep_the_bits_at_the_proven_radius_and_at_half_of_it(gpu, debug_defines, dtype, box_on_state, scale):
    for tilt in NEAR_TIE_TILTS[dtype]:
        make = lambda: _flat(tilt, box_on_state=box_on_state)[:2]
        V = _flat(tilt)[2]
        off = _sweep(make, False, V, dtype)
        if scale:
            debug_defines.set(SDP_COL_FILTER_SCALE=scale)
        on = _sweep(make, True, V, dtype)
        if scale:
            assert 'SDP_COL_FILTER_SCALE' in on[3]._kernel_plan()['source']
            debug_defines.unset('SDP_COL_FILTER_SCALE')
        assert on[3].backend_info['certified_filter'] and not off[3].backend_info['certified_filter']
        _same(on, off)
        if tilt == 0.0:                          # (the case is what it claims to be: no clear winner)
            assert len(np.unique(off[2])) > 3


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_a_radius_far_too_small_is_noticed(gpu, debug_defines, dtype):
    """With the radius cut by 1e6 the first pass picks its own minimum of F where the reference's
    minimum of E differs in the last bits: policy indices (and J, by an ulp) change.  If this test
    ever fails, the near-tie tests above have stopped probing the radius."""
    make = lambda: _flat(0.0)[:2]
    V = _flat(0.0)[2]
    off = _sweep(make, False, V, dtype)
    debug_defines.set(SDP_COL_FILTER_SCALE='1e-6')
    on = _sweep(make, True, V, dtype)
    assert 'SDP_COL_FILTER_SCALE' in on[3]._kernel_plan()['source']
    assert (on[2] != off[2]).sum() > 0
    # ... while every J it returns is still the reference's value of SOME control: within rounding noise
    assert np.allclose(on[0], off[0], rtol=1e-5 if dtype == np.float32 else 1e-13, atol=0)


# ---------------------------------------------------------------------------
# Resident-chunk form (csrc/sdp_colres_kernel.h, SDP_COL_WRES): the table holds a chunk of the
# perturbation points at a time -- more workgroups per CU -- and the tail is built twice; the second
# pass accumulates head then tail in w order.  Same bits as the plain filtered kernel and as the kernel
# without the filter.
# ---------------------------------------------------------------------------
def _wres(debug_defines, make, V, k, **kw):
    debug_defines.set(SDP_COL_WRES=str(k))
    try:
        out = _sweep(make, True, V, **kw)
        out[3].wres_source = out[3]._kernel_plan()['source']           # (the plan follows the switches: kept for the caller)
        assert ('#define SDP_COL_WRES {}'.format(k) in out[3].wres_source) == (k > 0)
    finally:
        debug_defines.unset('SDP_COL_WRES')
    return out


@pytest.mark.parametrize('case', ['benchmark', 'stock', 'stock_box_on_state', 'nan', 'inf', 'huge', 'ties', 'near_ties'])
def test_resident_chunks_give_the_same_bits(gpu, debug_defines, case):
    if case == 'benchmark':
        make, k = (lambda: models.synthetic3d(N=24)), 16                 # 32 points: 16 resident, 16 built twice
        V = models.synthetic3d_V0(make()[1].state_grid)
    else:
        make, k = (lambda: _stock(box_on_state=(case == 'stock_box_on_state'), cost_has_u=(case != 'ties'))), 4
        shape = make()[1]._state_grid_shape                               # 7 points: 4 resident, 3 built twice
        V = np.random.default_rng(5).standard_normal(shape)
        if case == 'nan':
            V[10:14, 2:5] = np.nan
        elif case == 'inf':
            V[40:, :] = np.inf
        elif case == 'huge':
            V *= 1e302
        elif case == 'ties':
            V = np.full(shape, 2.5)                                       # every control survives: the global path
        elif case == 'near_ties':
            V = 1.37 * np.asarray(make()[1].state_grid[0])[:, None] + 0.0 * V
    plain, off = _sweep(make, True, V), _sweep(make, False, V)
    res = _wres(debug_defines, make, V, k)
    assert res[3].backend_info['certified_filter']
    _same(res, plain)
    _same(res, off)


def test_resident_chunks_chained_sweeps_relative_dp_and_policy_evaluation(gpu, debug_defines):
    """the chunked fixed-policy kernel (sdp_evalpol_col of sdp_colres_kernel.h) and chains of sweeps"""
    make = lambda: models.synthetic3d(N=24)
    _, a = make()
    _, b = make()
    V = models.synthetic3d_V0(a.state_grid)
    debug_defines.set(SDP_COL_WRES='16')
    Ja, pa = b.value_iterations(V, 3, report_time=False)
    Ea, fa = b.eval_policy(pa, 4, rel_dp=True, report_time=False, J_ref_full=True)
    assert '#define SDP_COL_WRES 16' in b._kernel_plan()['source']
    debug_defines.unset('SDP_COL_WRES')
    Jb, pb = a.value_iterations(V, 3, report_time=False)
    Eb, fb = a.eval_policy(pb, 4, rel_dp=True, report_time=False, J_ref_full=True)
    assert np.array_equal(Ja, Jb) and np.array_equal(pa, pb)
    assert np.array_equal(Ea, Eb) and np.array_equal(fa, fb)


def test_resident_chunks_only_where_they_apply(gpu, debug_defines):
    """a column longer than a workgroup, 4-byte reals, a cost that sees the perturbation: the plain kernel"""
    debug_defines.set(SDP_COL_WRES='4')
    for make, dtype in ((lambda: _stock(n_x=600), np.float64), (lambda: _stock(), np.float32),
                        (lambda: _stock_cost_w(), np.float64)):
        _, s = make()
        s.dtype = np.dtype(dtype)
        assert 'SDP_COL_WRES' not in s._kernel_plan()['source']


# ---------------------------------------------------------------------------
# Short first pass of the resident-chunk kernel (SDP_COL_LEAN2, codegen.short_pass_source): x0' = X(x) +- a(u),
# cost = K(x) +- h(u).  K left out of the ordered value, the cost and the positions bounded from the column's
# control table, the index packed into the low mantissa bits.  Same bits as the first pass of section 3.1c
# (SDP_COL_LEAN2 = 0) and as the kernel without the filter.
# ---------------------------------------------------------------------------
def _shaped(form, n_x=96, n_y=9, n_w=7):
    sysd = SysDescription((2, 1, 1), name='stock, ' + form)
    dyn = {'add': lambda x, y, u, w: (x + 0.7 * u, 0.8 * y + w),
           'sub': lambda x, y, u, w: (x - 0.7 * u, 0.8 * y + w),
           'rsub': lambda x, y, u, w: (0.7 * u - (0.0 - x), 0.8 * y + w),
           'h_only': lambda x, y, u, w: (x + 0.7 * u, 0.8 * y + w),
           'no_u_cost': lambda x, y, u, w: (0.7 * u + x, 0.8 * y + w),
           'not_additive': lambda x, y, u, w: (x * (1.0 + 0.01 * u) + 0.7 * u, 0.8 * y + w)}[form]
    cost = {'add': lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 * u * u),
            'sub': lambda x, y, u, w: (0.05 * x + y) - ((0.3 - y) * u - 0.2 * u * u),
            'rsub': lambda x, y, u, w: ((y - 0.3) * u + 0.2 * u * u) - 0.05 * x,
            'h_only': lambda x, y, u, w: (y - 0.3) * u + 0.2 * u * u,
            'no_u_cost': lambda x, y, u, w: 0.05 * x + y * y,
            'not_additive': lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 * u * u)}[form]
    sysd.dyn, sysd.cost = dyn, cost
    sysd.control_box = lambda x, y: ((-1.0, 1.0),)
    sysd.perturb_laws = [NormalLaw(0, 0.2)]
    s = DPSolver(sysd)
    s.discretize_state(0, 3, n_x, -1, 1, n_y)
    s.discretize_perturb(-0.5, 0.5, n_w)
    s.control_steps = (0.0625,)
    return sysd, s


# (form 'add' with NaN / infinite / huge values, ties and near-ties: test_resident_chunks_give_the_same_bits above --
# its stock model has the shape, so it runs the short first pass)
@pytest.mark.parametrize('form,values', [(f, v) for f in ('sub', 'rsub', 'h_only', 'no_u_cost') for v in ('random', 'nan', 'ties')] +
                         [('add', 'random'), ('not_additive', 'random'), ('sub', 'inf'), ('sub', 'huge')])
def test_the_short_first_pass_gives_the_same_bits(gpu, debug_defines, form, values):
    make = lambda: _shaped(form)
    shape = make()[1]._state_grid_shape
    V = np.random.default_rng(11).standard_normal(shape)
    if values == 'nan':
        V[10:14, 2:5] = np.nan
    elif values == 'inf':
        V[40:, :] = np.inf
    elif values == 'huge':
        V *= 1e302
    elif values == 'ties':
        V = np.full(shape, 2.5)
    off = _sweep(make, False, V)
    short = _wres(debug_defines, make, V, 4)
    assert ('#define SDP_COL_LEAN2 1' in short[3].wres_source) == (form != 'not_additive')
    debug_defines.set(SDP_COL_LEAN2='0')
    try:
        long_ = _wres(debug_defines, make, V, 4)
        assert '#define SDP_COL_LEAN2 1' not in long_[3].wres_source
    finally:
        debug_defines.unset('SDP_COL_LEAN2')
    _same(short, off)
    _same(short, long_)


@pytest.mark.parametrize('special', ['nan_in_h', 'inf_in_h', 'nan_in_K', 'huge_a', 'extrapolation'])
def test_the_short_first_pass_with_special_values_in_the_model(gpu, debug_defines, special):
    """what the statistics of the control table must catch: a cost or a position that is not finite for SOME control /
    node; controls that leave the grid by many cells (L >> 1)"""
    def make():
        sysd, s = _shaped('add')
        if special == 'nan_in_h':
            sysd.cost = lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 * u * u) / (1.0 + 0.0 * np.sqrt(0.5 - u * u * u))
        elif special == 'inf_in_h':
            sysd.cost = lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 / (u * u))      # u = 0 is a lattice point
        elif special == 'nan_in_K':
            sysd.cost = lambda x, y, u, w: np.sqrt(x - 0.5) + ((y - 0.3) * u + 0.2 * u * u)
        elif special == 'huge_a':
            sysd.dyn = lambda x, y, u, w: (x + 1e300 * (u * u * u * u * u * u * u * u * u), 0.8 * y + w)
        elif special == 'extrapolation':
            sysd.dyn = lambda x, y, u, w: (x + 40.0 * u, 0.8 * y + w)
        return sysd, s
    V = np.random.default_rng(12).standard_normal(make()[1]._state_grid_shape)
    off = _sweep(make, False, V)
    short = _wres(debug_defines, make, V, 4)
    assert '#define SDP_COL_LEAN2 1' in short[3].wres_source
    _same(short, off)


@pytest.mark.parametrize('scale', [None, '0.5'])
def test_the_short_first_pass_on_near_ties(gpu, debug_defines, scale):
    for tilt in NEAR_TIE_TILTS[np.float64]:
        make = lambda: _flat(tilt)[:2]
        V = _flat(tilt)[2]
        off = _sweep(make, False, V)
        if scale:
            debug_defines.set(SDP_COL_FILTER_SCALE=scale)
        try:
            on = _wres(debug_defines, make, V, 4)
        finally:
            debug_defines.unset('SDP_COL_FILTER_SCALE')
        assert '#define SDP_COL_LEAN2 1' in on[3].wres_source
        _same(on, off)


def test_the_short_first_pass_notices_a_radius_far_too_small(gpu, debug_defines):
    make = lambda: _flat(0.0)[:2]
    V = _flat(0.0)[2]
    off = _sweep(make, False, V)
    debug_defines.set(SDP_COL_FILTER_SCALE='1e-6')
    try:
        on = _wres(debug_defines, make, V, 4)
    finally:
        debug_defines.unset('SDP_COL_FILTER_SCALE')
    assert '#define SDP_COL_LEAN2 1' in on[3].wres_source
    assert (on[2] != off[2]).sum() > 0
    assert np.allclose(on[0], off[0], rtol=1e-13, atol=0)


# the same shape with 4-byte reals: the short WIDE first pass of the full-table kernel (SDP_COL_WIDE2 of
# sdp_col_filter_nodes): F' in 8-byte arithmetic, one bound per node from the table's statistics and the column's
# largest |T|, both survivors' indices in the low mantissa bits
@pytest.mark.parametrize('form,values', [(f, v) for f in ('sub', 'rsub', 'no_u_cost') for v in ('random', 'nan')] +
                         [('h_only', 'random'), ('not_additive', 'random'), ('add', 'ties'), ('add', 'inf'), ('add', 'huge')])
def test_the_short_wide_first_pass_gives_the_same_bits(gpu, debug_defines, form, values):
    make = lambda: _shaped(form)
    shape = make()[1]._state_grid_shape
    V = np.random.default_rng(13).standard_normal(shape)
    if values == 'nan':
        V[10:14, 2:5] = np.nan
    elif values == 'inf':
        V[40:, :] = np.inf
    elif values == 'huge':
        V *= 1e37
    elif values == 'ties':
        V = np.full(shape, 2.5)
    off = _sweep(make, False, V, np.float32)
    short = _sweep(make, True, V, np.float32)
    assert ('#define SDP_COL_WIDE2 1' in short[3]._kernel_plan()['source']) == (form != 'not_additive')
    debug_defines.set(SDP_COL_LEAN2='0')
    try:
        long_ = _sweep(make, True, V, np.float32)
        assert 'SDP_COL_WIDE2' not in long_[3]._kernel_plan()['source']
    finally:
        debug_defines.unset('SDP_COL_LEAN2')
    _same(short, off)
    _same(short, long_)


@pytest.mark.parametrize('special', ['nan_in_h', 'inf_in_h', 'nan_in_K', 'huge_a', 'extrapolation'])
def test_the_short_wide_first_pass_with_special_values_in_the_model(gpu, special):
    def make():
        sysd, s = _shaped('add')
        if special == 'nan_in_h':
            sysd.cost = lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 * u * u) / (1.0 + 0.0 * np.sqrt(0.5 - u * u * u))
        elif special == 'inf_in_h':
            sysd.cost = lambda x, y, u, w: 0.05 * x + ((y - 0.3) * u + 0.2 / (u * u))
        elif special == 'nan_in_K':
            sysd.cost = lambda x, y, u, w: np.sqrt(x - 0.5) + ((y - 0.3) * u + 0.2 * u * u)
        elif special == 'huge_a':
            sysd.dyn = lambda x, y, u, w: (x + 1e37 * (u * u * u * u * u * u * u * u * u), 0.8 * y + w)
        elif special == 'extrapolation':
            sysd.dyn = lambda x, y, u, w: (x + 40.0 * u, 0.8 * y + w)
        return sysd, s
    V = np.random.default_rng(14).standard_normal(make()[1]._state_grid_shape)
    off = _sweep(make, False, V, np.float32)
    short = _sweep(make, True, V, np.float32)
    assert '#define SDP_COL_WIDE2 1' in short[3]._kernel_plan()['source']
    _same(short, off)


def test_the_short_wide_first_pass_when_lanes_share_a_node(gpu):
    """a column shorter than the workgroup's waves x 64: the lattice of a node is cut into ranges, a lane each, and
    the three smallest values of the ranges meet through shuffles -- indices still in their low bits"""
    make = lambda: _stock(n_x=40)
    V = np.random.default_rng(15).standard_normal(make()[1]._state_grid_shape)
    off = _sweep(make, False, V, np.float32)
    short = _sweep(make, True, V, np.float32)
    assert '#define SDP_COL_WIDE2 1' in short[3]._kernel_plan()['source']
    _same(short, off)


def test_a_budget_of_64_registers_gives_the_same_bits(gpu, debug_defines):
    """Round 5: with 8 waves per SIMD asked of the register allocator (64 registers; what the planner chose for small
    tables until then) the compiler stored a spilled threadIdx.x before it restored the execution mask, and the kernel
    of this very configuration never ended.  The build now scans for that pattern and rebuilds with more registers
    (codegen.spill_hazards, _native.compile_model; tests/test_trace_codegen.py checks the build log): forced back to
    that budget, the kernel ends and gives the bits of the kernel without the filter."""
    make = lambda: _shaped('no_u_cost')
    V = np.random.default_rng(11).standard_normal(make()[1]._state_grid_shape)
    off = _sweep(make, False, V)
    debug_defines.set(SDP_COL_MIN_WAVES='8', SDP_COL_LEAN2='0')
    try:
        forced = _wres(debug_defines, make, V, 4)
        assert '#define SDP_COL_MIN_WAVES 8' in forced[3].wres_source
    finally:
        debug_defines.unset('SDP_COL_MIN_WAVES', 'SDP_COL_LEAN2')
    _same(forced, off)
