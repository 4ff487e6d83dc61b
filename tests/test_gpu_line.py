"""The filtered line kernel (csrc/sdp_line_kernel.h, round 6): ONE state variable whose perturbation enters the next state
through final sums -- the reference's tutorial, `x + u - w` (doc/example_inventory.py:31-33) -- with the certified filter
on the shifted lattice and the value array itself as the table.  J, policy and index must be the direct kernel's and the
numpy oracle's bit for bit: the filter only decides which controls are never evaluated."""
import numpy as np
import pytest

from stodynprog_amd import SysDescription, DPSolver, models
from oracle import vi_numpy

pytestmark = pytest.mark.gpu


def shop(n_x=600, n_u=257, n_w=16, dyn=None, cost=None, box=None, grid=(-8., 24.), wgrid=(0., 4.), law=None, steps=None):
    sysd = SysDescription((1, 1, 1), name='shop')
    sysd.dyn = dyn or (lambda x, u, w: (x + u - w,))
    sysd.cost = cost or (lambda x, u, w: np.where(x > 0, x * 0.5, -x * 3.) + u * 1.)
    sysd.control_box = box or (lambda x: ((0., 8.),))
    sysd.perturb_laws = [law or models.NormalLaw(2.0, 0.8)]
    s = DPSolver(sysd)
    s.discretize_state(grid[0], grid[1], n_x)
    s.discretize_perturb(wgrid[0], wgrid[1], n_w)
    s.control_steps = steps or (8. / (n_u - 1),)
    return s


def chain(make, kernel, V0, sweeps=3, debug=None):
    s = make()
    s.kernel = kernel
    s.debug_defines = debug
    out, V = [], V0
    with np.errstate(all='ignore'):
        for _ in range(sweeps):
            J, pol = s.value_iteration(V, report_time=False)
            out.append((J.copy(), pol.copy(), s.last_policy_index.copy()))
            V = J
    return out, s.backend_info


def same(a, b):
    return all(np.array_equal(x[0], y[0], equal_nan=True) and np.array_equal(x[1], y[1], equal_nan=True) and np.array_equal(x[2], y[2])
               for x, y in zip(a, b))


V_OF = {
    'zeros': lambda x, rng: np.zeros_like(x),
    'smooth': lambda x, rng: 0.3 * (x - 3) ** 2 + np.sin(x),
    'random': lambda x, rng: rng.standard_normal(x.size),
    'kinked': lambda x, rng: np.abs(x - 1.3) * 2 + np.maximum(x - 7, 0) ** 2,
}


@pytest.mark.parametrize('vname', sorted(V_OF))
@pytest.mark.parametrize('size', [(600, 257, 16), (3000, 513, 32), (150, 65, 9), (5000, 33, 5)])
def test_the_line_kernel_gives_the_direct_kernels_bits(gpu, size, vname):
    make = lambda: shop(*size)
    x = np.linspace(-8, 24, size[0])
    V0 = V_OF[vname](x, np.random.default_rng(size[0]))
    ref, info_ref = chain(make, 'generic', V0)
    got, info = chain(make, 'line', V0)
    assert info['kernel'] == 'line' and info['filter_form'] == 'shifted lattice' and info_ref['kernel'] == 'generic'
    assert same(ref, got)


def test_the_planner_takes_the_line_kernel_for_the_fine_inventory(gpu):
    _, s = models.inventory_fine()
    J, pol = s.value_iteration(np.zeros(600), report_time=False)
    assert s.backend_info['kernel'] == 'line' and s.backend_info['filter_form'] == 'shifted lattice'
    spec = vi_numpy.Spec.from_solver(s)
    nodes = np.arange(0, 600, 7)
    Jo, po, io, _ = vi_numpy.value_iteration(spec, np.zeros(600), nodes=nodes)
    assert np.array_equal(J[nodes], Jo) and np.array_equal(pol[nodes], po) and np.array_equal(s.last_policy_index[nodes], io)
    # the tutorial itself (10 nodes x 11 controls x 4 w) stays on the direct kernel: two launches are not worth it
    _, t = models.inventory()
    t.value_iteration(np.zeros(10), report_time=False)
    assert t.backend_info['kernel'] == 'generic'


def test_against_the_numpy_oracle_on_every_node_of_a_small_problem(gpu):
    for kw in (dict(n_x=120, n_u=41, n_w=7), dict(n_x=90, n_u=33, n_w=5, grid=(-3., 6.)),        # a span that is no power of two
               dict(n_x=64, n_u=17, n_w=3, grid=(1000., 1002.), wgrid=(0., .5), box=lambda x: ((0., 1.),), steps=(1. / 16,))):
        s = shop(**kw)
        s.kernel = 'line'
        rng = np.random.default_rng(kw['n_x'])
        V = rng.standard_normal(kw['n_x'])
        for _ in range(2):
            J, pol = s.value_iteration(V, report_time=False)
            Jo, po, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(s), V)
            assert np.array_equal(J, Jo) and np.array_equal(pol, po) and np.array_equal(s.last_policy_index, io)
            V = J


@pytest.mark.parametrize('case', ['per-node box', 'two terms', 'nested x+(w-u)', 'nested (x-w)+u', 'stock leaves the grid',
                                  'weights', 'wide shifts', 'three rows', 'two controls'])
def test_shapes_of_models(gpu, case):
    kw = dict(n_x=400, n_u=129, n_w=9)
    if case == 'per-node box':
        kw['box'] = lambda x: ((np.max((0., -x)), np.min((8., 24. - x))),)
    elif case == 'two terms':
        kw['dyn'] = lambda x, u, w: (x + u - 0.5 * w - 0.25 * w * w,)
    elif case == 'nested x+(w-u)':
        kw['dyn'] = lambda x, u, w: (x + (-w + u),)
    elif case == 'nested (x-w)+u':
        kw['dyn'] = lambda x, u, w: ((x - w) + u,)
    elif case == 'stock leaves the grid':
        kw.update(grid=(0., 4.), box=lambda x: ((-6., 6.),), steps=(12. / 128,))
    elif case == 'weights':
        class Odd(object):
            def pdf(self, w):
                return np.where(w < 1.0, -0.3, 1.7) * (1 + 0.1 * w)         # negative weights, a sum far from one
        kw['law'] = Odd()
    elif case == 'wide shifts':
        kw.update(wgrid=(-20., 30.), n_w=12)                                  # shifts of more rows than the axis has
    elif case == 'three rows':
        kw.update(n_x=3, n_u=33)
    elif case == 'two controls':
        def make():
            sysd = SysDescription((1, 2, 1), name='two orders')
            sysd.dyn = lambda x, u, v, w: (x + (u + 0.5 * v) - w,)
            sysd.cost = lambda x, u, v, w: np.where(x > 0, x * 0.5, -x * 3.) + u * 1. + v * 0.45 + 0.01 * v * v
            sysd.control_box = lambda x: ((0., 4.), (0., 6.))
            sysd.perturb_laws = [models.NormalLaw(2.0, 0.8)]
            s = DPSolver(sysd)
            s.discretize_state(-8., 24., 300)
            s.discretize_perturb(0., 4., 9)
            s.control_steps = (0.25, 0.5)
            return s
    if case != 'two controls':
        make = lambda: shop(**kw)
    n = make()._state_grid_shape[0]
    x = np.asarray(make().state_grid[0])
    for vname in ('smooth', 'random'):
        V0 = V_OF[vname](x, np.random.default_rng(3))
        ref, _ = chain(make, 'generic', V0)
        got, info = chain(make, 'line', V0)
        assert info['kernel'] == 'line'
        assert same(ref, got), (case, vname)


@pytest.mark.parametrize('special', ['nan', 'inf', '-inf', '1e302', 'subnormal', 'constant'])
def test_special_values_in_the_cost_to_go(gpu, special):
    make = lambda: shop(500, 129, 9)
    x = np.linspace(-8, 24, 500)
    V0 = 0.3 * (x - 3) ** 2
    if special == 'nan':
        V0[[40, 250, 251, 499]] = np.nan
    elif special == 'inf':
        V0[[0, 300]] = np.inf
    elif special == '-inf':
        V0[[120]] = -np.inf
    elif special == '1e302':
        V0 = V0 * 1e302
    elif special == 'subnormal':
        V0 = V0 * 1e-310
    else:
        V0[:] = 7.25
    ref, _ = chain(make, 'generic', V0, sweeps=2)
    got, _ = chain(make, 'line', V0, sweeps=2)
    assert same(ref, got)


def test_any_larger_radius_gives_the_same_bits_and_one_far_too_small_is_noticed(gpu):
    """SDP_LINE_FILTER_SCALE multiplies every half-width: larger ones only send more controls through the second level and
    the long way (1e18: every control of every node the long way); one a million times too small skips controls it must not"""
    make = lambda: shop(800, 513, 16)
    x = np.linspace(-8, 24, 800)
    V0 = 0.3 * (x - 3) ** 2 + np.sin(3 * x)
    ref, _ = chain(make, 'generic', V0)
    for scale in ('1e3', '1e6', '1e9', '1e18'):
        got, info = chain(make, 'line', V0, debug={'SDP_LINE_FILTER_SCALE': scale})
        assert info['debug_defines'] == {'SDP_LINE_FILTER_SCALE': scale}
        assert same(ref, got), scale
    got, _ = chain(make, 'line', V0, debug={'SDP_LINE_FILTER_SCALE': '1e-6'})
    assert not same(ref, got)                                # (the chord bound is what decides on a curved cost-to-go)


def test_flat_objective_near_ties(gpu):
    """a cost-to-go linear in the stock and a cost linear in the control with the opposite slope: every control of a node gives
    the same value up to rounding -- the reference's argmin hangs on rounding noise, the filter must keep all of them"""
    def make():
        return shop(300, 65, 5, cost=lambda x, u, w: u * 2.0 + 0. * x, box=lambda x: ((0., 4.),), steps=(4. / 64,), wgrid=(0., 2.))
    x = np.linspace(-8, 24, 300)
    V0 = -2.0 * x
    ref, _ = chain(make, 'generic', V0, sweeps=2)
    got, _ = chain(make, 'line', V0, sweeps=2)
    assert same(ref, got)
    got, _ = chain(make, 'line', V0, sweeps=2, debug={'SDP_LINE_FILTER_SCALE': '0.5'})       # half of the proven bounds: still the same bits
    assert same(ref, got)
