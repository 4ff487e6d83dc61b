"""Randomised models for the two filter forms added in round 3 (tracer + code generator + kernels against the
kernels that do no filtering): random expressions built from correctly rounded operators, so every family must
give the same bits.
* shifted lattice (csrc/sdp_colfilter_kernel.h, SDP_COL_SHIFT): x0' = a(x, y, u) +- b_1(y, w) [+- b_2(y, w)];
* reduced array (csrc/sdp_lead_kernel.h): two controlled stocks, one exogenous variable, two controls."""
import numpy as np
import pytest

from stodynprog_amd import SysDescription, DPSolver, models
from tests.test_gpu_sweep import _random_expr

pytestmark = pytest.mark.gpu


def _run(solver, V, **attrs):
    for k, v in attrs.items():
        setattr(solver, k, v)
    with np.errstate(all='ignore'):
        J, pol = solver.value_iteration(V, report_time=False)
    return J, pol, solver.last_policy_index


@pytest.mark.parametrize('seed', range(6))
def test_random_stocks_with_the_perturbation_in_a_final_sum(gpu, seed):
    rng = np.random.default_rng(7000 + seed)
    a = _random_expr(rng, ['x', 'y', 'u'], 2)
    b1 = _random_expr(rng, ['y', 'w'], 2)
    b2 = _random_expr(rng, ['y', 'w'], 1)
    trail = _random_expr(rng, ['y', 'w'], 3)
    cst = _random_expr(rng, ['x', 'y', 'u', 'w'] if seed % 2 else ['x', 'y', 'u'], 3)
    op1, op2 = ('+', '-')[seed % 2], ('-', '+')[(seed // 2) % 2]
    chain = '(x + 0.3 * u + 0.1 * ({})) {} 0.2 * (w + 0.3 * ({}))'.format(a, op1, b1)
    if seed >= 3:
        chain = '({}) {} 0.05 * ({})'.format(chain, op2, b2)
    ns = {'np': np}
    exec('def dyn(x, y, u, w):\n    return ({}, 0.5 * y + 0.2 * ({}))\n'
         'def cost(x, y, u, w):\n    return {} + 0.3 * u * u\n'.format(chain, trail, cst), ns)

    def make():
        s = SysDescription((2, 1, 1), name='fuzz shift %d' % seed)
        s.dyn, s.cost = ns['dyn'], ns['cost']
        s.control_box = lambda x, y: ((-1., 1.),)
        s.perturb_laws = [models.NormalLaw(0, 0.3)]
        solver = DPSolver(s)
        solver.discretize_state(-1, 1, 70, -1, 1, 9)
        solver.discretize_perturb(-0.6, 0.6, 5)
        solver.control_steps = (0.125,)
        return solver
    on = make()
    model = on._traced()
    assert model.storage_separable and model.lead_split() is not None, chain
    V = rng.standard_normal((70, 9))
    r_on = _run(on, V)
    assert on.backend_info['filter_form'] == 'shifted lattice', chain
    r_off = _run(make(), V, certified_filter=False)
    r_gen = _run(make(), V, kernel='generic')
    for other in (r_off, r_gen):
        assert np.array_equal(r_on[0], other[0], equal_nan=True), (chain, trail, cst)
        assert np.array_equal(r_on[2], other[2]), (chain, trail, cst)
    # and on a smooth cost-to-go (one survivor per node instead of many)
    g = on.state_grid
    V = 0.3 * (np.asarray(g[0])[:, None] - 0.2) ** 2 + np.cos(2 * np.asarray(g[1]))[None, :]
    r_on, r_off = _run(make(), V), _run(make(), V, certified_filter=False)
    assert np.array_equal(r_on[0], r_off[0], equal_nan=True) and np.array_equal(r_on[2], r_off[2])


@pytest.mark.parametrize('seed', range(6))
def test_random_models_with_two_stocks(gpu, seed):
    rng = np.random.default_rng(9000 + seed)
    la = _random_expr(rng, ['a', 'b', 'y', 'u', 'v'], 3)
    lb = _random_expr(rng, ['a', 'b', 'y', 'u', 'v'], 3)
    ty = _random_expr(rng, ['y', 'w'], 3)
    cst = _random_expr(rng, ['a', 'b', 'y', 'u', 'v', 'w'] if seed % 2 else ['a', 'b', 'y', 'u', 'v'], 4)
    ns = {'np': np}
    exec('def dyn(a, b, y, u, v, w):\n    return (0.6 * a + 0.3 * u + 0.2 * ({}), 0.6 * b + 0.3 * v + 0.2 * ({}), 0.5 * y + 0.2 * ({}))\n'
         'def cost(a, b, y, u, v, w):\n    return {} + 0.2 * u * u + 0.1 * v * v\n'.format(la, lb, ty, cst), ns)

    def make():
        s = SysDescription((3, 2, 1), name='fuzz lead %d' % seed)
        s.dyn, s.cost = ns['dyn'], ns['cost']
        s.control_box = lambda a, b, y: ((-1., 1.), (-0.5, 0.5))
        s.perturb_laws = [models.NormalLaw(0, 0.3)]
        solver = DPSolver(s)
        solver.discretize_state(-1, 1, 11, -1, 1, 9, -1, 1, 7)
        solver.discretize_perturb(-0.6, 0.6, 5)
        solver.control_steps = (0.25, 0.25)
        return solver
    lead = make()
    assert lead._traced().controlled_axes() == 2
    V = rng.standard_normal((11, 9, 7))
    r_lead = _run(lead, V)
    assert lead.backend_info['kernel'] == 'lead'
    r_gen = _run(make(), V, kernel='generic')
    assert np.array_equal(r_lead[0], r_gen[0], equal_nan=True), (la, lb, ty, cst)
    assert np.array_equal(r_lead[2], r_gen[2]), (la, lb, ty, cst)
    assert np.array_equal(r_lead[1], r_gen[1], equal_nan=True)
