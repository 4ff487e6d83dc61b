"""Parity of the HIP value-iteration path (through DPSolver -> C ABI) with the
reference: golden vectors produced by the real reference for every BASELINE
config, the CPU oracle on the same inputs, and size-independent properties at
full benchmark size.  Bars: fp64 |dJ|/|J| < 1e-10, argmin indices exact outside
near-ties, control values bit-identical where the index agrees."""
import io
import contextlib

import numpy as np
import pytest

from conftest import golden, assert_sweep_parity
from oracle import c_oracle, vi_numpy
from stodynprog_amd import SysDescription, DPSolver, models
from stodynprog_amd.trace import TraceError

pytestmark = pytest.mark.gpu


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def check_policy_values(pol, idx, pol_ref, idx_ref):
    same = np.asarray(idx).astype(np.int64) == np.asarray(idx_ref).astype(np.int64)
    assert np.array_equal(np.asarray(pol)[same], np.asarray(pol_ref)[same])


# ---------------------------------------------------------------- config 1
def test_inventory_tutorial(gpu):
    g = golden('g2_inventory')
    _, solver = models.inventory()
    J = np.zeros(10)
    for k in range(6):
        J_prev = J
        J, u = solver.value_iteration(J, report_time=False)
        assert solver.backend_info['mode'] == 'traced'
        assert_sweep_parity(J, solver.last_policy_index, g['J'][k], g['idx'][k], g['margin'][k],
                            'inventory sweep %d' % k, prove=(solver, J_prev))
        assert np.array_equal(u, g['pol'][k])
        if k == 0:      # doc/example_inventory.rst:220-222
            assert np.allclose(J, [9, 6, 3, 0, 0.5, 1, 1.5, 2, 2.5, 3], rtol=0, atol=1e-14)
            assert np.array_equal(u[..., 0], np.zeros(10))
        if k == 1:      # :231
            assert np.array_equal(u[..., 0], [4, 3, 2, 1, 0, 0, 0, 0, 0, 0])
        if k in (2, 3):  # :235, :239
            assert np.array_equal(u[..., 0], [5, 4, 3, 2, 1, 0, 0, 0, 0, 0])


def test_value_iteration_prints_like_the_reference(gpu, capsys):
    _, solver = models.inventory()
    solver.value_iteration(np.zeros(10))
    out = capsys.readouterr().out
    assert out.startswith('value iteration...')
    assert '\rvalue iteration run in 0.' in out


# ---------------------------------------------------------------- NaS demo
def test_value_iterations_device_resident_loop(gpu):
    """DPSolver.value_iterations (extension): n sweeps with the cost-to-go kept
    on the device = the user's loop over value_iteration, bit for bit"""
    g = golden('g2_inventory')
    _, inv = models.inventory()
    J, pol = quiet(inv.value_iterations, np.zeros(10), 6)
    assert_sweep_parity(J, inv.last_policy_index, g['J'][5], g['idx'][5], g['margin'][5], 'inventory x6')
    assert np.array_equal(pol, g['pol'][5])
    for name, kw in (('synthetic3d', dict(N=20)), ('storage_ar1', dict(n_E=21, n_P=15, steps=(0.1, 0.1)))):
        _, a = getattr(models, name)(**kw)
        _, b = getattr(models, name)(**kw)
        V = np.random.default_rng(8).standard_normal(a._state_grid_shape)
        V -= V[a._state_ref_ind]
        Ja, refs_a = V, []
        for k in range(4):
            (Ja, r), pa = a.value_iteration((Ja, 0.), rel_dp=True, report_time=False)
            refs_a.append(r)
        (Jb, refs_b), pb = quiet(b.value_iterations, (V, 0.), 4, True, J_ref_full=True)
        assert np.array_equal(Ja, Jb) and np.array_equal(pa, pb) and np.array_equal(refs_a, refs_b)
        assert np.array_equal(a.last_policy_index, b.last_policy_index)
        Jc, pc = quiet(b.value_iterations, V, 1)
        Jd, pd = a.value_iteration(V, report_time=False)
        assert np.array_equal(Jc, Jd) and np.array_equal(pc, pd)
    # untraceable callables take the sweep-by-sweep path
    s = SysDescription((1, 1, 1))

    def dyn(x, u, w):
        return (x + np.asarray(u).reshape(np.shape(u)) - w,)        # needs a concrete array
    s.dyn = dyn
    s.cost = lambda x, u, w: np.where(x > 0, 0.5 * x, -3. * x) + u
    s.control_box = lambda x: ((0., 4.),)
    s.perturb_laws = [models.DiscreteLaw([0, 1, 2], [0.3, 0.4, 0.3])]
    t = DPSolver(s)
    t.discretize_state(-2, 5, 8)
    t.discretize_perturb(0, 2, 3)
    t.control_steps = (1.,)
    assert isinstance(t._traced(), TraceError)
    J2, p2 = quiet(t.value_iterations, np.zeros(8), 2)
    J1, _ = t.value_iteration(np.zeros(8), report_time=False)
    J1, p1 = t.value_iteration(J1, report_time=False)
    assert np.array_equal(J1, J2) and np.array_equal(p1, p2)


def test_nas_demo_two_sweeps(gpu):
    g = golden('g7_nas')
    _, solver = models.nas_demo()
    J1, u1 = solver.value_iteration(np.zeros((51, 41)), report_time=False)
    assert solver.backend_info['box_per_node']
    assert_sweep_parity(J1, solver.last_policy_index, g['J1'], g['idx1'], g['margin1'], 'nas 1',
                        prove=(solver, np.zeros((51, 41))))
    check_policy_values(u1, solver.last_policy_index, g['pol1'], g['idx1'])
    J2, u2 = solver.value_iteration(g['J1'], report_time=False)
    assert_sweep_parity(J2, solver.last_policy_index, g['J2'], g['idx2'], g['margin2'], 'nas 2',
                        prove=(solver, g['J1']))
    check_policy_values(u2, solver.last_policy_index, g['pol2'], g['idx2'])


# ---------------------------------------------------------------- config 2
def test_storage_ar1_reference_size(gpu):
    """41 x 61 nodes, 4001..8001 x 1 controls, 9 perturbations: the notebook problem"""
    g = golden('g3_ar1_ref')
    _, solver = models.storage_ar1()
    J1, u1 = solver.value_iteration(np.zeros((41, 61)), report_time=False)
    assert solver.backend_info['max_controls'] == 8001
    assert_sweep_parity(J1, solver.last_policy_index, g['J1'], g['idx1'], g['margin1'], 'ar1 1',
                        prove=(solver, np.zeros((41, 61))))
    check_policy_values(u1, solver.last_policy_index, g['pol1'], g['idx1'])
    J2, u2 = solver.value_iteration(g['J1'], report_time=False)
    # dead-band cost: exact ties may resolve differently (np.inner's BLAS summation order
    # vs sequential); every differing node is proved a tie by the oracle
    assert_sweep_parity(J2, solver.last_policy_index, g['J2'], g['idx2'], g['margin2'], 'ar1 2',
                        prove=(solver, g['J1']))
    check_policy_values(u2, solver.last_policy_index, g['pol2'], g['idx2'])
    (J3, J3ref), u3 = solver.value_iteration((g['Jd'], 0.), rel_dp=True, report_time=False)
    assert_sweep_parity(J3, solver.last_policy_index, g['J3'], g['idx3'], g['margin3'], 'ar1 3',
                        prove=(solver, g['Jd']))
    assert abs(J3ref - float(g['J3ref'])) <= 1e-12 * abs(float(g['J3ref']))
    assert J3[solver._state_ref_ind] == 0.0


def test_storage_ar1_config2_200x200(gpu):
    g = golden('g3_ar1_c2')
    _, solver = models.storage_ar1(n_E=200, n_P=200, steps=(8. / 49, 0.1))
    x0 = solver.state_grid[0].reshape(-1, 1)
    x1 = solver.state_grid[1].reshape(1, -1)
    V0 = 0.05 * (x0 - 4.) * (x0 - 4.) + 0.3 * (x1 * x1) + 0.02 * x0 * x1
    J, u = solver.value_iteration(V0, report_time=False)
    assert solver.backend_info['max_controls'] == 51      # 8/(8/49) rounds just above 49
    assert_sweep_parity(J, solver.last_policy_index, g['J'], g['idx'], g['margin'], 'ar1 c2',
                        prove=(solver, V0))
    check_policy_values(u[..., 0], solver.last_policy_index, g['pol0'], g['idx'])


# ---------------------------------------------------------------- config 3
def _searev_V0(solver):
    E, S, A = solver.state_grid_full
    return np.ascontiguousarray(0.02 * (E - 5.) * (E - 5.) + 1.5 * (S * S) + 0.7 * (A * A)
                                + 0.1 * S * A - 0.01 * E)


def test_searev_config3_128cubed(gpu):
    g = golden('g4_searev')
    _, solver = models.searev(n_E=128, n_S=128, n_A=128, step=2.2 / 31)
    V0 = _searev_V0(solver)
    J, u = solver.value_iteration(V0, report_time=False)
    assert solver.backend_info['max_controls'] == int(g['npts'].max())
    nodes = g['nodes']
    idx = solver.last_policy_index.ravel()[nodes]
    assert_sweep_parity(J.ravel()[nodes], idx, g['J'], g['idx'], g['margin'], 'searev c3',
                        prove=(solver, V0), nodes=nodes)
    check_policy_values(u.reshape(-1, 1)[nodes], idx, g['pol'], g['idx'])


def test_searev_reference_size(gpu):
    """31 x 61 x 61 nodes with 1101 / 2201 controls (control_steps = 0.001)"""
    g = golden('g4_searev')
    _, solver = models.searev()
    V0 = _searev_V0(solver)
    J, u = solver.value_iteration(V0, report_time=False)
    assert solver.backend_info['max_controls'] == 2201
    nodes = g['nodes_s']
    idx = solver.last_policy_index.ravel()[nodes]
    assert_sweep_parity(J.ravel()[nodes], idx, g['J_s'], g['idx_s'], g['margin_s'], 'searev ref',
                        prove=(solver, V0), nodes=nodes)
    check_policy_values(u.reshape(-1, 1)[nodes], idx, g['pol_s'], g['idx_s'])
    lo, hi, n = solver._box_table()
    assert np.array_equal(n[0][nodes], g['npts_s'][:, 0])


# ---------------------------------------------------------------- config 4
def test_synthetic_benchmark_problem_full_size(gpu):
    """256^3 x 64 x 32 fp64 (the bench.py workload): 4099 nodes against the
    reference, 20000 more against the C oracle, plus determinism."""
    g = golden('g5_synth')
    _, solver = models.synthetic3d()
    V0 = models.synthetic3d_V0(solver.state_grid)
    J, u = solver.value_iteration(V0, report_time=False)
    idx = solver.last_policy_index
    info = solver.backend_info
    assert info['mode'] == 'traced' and info['lanes_per_node'] == 64 and not info['box_per_node']
    nodes = g['nodes']
    _, ndiff = assert_sweep_parity(J.ravel()[nodes], idx.ravel()[nodes], g['J'], g['idx'],
                                   g['margin'], 'synthetic c4 vs reference')
    assert ndiff == 0
    assert np.array_equal(u.reshape(-1, 1)[nodes], g['pol'])
    more = np.random.default_rng(7).integers(0, V0.size, 20000)
    Jo, io, _ = c_oracle.vi_synth3d(solver.state_grid, V0, models.SYNTH_PAR, -1., 1., 64,
                                    solver.perturb_grid[0], solver.perturb_proba[0],
                                    node_ids=more, n_threads=8)
    assert np.array_equal(J.ravel()[more], Jo)           # same summation order: bit-exact
    assert np.array_equal(idx.ravel()[more], io)
    # size-independent properties: every index valid, controls on the lattice, rerun identical
    assert idx.min() >= 0 and idx.max() <= 63
    ugrid = np.linspace(-1., 1., 64)
    assert np.array_equal(u[..., 0], ugrid[idx])
    J_again, _ = solver.value_iteration(V0, report_time=False)
    assert np.array_equal(J, J_again)


def test_synthetic_small_full_grid_two_sweeps_and_slabs(gpu):
    g = golden('g5_synth')
    _, solver = models.synthetic3d(N=20)
    V0 = models.synthetic3d_V0(solver.state_grid)
    J1, u1 = solver.value_iteration(V0, report_time=False)
    assert_sweep_parity(J1, solver.last_policy_index, g['s_J1'], g['s_idx1'], g['s_margin1'], 's1',
                        prove=(solver, V0))
    assert np.array_equal(u1, g['s_pol1'])
    J2, u2 = solver.value_iteration(g['s_J1'], report_time=False)
    assert_sweep_parity(J2, solver.last_policy_index, g['s_J2'], g['s_idx2'], g['s_margin2'], 's2',
                        prove=(solver, g['s_J1']))
    # slab invariance: sweeping two node ranges separately gives the same bits
    # (this is what each rank does in the multi-GPU sweep)
    from stodynprog_amd.solver import _DeviceProblem
    prob = solver._problem()
    full = prob.get_value().ravel()
    parts = np.zeros_like(full)
    for lo, hi in ((0, 3333), (3333, 8000)):
        sub = _DeviceProblem(prob._keep, solver.backend_info['module'], np.float64, (20, 20, 20),
                             1, 32, 64, False, (lo, hi))
        sub.set_value(g['s_J1'])
        sub.sweep()
        parts[lo:hi] = sub.get_value().ravel()[lo:hi]
        sub.close()
    assert np.array_equal(parts, full)


# ---------------------------------------------------------------- config 5 (fp32)
def test_fp32_sweep_against_fp64(gpu):
    _, s64 = models.synthetic3d(N=48)
    V0 = models.synthetic3d_V0(s64.state_grid)
    J64, u64 = s64.value_iteration(V0, report_time=False)
    idx64 = s64.last_policy_index
    sysd, ref = models.synthetic3d(N=48)
    s32 = DPSolver(sysd, dtype=np.float32)
    s32.discretize_state(0, 1, 48, 0, 1, 48, 0, 1, 48)
    s32.perturb_grid, s32.perturb_proba = ref.perturb_grid, ref.perturb_proba
    s32.control_steps = ref.control_steps
    J32, u32 = s32.value_iteration(V0.astype(np.float32), report_time=False)
    assert J32.dtype == np.float32
    assert np.abs(J32 - J64).max() / np.abs(J64).max() < 1e-5       # BASELINE config 5 bar
    assert (s32.last_policy_index != idx64).mean() < 0.02            # fp32 near-ties only
    assert np.abs(u32[..., 0] - u64[..., 0]).max() <= 2.01 * (2. / 63)


# ---------------------------------------------------------------- generic shapes
def _small_problem(d, nu, with_w):
    dims = (d, nu, 1) if with_w else (d, nu)
    s = SysDescription(dims)
    a = [0.9, 0.8, 0.7, 0.6][:d]

    def dyn(*args):
        x, u = args[:d], args[d:d + nu]
        w = args[d + nu] if with_w else 0.
        return tuple(a[k] * x[k] + 0.3 * u[k % nu] + (w if k == d - 1 else 0. * u[0])
                     for k in range(d))

    def cost(*args):
        x, u = args[:d], args[d:d + nu]
        c = sum((x[k] - 0.2) * (x[k] - 0.2) for k in range(d)) + sum(0.1 * abs(ui) for ui in u)
        if nu > 1:
            c = c + 0.05 * u[0] * u[1]
        return c + 0. * u[0]

    def box(*x):
        return tuple((-1., 1. + 0.25 * c) for c in range(nu))
    # wrap in functions with explicit signatures (the API inspects them)
    names = ['x%d' % i for i in range(d)] + ['u%d' % i for i in range(nu)] + (['w'] if with_w else [])
    ns = {}
    exec('def dyn_f({0}): return _dyn({0})\ndef cost_f({0}): return _cost({0})\n'
         'def box_f({1}): return _box({1})'.format(', '.join(names), ', '.join(names[:d])),
         dict(_dyn=dyn, _cost=cost, _box=box), ns)
    s.dyn, s.cost, s.control_box = ns['dyn_f'], ns['cost_f'], ns['box_f']
    if with_w:
        s.perturb_laws = [models.NormalLaw(0, 0.2)]
    solver = DPSolver(s)
    solver.discretize_state(*sum(([-1., 1., 5 + k] for k in range(d)), []))
    if with_w:
        solver.discretize_perturb(-0.5, 0.5, 5)
    solver.control_steps = tuple(0.4 + 0.1 * c for c in range(nu))
    return solver


@pytest.mark.parametrize('d,nu,with_w', [(1, 1, True), (2, 2, True), (3, 1, False),
                                         (4, 1, True), (2, 3, False), (4, 2, True)])
def test_generic_dimensions_against_numpy_oracle(gpu, d, nu, with_w):
    solver = _small_problem(d, nu, with_w)
    rng = np.random.default_rng(d * 10 + nu)
    V = rng.standard_normal(solver._state_grid_shape)
    J, u = solver.value_iteration(V, report_time=False)
    Jo, uo, io, mo = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    assert np.array_equal(J, Jo)            # same operators, same order: bit-exact
    assert np.array_equal(solver.last_policy_index, io)
    assert np.array_equal(u, uo)


def test_nan_and_inf_costs_follow_numpy_argmin(gpu):
    s = SysDescription((1, 1, 1))

    def dyn(x, u, w):
        return (x + u + w,)

    def cost(x, u, w):
        bad = np.where(u > 0.5, np.nan, np.where(u < -0.5, np.inf, u * u))
        return np.where(x > 0.4, bad, np.where(x < -0.4, np.inf + 0. * u, u * u))
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda x: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.1)]
    solver = DPSolver(s)
    solver.discretize_state(-1, 1, 9)
    solver.discretize_perturb(-0.2, 0.2, 3)
    solver.control_steps = (0.25,)
    V = np.linspace(0, 1, 9)
    with np.errstate(all='ignore'):
        J, u = solver.value_iteration(V, report_time=False)
        Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    assert np.array_equal(J, Jo, equal_nan=True)
    assert np.array_equal(solver.last_policy_index, io)
    assert np.isnan(J[-1]) and np.isinf(J[0])         # first NaN wins; all-inf keeps index 0
    assert io[0] == 0


def test_column_kernel_nan_inf_costs_and_chunk_merge(gpu):
    """NaN / inf costs through the column kernel: few nodes per column, so the
    control lattice is cut into chunks over the waves and the partial minima
    are merged through LDS -- first NaN wins, all-inf keeps index 0, ties keep
    the first index (numpy argmin, stodynprog.py:686)."""
    s = SysDescription((2, 1, 1))

    def dyn(x, y, u, w):
        return (x + 0.5 * u, 0.5 * y + w)

    def cost(x, y, u, w):
        plateau = np.where(np.abs(u) < 0.3, 0. * u, u * u)              # exact ties around u = 0
        bad = np.where(u > 0.9, np.nan, np.where(u < -0.9, np.inf, plateau))
        return np.where(x > 0.4, bad, np.where(x < -0.4, np.inf + 0. * u, plateau)) + 0. * y
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda x, y: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.1)]
    ref = DPSolver(s)
    ref.discretize_state(-1, 1, 9, -1, 1, 5)
    ref.discretize_perturb(-0.2, 0.2, 3)
    ref.control_steps = (2. / 200,)                    # 201 controls, 9 nodes per column
    assert ref._traced().storage_separable
    V = np.add.outer(np.linspace(0, 1, 9), np.linspace(0, 0.5, 5) ** 2)
    with np.errstate(all='ignore'):
        Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(ref), V)
    assert np.isnan(Jo[-1]).all() and np.isinf(Jo[0]).all() and (io[0] == 0).all()
    for kernel, dtype in (('column', np.float64), ('generic', np.float64), ('column', np.float32)):
        sol = _clone_with_kernel(s, ref, kernel, dtype)
        J, u = sol.value_iteration(V, report_time=False)
        assert sol.backend_info['kernel'] == kernel
        if dtype == np.float64:
            assert np.array_equal(J, Jo, equal_nan=True), kernel
            assert np.array_equal(sol.last_policy_index, io), kernel
            assert np.array_equal(u, uo), kernel
        else:
            assert np.array_equal(np.isnan(J), np.isnan(Jo)) and np.array_equal(np.isinf(J), np.isinf(Jo))
            assert (sol.last_policy_index[0] == 0).all()      # (the first NaN moves with float32 rounding of u)


def test_tabulated_mode_for_untraceable_callables(gpu):
    _, ref = models.nas_demo(n_E=11, n_P=9, n_w=5)
    sysd = SysDescription((2, 1, 1), name='untraceable')

    def dyn(E, P_req, P_sto, innov):
        shape = np.shape(P_sto)                     # needs a concrete array
        return ref.sys.dyn(E, P_req, np.asarray(P_sto).reshape(shape), innov)
    sysd.dyn = dyn
    sysd.cost = ref.sys.cost
    sysd.control_box = ref.sys.control_box
    sysd.perturb_laws = ref.sys.perturb_laws
    solver = DPSolver(sysd)
    solver.discretize_state(0, 7.2, 11, -6, 6, 9)
    solver.perturb_grid, solver.perturb_proba = ref.perturb_grid, ref.perturb_proba
    solver.control_steps = (.1,)
    assert isinstance(solver._traced(), TraceError)
    V = np.random.default_rng(0).standard_normal((11, 9))
    J, u = solver.value_iteration(V, report_time=False)
    assert solver.backend_info['mode'] == 'tabulated'
    Jf, uf = ref.value_iteration(V, report_time=False)
    assert ref.backend_info['mode'] == 'traced'
    assert np.array_equal(J, Jf) and np.array_equal(u, uf)
    assert np.array_equal(solver.last_policy_index, ref.last_policy_index)
    (Jr, r), _ = solver.value_iteration((V - V[5, 4], 0.), rel_dp=True, report_time=False)
    assert Jr[5, 4] == 0.0
    # policy evaluation and policy iteration also work in tabulated mode
    E, refs = quiet(solver.eval_policy, uf, 4, True, V, J_ref_full=True)
    Ef, refs_f = quiet(ref.eval_policy, uf, 4, True, V, J_ref_full=True)
    assert np.array_equal(E, Ef) and np.array_equal(refs, refs_f)
    (Jp, rp), polp = quiet(solver.policy_iteration, uf, 3, 1, True)
    (Jq, rq), polq = quiet(ref.policy_iteration, uf, 3, 1, True)
    assert rp == rq and np.array_equal(Jp, Jq) and np.array_equal(polp, polq)


def test_value_at_state_entry_points(gpu):
    g = golden('g7_nas')
    _, solver = models.nas_demo()
    interp = solver.interp_on_state(g['J1'])
    for ind in ((0, 0), (25, 20), (50, 40), (13, 7)):
        x_k = (solver.state_grid[0][ind[0]], solver.state_grid[1][ind[1]])
        J_opt, u_opt = solver._value_at_state_vect(x_k, interp)
        assert abs(J_opt - g['J2'][ind]) <= 1e-12 * max(1, abs(g['J2'][ind]))
        if g['margin2'][ind] > 1e-12:
            assert u_opt[0] == g['pol2'][ind][0]
        J_loop, u_loop = solver._value_at_state_loop(x_k, interp)
        assert J_loop == J_opt and tuple(u_loop) == tuple(u_opt)


def test_bellman_recursion_against_reference(gpu):
    """finite-horizon recursion of a time-dependent system against the
    reference's own bellman_recursion (sdp.py:536-591)"""
    g = golden('g8_bellman')
    _, solver = models.finite_horizon()
    J, pol = quiet(solver.bellman_recursion, 5, g['J_fin'])
    assert J.shape == g['J'].shape and pol.shape == g['pol'].shape
    assert np.abs(J - g['J']).max() <= 1e-13 * np.abs(g['J']).max()
    assert np.array_equal(pol, g['pol'])


def test_bellman_recursion_with_time_indexed_data(gpu, monkeypatch):
    """The reference's finite-horizon examples look their input up by time
    index inside the cost (`P_prod_data[k]`, examples/01 .../pv_storage_control.py:82):
    a symbolic k cannot index an array, so each step is traced with its concrete
    k and the constants of the step are lifted into kernel parameters -- every
    step runs the SAME fused code object, none falls back to host callbacks.
    All 48 steps equal the reference bit for bit (J and policy values)."""
    from stodynprog_amd import _native as nat
    g = golden('g9_pv_storage')
    _, solver = models.pv_storage()
    assert isinstance(solver._traced(), TraceError)           # no symbolic-time trace
    step0, step1 = solver._trace_now(0), solver._trace_now(30)
    assert step0.t_value == 0 and step0.param_index and not step0.time_dep
    assert step0.structure_key() == step1.structure_key()
    assert step0.param_values() != step1.param_values()
    compiled = []
    real_compile = nat.compile_model
    monkeypatch.setattr(nat, 'compile_model', lambda src, **k: compiled.append(src) or real_compile(src, **k))
    J, pol = quiet(solver.bellman_recursion, 48, np.zeros(50))
    assert solver.backend_info['mode'] == 'traced' and solver.backend_info['time_specialized']
    assert solver.backend_info['lifted_constants'] == len(step0.param_index)
    assert len(set(compiled)) == 1, 'all time steps must share one code object'
    assert '__constant__ sdp_real sdp_model_prm' in compiled[0]
    assert np.array_equal(J, g['J'])
    assert np.array_equal(pol, g['pol'])
    # the data array changes (another scenario): no stale constants, no new kernel
    solver.P_prod_data *= 0.5
    J2, _ = quiet(solver.bellman_recursion, 48, np.zeros(50))
    assert len(set(compiled)) == 1 and not np.array_equal(J2, J)
    spec = vi_numpy.Spec.from_solver(solver)
    Jo, _, _, _ = vi_numpy.value_iteration(spec, J2[1], t_k=0)
    assert np.array_equal(J2[0], Jo)


def test_parameter_study_compiles_at_most_two_code_objects(gpu, monkeypatch):
    """The callables are traced at every call (module-level data may change,
    as in the reference where they are plain Python calls).  A loop over a
    cost coefficient keeps one expression structure: the first value is
    compiled with literal constants, the second switches to lifted constants,
    every later value reuses that code object -- and every result equals the
    numpy oracle evaluated with the same coefficient, bit for bit.  (A value
    that coincides with another constant of the model merges two DAG leaves
    and counts as a new structure: one more compile, same results.)"""
    from stodynprog_amd import _native as nat
    coef = {'penalty': 0.1, 'loss': 0.02}
    s = SysDescription((2, 1, 1), name='study')

    def dyn(e, p, u, w):
        return (e + u - coef['loss'] * abs(u), 0.8 * p + w)

    def cost(e, p, u, w):
        return (p - u) * (p - u) + coef['penalty'] * u * u
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda e, p: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.3)]
    solver = DPSolver(s)
    solver.discretize_state(0, 4, 21, -2, 2, 13)
    solver.discretize_perturb(-0.9, 0.9, 5)
    solver.control_steps = (0.05,)
    compiled = []
    real_compile = nat.compile_model
    monkeypatch.setattr(nat, 'compile_model', lambda src, **k: compiled.append(src) or real_compile(src, **k))
    V = np.random.default_rng(4).standard_normal((21, 13))
    results = []
    for penalty, loss in ((0.1, 0.02), (0.2, 0.02), (0.4, 0.05), (0.75, 0.01), (0.1, 0.02)):
        coef['penalty'], coef['loss'] = penalty, loss
        J, pol = solver.value_iteration(V, report_time=False)
        Jo, polo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
        assert np.array_equal(J, Jo) and np.array_equal(pol, polo), (penalty, loss)
        assert np.array_equal(solver.last_policy_index, io)
        results.append(J)
    assert len(set(compiled)) == 2, len(set(compiled))
    assert 'sdp_model_prm' not in compiled[0] and 'sdp_model_prm' in compiled[-1]
    assert solver.backend_info['lifted_constants'] > 0 and solver.backend_info['kernel'] == 'column'
    assert np.array_equal(results[0], results[-1]) and not np.array_equal(results[0], results[1])


def test_control_box_reading_changed_data_is_not_served_stale(gpu):
    """the control-box table is cached; a rating read from module-level data
    that changes between two calls must be noticed (probe of the callback at
    every call) -- the reference evaluates control_box at every sweep"""
    rating = {'P': 1.0, 'E': 4.0}
    s = SysDescription((2, 1, 1), name='ratings')
    s.dyn = lambda e, p, u, w: (e + u, 0.8 * p + w)
    s.cost = lambda e, p, u, w: (p - u) * (p - u)

    def box(e, p):
        return ((np.max((-e, -rating['P'])), np.min((rating['E'] - e, rating['P']))),)
    s.control_box = box
    s.perturb_laws = [models.NormalLaw(0, 0.3)]
    solver = DPSolver(s)
    solver.discretize_state(0, 4, 21, -2, 2, 13)
    solver.discretize_perturb(-0.9, 0.9, 5)
    solver.control_steps = (0.05,)
    V = np.random.default_rng(4).standard_normal((21, 13))
    seen = []
    for P in (1.0, 0.5, 0.5, 2.0):
        rating['P'] = P
        J, pol = solver.value_iteration(V, report_time=False)
        Jo, polo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
        assert np.array_equal(J, Jo) and np.array_equal(pol, polo), P
        assert np.abs(pol).max() <= P
        seen.append(J)
    assert not np.array_equal(seen[0], seen[1]) and np.array_equal(seen[1], seen[2])


def test_bellman_recursion_time_dependent(gpu):
    s = SysDescription((1, 1, 1), stationnary=False)

    def dyn(k, x, u, w):
        return (0.9 * x + u + w + 0.05 * k,)

    def cost(k, x, u, w):
        return (x - 0.1 * k) ** 2 + 0.1 * u * u

    def box(k, x):
        return ((-1., 1. + 0.5 * k),)
    s.dyn, s.cost, s.control_box = dyn, cost, box
    s.perturb_laws = [models.NormalLaw(0, 0.1)]
    solver = DPSolver(s)
    solver.discretize_state(-2, 2, 17)
    solver.discretize_perturb(-0.3, 0.3, 5)
    solver.control_steps = (0.125,)
    J_fin = solver.state_grid[0] ** 2
    J, pol = quiet(solver.bellman_recursion, 4, J_fin)
    assert J.shape == (4, 17) and pol.shape == (4, 17, 1)
    spec = vi_numpy.Spec.from_solver(solver)
    nxt = J_fin
    for t in (3, 2, 1, 0):
        Jo, po, _, _ = vi_numpy.value_iteration(spec, nxt, t_k=t)
        assert np.array_equal(J[t], Jo) and np.array_equal(pol[t], po)
        nxt = Jo


# ---------------------------------------------------------------- next rows (8f)
def test_eval_policy_against_reference(gpu):
    g = golden('g6_policy')
    _, solver = models.storage_ar1()
    pol = models.storage_ar1_empirical_policy(solver)
    J, J_ref = quiet(solver.eval_policy, pol, 50, rel_dp=True, J_ref_full=True)
    assert J_ref.shape == (50,)
    assert np.allclose(J_ref, g['ar1_J_ref'], rtol=1e-12, atol=1e-15)
    assert np.abs(J - g['ar1_J']).max() < 1e-12
    assert '{:g}'.format(J_ref[-1]) == '0.105724'            # AR1.ipynb:594
    J7 = quiet(solver.eval_policy, pol, 7)
    assert np.abs(J7 - g['ar1_J7']).max() < 1e-12
    Jl, last = quiet(solver.eval_policy, pol, 50, rel_dp=True)
    assert last == J_ref[-1] and np.array_equal(Jl, J)
    # bit-exact against the sequential-order oracle
    Jo, Jo_ref = vi_numpy.eval_policy(vi_numpy.Spec.from_solver(solver), pol, 50, True,
                                      J_ref_full=True)
    assert np.array_equal(J, Jo) and np.array_equal(J_ref, Jo_ref)
    _, sea = models.searev()
    Js, Js_ref = quiet(sea.eval_policy, models.searev_linear_policy(sea), 20, True,
                       J_ref_full=True)
    assert np.allclose(Js_ref, g['searev_J_ref'], rtol=1e-12, atol=1e-16)
    assert np.abs(Js - g['searev_J']).max() < 1e-12


def test_policy_iteration_reproduces_published_costs(gpu, capsys):
    """AR1.ipynb:594-602: ref policy cost 0.105724 -> 0.0486519 -> 0.0468464"""
    g = golden('g6_policy')
    _, solver = models.storage_ar1()
    pol_ini = models.storage_ar1_empirical_policy(solver)
    (J, J_ref), pol = solver.policy_iteration(pol_ini, 50, 2, rel_dp=True)
    out = capsys.readouterr().out.replace('\r', '\n')
    costs = [l.split(':')[1].strip() for l in out.split('\n') if l.startswith('ref policy cost')]
    assert costs == ['0.105724', '0.0486519', '0.0468464']
    assert 'policy iteration 1/2' in out and 'policy iteration 2/2' in out
    assert abs(J_ref - float(g['ar1_pi_Jref'])) < 1e-12
    assert np.abs(J - g['ar1_pi_J']).max() < 1e-10
    pol_ref = pol.copy()                          # the fixture holds control 0 (the second is pinned to one point)
    pol_ref[..., 0] = g['ar1_pi_pol0']
    _prove_policy_ties(solver, pol_ini, 50, 2, pol, pol_ref, 'ar1 policy iteration')


def _prove_policy_ties(solver, pol_ini, n_val, n_pol, pol, pol_ref, what):
    """The reference's final policy differs from ours at a few nodes (np.inner's BLAS
    summation order vs sequential; dead-band / flat costs give exact ties).  Proof
    node by node, no fraction threshold: the cost-to-go that the LAST improvement
    step consumed is rebuilt (same calls as policy_iteration, stodynprog.py:777-812),
    and at every differing node the CPU oracle's per-control expected costs of the
    two control values must agree within 1e-12 * max(1, |J|), both at the minimum."""
    from conftest import TIE_RTOL, _report
    from oracle import vi_numpy
    diff = np.flatnonzero((pol != pol_ref).any(axis=-1).ravel())
    n_nodes = pol[..., 0].size
    if diff.size == 0:
        _report('{:32s} nodes {:9d}  policy entries identical'.format(what, n_nodes))
        return
    p = pol_ini
    Jp = quiet(solver.eval_policy, p, n_val, True)
    for k in range(n_pol - 1):
        _, p = quiet(solver.value_iteration, Jp, True)
        Jp = quiet(solver.eval_policy, p, n_val, True)
    J_last = Jp[0]                       # differential cost fed to the last value_iteration
    spec = vi_numpy.Spec.from_solver(solver)
    interp = vi_numpy.Interp(*spec.state_grid)
    interp.set_values(np.asarray(J_last, dtype=float))
    worst = 0.0
    nu = pol.shape[-1]
    for flat in diff:
        ind = np.unravel_index(int(flat), spec.shape)
        x_k = tuple(g_[i] for g_, i in zip(spec.state_grid, ind))
        J_opt, u_opt, flat_opt, margin, Jfull = vi_numpy.backup_node(spec, x_k, interp, None, full=True)
        u_grids, dims = vi_numpy.control_grids(spec, x_k)
        costs = np.asarray(Jfull, dtype=float)
        pick = []
        for values in (pol[ind], pol_ref[ind]):
            sub = tuple(int(np.argmin(np.abs(u_grids[c] - values[c]))) for c in range(nu))
            assert all(u_grids[c][sub[c]] == values[c] for c in range(nu)), (what, ind, values)
            pick.append(costs[sub])
        tol = TIE_RTOL * max(1.0, abs(float(J_opt)))
        assert abs(pick[0] - pick[1]) <= tol and max(pick) - costs.min() <= tol, \
            '{}: node {} controls {} / {} cost {!r} / {!r}: not a tie'.format(
                what, ind, pol[ind], pol_ref[ind], pick[0], pick[1])
        worst = max(worst, abs(pick[0] - pick[1]))
    _report('{:32s} nodes {:9d}  policy entries differing {:5d}  (all proved ties, largest cost gap '
            '{:.2e})'.format(what, n_nodes, diff.size, worst))


# ---------------------------------------------------------------- kernel families
def _clone_with_kernel(sysd, ref, kernel, dtype=np.float64):
    s = DPSolver(sysd, dtype=dtype)
    s.state_grid, s._state_grid_shape = ref.state_grid, ref._state_grid_shape
    s._state_ref_ind, s._state_ref = ref._state_ref_ind, ref._state_ref
    s.perturb_grid, s.perturb_proba = ref.perturb_grid, ref.perturb_proba
    s.control_steps = ref.control_steps
    s.kernel = kernel
    return s


@pytest.mark.parametrize('name,kw', [
    ('nas_demo', {}),
    ('storage_ar1', dict(n_E=33, n_P=20, steps=(0.05, 0.1))),
    ('searev', dict(n_E=17, n_S=12, n_A=9, step=0.01)),
    ('synthetic3d', dict(N=24)),
])
def test_column_and_generic_kernels_agree_bitwise(gpu, name, kw):
    """the LDS-table column kernel only removes repeated work: same bits as the
    generic per-cell kernel for J, index, control values, relative DP and
    policy evaluation"""
    sysd, ref = getattr(models, name)(**kw)
    assert ref._traced().storage_separable
    col = _clone_with_kernel(sysd, ref, 'column')
    gen = _clone_with_kernel(sysd, ref, 'generic')
    V = np.random.default_rng(11).standard_normal(ref._state_grid_shape)
    Jc, uc = col.value_iteration(V, report_time=False)
    Jg, ug = gen.value_iteration(V, report_time=False)
    assert col.backend_info['kernel'] == 'column' and gen.backend_info['kernel'] == 'generic'
    assert np.array_equal(Jc, Jg) and np.array_equal(uc, ug)
    assert np.array_equal(col.last_policy_index, gen.last_policy_index)
    Vd = V - V[ref._state_ref_ind]
    (Jc2, rc), _ = col.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    (Jg2, rg), _ = gen.value_iteration((Vd, 0.), rel_dp=True, report_time=False)
    assert rc == rg and np.array_equal(Jc2, Jg2)
    Ec, refs_c = quiet(col.eval_policy, ug, 6, True, V * 0.1, J_ref_full=True)
    Eg, refs_g = quiet(gen.eval_policy, ug, 6, True, V * 0.1, J_ref_full=True)
    assert np.array_equal(Ec, Eg) and np.array_equal(refs_c, refs_g)


def _priced_storage_problem(n_w):
    """stock + exogenous price; the cost depends on the perturbation, x0' does not"""
    s = SysDescription((2, 1, 1), name='priced storage')

    def dyn(e, p, u, w):
        return (e + 0.25 * u, 0.7 * p + w)

    def cost(e, p, u, w):
        return (p + 0.5 * w) * u + 0.05 * u * u + 0.01 * (e - 1.0) * (e - 1.0)

    def box(e, p):
        return ((-1., 1.),)
    s.dyn, s.cost, s.control_box = dyn, cost, box
    s.perturb_laws = [models.NormalLaw(0, 0.3)]
    solver = DPSolver(s)
    solver.discretize_state(0, 2, 37, -1, 1, 11)
    solver.discretize_perturb(-0.9, 0.9, n_w)
    solver.control_steps = (0.125,)
    return s, solver


@pytest.mark.parametrize('case', ['storage_ar1', 'searev', 'synthetic3d', 'priced-8', 'priced-7', 'priced-1'])
def test_float32_pair_table_matches_generic_kernel_bitwise(gpu, case):
    """float32 column kernels keep perturbation points 2k, 2k+1 side by side in
    the LDS table (SDP_COL_WPAIR: 8-byte reads, packed arithmetic).  Same bits
    as the per-cell generic kernel in float32 -- even and odd W (tail point),
    cost with and without w, eval_policy; float32 within the north star's 1e-5 of float64."""
    if case.startswith('priced'):
        sysd, ref = _priced_storage_problem(int(case.split('-')[1]))
    else:
        kw = {'storage_ar1': dict(n_E=33, n_P=20, steps=(0.05, 0.1)),
              'searev': dict(n_E=17, n_S=12, n_A=9, step=0.01),
              'synthetic3d': dict(N=24)}[case]
        sysd, ref = getattr(models, case)(**kw)
    from stodynprog_amd import codegen
    assert codegen.use_wpair(ref._traced(), np.float32) and not codegen.use_wpair(ref._traced(), np.float64)
    col = _clone_with_kernel(sysd, ref, 'column', np.float32)
    gen = _clone_with_kernel(sysd, ref, 'generic', np.float32)
    V = np.random.default_rng(5).standard_normal(ref._state_grid_shape).astype(np.float32)
    Jc, uc = col.value_iteration(V, report_time=False)
    Jg, ug = gen.value_iteration(V, report_time=False)
    assert col.backend_info['kernel'] == 'column' and gen.backend_info['kernel'] == 'generic'
    assert '#define SDP_COL_WPAIR 1' in col._kernel_plan()['source']
    assert Jc.dtype == np.float32
    assert np.array_equal(Jc, Jg) and np.array_equal(uc, ug)
    assert np.array_equal(col.last_policy_index, gen.last_policy_index)
    Ec = quiet(col.eval_policy, ug, 5, False, V * np.float32(0.1))
    Eg = quiet(gen.eval_policy, ug, 5, False, V * np.float32(0.1))
    assert np.array_equal(Ec, Eg)
    # and float32 stays within 1e-5 of the float64 sweep of the same problem
    f64 = _clone_with_kernel(sysd, ref, 'column')
    Jd, _ = f64.value_iteration(V.astype(float), report_time=False)
    assert np.abs(Jc - Jd).max() / np.abs(Jd).max() < 1e-5


def test_column_kernel_is_the_default_for_storage_problems(gpu):
    for name in ('nas_demo', 'storage_ar1', 'searev', 'synthetic3d'):
        _, s = getattr(models, name)()
        plan = s._kernel_plan()
        assert plan['column'], name
    _, inv = models.inventory()
    assert not inv._kernel_plan()['column']          # x' = x + u - w is not separable
    # a table that does not fit the 160 KiB LDS: the column kernel tabulates a window of
    # rows per segment of the column (tests/test_gpu_window.py) -- no cliff
    _, big = models.synthetic3d(N=20)
    big.state_grid[0] = np.linspace(0, 1, 1000)
    big._state_grid_shape = (1000, 20, 20)
    big.kernel = 'column'
    plan = big._kernel_plan()
    assert plan['column'] and plan['window'] and plan['window'][2] < 1000
    big.kernel = 'auto'                              # (8-byte reals: the reduced-array sweep, one controlled axis)
    assert big._kernel_plan()['lead_axes'] == 1
    # unless the controls of one node span more rows than a window can hold: then the
    # LDS-staged tile kernel runs, and forcing 'column' says why it cannot
    _, wide = models.synthetic3d(N=20)
    wide.state_grid[0] = np.linspace(0, 0.01, 1000)    # the controls cross this axis end to end
    wide._state_grid_shape = (1000, 20, 20)
    plan = wide._kernel_plan()
    assert not plan['column'] and plan['lead_axes'] == 1
    wide.certified_filter = False                      # (without the filter there is no reduced-array sweep)
    plan = wide._kernel_plan()
    assert not plan['column'] and plan['staged']
    wide.kernel = 'column'
    with pytest.raises(ValueError):
        wide._kernel_plan()


# ---------------------------------------------------------------- RCCL plumbing
def test_rccl_communicator_single_rank(gpu):
    """librccl loads, a 1-rank communicator initialises, attaches to a problem
    handle, and the device collectives used by bench.py work (the N > 1 data
    path is the same code with more ranks; its slab logic is covered on CPU in
    test_dist_cpu.py)."""
    from stodynprog_amd import dist
    uid = dist.RcclCommunicator.new_unique_id()
    assert len(uid) == 128
    comm = dist.RcclCommunicator(0, 1, uid)
    try:
        assert comm.allreduce_max(3.5) == 3.5
        comm.barrier()
        sysd, ref = models.synthetic3d(N=20)
        s = DPSolver(sysd, comm=comm)
        s.discretize_state(0, 1, 20, 0, 1, 20, 0, 1, 20)
        s.perturb_grid, s.perturb_proba = ref.perturb_grid, ref.perturb_proba
        s.control_steps = ref.control_steps
        V0 = models.synthetic3d_V0(s.state_grid)
        J, u = s.value_iteration(V0, report_time=False)
        Jr, ur = ref.value_iteration(V0, report_time=False)
        assert np.array_equal(J, Jr) and np.array_equal(u, ur)
        (Jd, r), _ = s.value_iteration((J - J[10, 10, 10], 0.), rel_dp=True, report_time=False)
        assert Jd[10, 10, 10] == 0.0
        # the communicator path runs every backup phase by phase (4 launches here)
        prob = s._problem()
        assert prob.parts is not None and prob.parts.shape == (4, 2)
        assert prob.parts[0, 0] == 0 and prob.parts[-1, -1] == 8000
        E, refs = quiet(s.eval_policy, u, 5, True, J * 0.5, J_ref_full=True)
        Er, refs_r = quiet(ref.eval_policy, u, 5, True, J * 0.5, J_ref_full=True)
        assert np.array_equal(E, Er) and np.array_equal(refs, refs_r)
    finally:
        s._cache.clear()
        comm.close()


def test_column_kernel_partial_ranges_match_full_sweep(gpu):
    """what each rank computes in the multi-GPU sweep: the column kernel over
    sub-ranges of columns (the parts of dist.phase_partition) gives the same
    bits as one launch over everything"""
    from stodynprog_amd import dist, _native as nat
    from stodynprog_amd.solver import _DeviceProblem
    _, solver = models.synthetic3d(N=24)
    V = np.random.default_rng(5).standard_normal((24, 24, 24))
    J, u = solver.value_iteration(V, report_time=False)
    full = solver._problem()
    assert full.layout == nat.LAYOUT_COLUMNS
    Jd = full._to_device_order(J).ravel()
    parts = dist.phase_partition(24 * 24, 24, 3, 4)        # 3 ranks, 4 phases
    merged = np.full(Jd.shape, np.nan)
    for rank in range(3):
        for ph in range(parts.shape[0]):
            lo, hi = int(parts[ph, rank]), int(parts[ph, rank + 1])
            if hi == lo:
                continue
            sub = _DeviceProblem(full._keep, solver.backend_info['module'], np.float64,
                                 (24, 24, 24), 1, 32, solver.backend_info['lanes_per_node'],
                                 False, (lo, hi), layout=nat.LAYOUT_COLUMNS)
            sub.set_value(V)
            sub.sweep()
            merged[lo:hi] = sub._to_device_order(sub.get_value()).ravel()[lo:hi]
            sub.close()
    assert np.array_equal(merged, Jd)


# ---------------------------------------------------------------- end-to-end known answers
def test_searev_policy_iteration_reproduces_the_committed_policy(gpu, capsys):
    """The reference example ships the optimal storage policy computed by
    policy_iteration(pol_lin, n_val=1000, n_pol=5, rel_dp=True)
    (examples/20 Searev storage control/storage_control.py:137-160; 773 s on one
    CPU core).  Same call here: 6000 policy evaluations + 5 sweeps of
    31x61x61 nodes x up to 2201 controls x 9 perturbations."""
    g = golden('g4_searev')
    committed = g['committed_policy']
    assert committed.shape == (31, 61, 61, 1)
    _, solver = models.searev()
    pol_lin = models.searev_linear_policy(solver)
    (J, J_ref), pol = solver.policy_iteration(pol_lin, 1000, 5, rel_dp=True)
    out = capsys.readouterr().out.replace('\r', '\n')
    costs = [float(l.split(':')[1]) for l in out.split('\n') if l.startswith('ref policy cost')]
    assert len(costs) == 6
    assert '{:g}'.format(costs[-1]) == '0.0746743'
    # entries that differ from the committed array are proved ties node by node
    _prove_policy_ties(solver, pol_lin, 1000, 5, pol, committed, 'searev committed policy')


# ---------------------------------------------------------------- config 5
def test_fp32_512cubed_against_fp64_oracle(gpu):
    """BASELINE config 5: 512^3 fp32 sweep with policy-index extraction,
    tolerance 1e-5 against the fp64 oracle on the same grid (sampled nodes)."""
    sysd, ref = models.synthetic3d(N=512)
    s32 = DPSolver(sysd, dtype=np.float32)
    s32.state_grid, s32._state_grid_shape = ref.state_grid, ref._state_grid_shape
    s32._state_ref_ind = ref._state_ref_ind
    s32.perturb_grid, s32.perturb_proba = ref.perturb_grid, ref.perturb_proba
    s32.control_steps = ref.control_steps
    V0 = models.synthetic3d_V0(ref.state_grid)                # fp64, 1.07 GB
    J32, u32 = s32.value_iteration(V0.astype(np.float32), report_time=False)
    assert s32.backend_info['kernel'] == 'column' and J32.dtype == np.float32
    idx32 = s32.last_policy_index
    nodes = np.random.default_rng(9).integers(0, V0.size, 20000)
    Jo, io, mo = c_oracle.vi_synth3d(ref.state_grid, V0, models.SYNTH_PAR, -1., 1., 64,
                                     ref.perturb_grid[0], ref.perturb_proba[0],
                                     node_ids=nodes, n_threads=8)
    rel = np.abs(J32.ravel()[nodes] - Jo).max() / np.abs(Jo).max()
    assert rel < 1e-5, rel
    # indices: exact wherever the fp64 margin is above fp32 resolution
    clear = mo > 1e-5 * np.maximum(1.0, np.abs(Jo))
    assert (idx32.ravel()[nodes][clear] == io[clear]).all()
    assert clear.mean() > 0.5
    assert idx32.min() >= 0 and idx32.max() <= 63
    # ... and against the REFERENCE itself on this grid (golden g11: fp64 stodynprog on 3003
    # sampled nodes of the 512^3 problem, tests/golden/make_golden.py g11)
    g = golden('g11_synth512')
    import zlib
    assert int(g['V0_crc']) == zlib.crc32(V0.tobytes())
    gn = g['nodes']
    rel = np.abs(J32.ravel()[gn] - g['J']).max() / np.abs(g['J']).max()
    assert rel < 1e-5, rel
    clear = g['margin'] > 1e-5 * np.maximum(1.0, np.abs(g['J']))
    assert (idx32.ravel()[gn][clear] == g['idx'][clear]).all() and clear.mean() > 0.5
    # (the fp32 sweep generates its control points in fp32: i*step+lo rounds differently)
    assert np.allclose(u32.reshape(-1)[gn][clear], g['pol'][clear, 0], rtol=1e-6, atol=1e-7)
    from conftest import _report
    _report('{:32s} nodes {:9d}  max|dJ|/|J| {:.2e}  (fp32 sweep vs fp64 reference, bar 1e-5); indices '
            'identical at the {} nodes whose reference margin exceeds fp32 resolution'.format(
                'synthetic c5 512^3 fp32', len(gn), rel, int(clear.sum())))
    # the fp64 sweep on the same grid: bit-level parity with the reference golden
    J64, u64 = ref.value_iteration(V0, report_time=False)
    assert_sweep_parity(J64.ravel()[gn], ref.last_policy_index.ravel()[gn], g['J'], g['idx'],
                        g['margin'], 'synthetic 512^3 fp64 vs reference', prove=(ref, V0), nodes=gn)


def _reservoir_problem():
    """controlled stock driven by the control AND the noise, next to an
    exogenous AR(1) axis: x0' depends on w, so the column kernel locates the
    axis-0 cell per lattice cell (SDP_LEAD_HAS_W)"""
    s = SysDescription((2, 1, 1), name='reservoir')

    def dyn(x, y, u, w):
        return (x + u - 0.5 * w - 0.1 * y, 0.8 * y + w)

    def cost(x, y, u, w):
        return (x - 0.3) * (x - 0.3) + 0.1 * u * u + 0.05 * w * u

    def box(x, y):
        return ((-1., 1.),)
    s.dyn, s.cost, s.control_box = dyn, cost, box
    s.perturb_laws = [models.NormalLaw(0, 0.2)]
    solver = DPSolver(s)
    solver.discretize_state(-1, 1, 33, -1, 1, 17)
    solver.discretize_perturb(-0.5, 0.5, 7)
    solver.control_steps = (0.1,)
    return s, solver


def test_column_kernel_with_noise_driven_stock(gpu):
    sysd, ref = _reservoir_problem()
    m = ref._traced()
    assert m.storage_separable and m.lead_depends_on_w and m.cost_depends_on_w
    col = _clone_with_kernel(sysd, ref, 'column')
    gen = _clone_with_kernel(sysd, ref, 'generic')
    V = np.random.default_rng(21).standard_normal((33, 17))
    Jc, uc = col.value_iteration(V, report_time=False)
    Jg, ug = gen.value_iteration(V, report_time=False)
    assert col.backend_info['kernel'] == 'column'
    # (since round 3 filtered on the shifted lattice: x0' = (x + u) - 0.5 w - 0.1 y is a chain of final sums)
    assert col.backend_info['filter_form'] == 'shifted lattice'
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(ref), V)
    assert np.array_equal(Jc, Jo) and np.array_equal(Jg, Jo)
    assert np.array_equal(col.last_policy_index, io) and np.array_equal(gen.last_policy_index, io)
    assert np.array_equal(uc, uo)
    Ec = quiet(col.eval_policy, uc, 4, False, V)
    Eg = quiet(gen.eval_policy, uc, 4, False, V)
    assert np.array_equal(Ec, Eg)


# ---------------------------------------------------------------- user-level scripts
def _load_example(name):
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', name)
    spec = importlib.util.spec_from_file_location(name[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_example_inventory_script(gpu):
    J, policies = quiet(_load_example('inventory.py').main, 6, True)
    assert np.array_equal(policies[1], [4, 3, 2, 1, 0, 0, 0, 0, 0, 0])       # example_inventory.rst:231
    assert np.array_equal(policies[3], [5, 4, 3, 2, 1, 0, 0, 0, 0, 0])       # :239
    assert np.allclose(J, golden('g2_inventory')['J'][5], rtol=0, atol=1e-12)


def test_example_searev_policy_lookup_and_simulation(gpu):
    out = quiet(_load_example('searev_storage.py').main, 100, 1, 1500, (16, 21, 21), True)
    E, P_prod, P_grid = out['E'], out['P_prod'], out['P_grid']
    assert np.isfinite(E).all() and E.min() > -1e-9 and E.max() < 10 + 1e-9   # storage stays in its box
    assert P_grid.std() < P_prod.std()          # the policy smooths the power sent to the grid
    assert out['pol'].shape == (16, 21, 21, 1) and 0 < out['J_ref'] < 1


def test_example_two_reservoirs(gpu):
    out = quiet(_load_example('two_reservoirs.py').main, 20, 18, 10, 40, 300, True)
    assert out['solver'].backend_info['kernel'] == 'lead' and out['solver'].backend_info['controlled_axes'] == 2
    assert out['pol'].shape == (20, 18, 10, 2) and np.isfinite(out['J']).all() and 0 <= out['J_ref'] < 1
    assert out['levels'][:, :2].min() > -0.3 and out['levels'][:, :2].max() < 2.5      # the policy keeps the levels in range
    assert abs(out['output'][50:].mean() - 0.8) < 0.25                             # and the turbine near its set point


def test_example_pv_storage_finite_horizon(gpu):
    J, pol, E, P_sto = quiet(_load_example('pv_storage.py').main, 48, 50)
    g = golden('g9_pv_storage')
    assert np.array_equal(J, g['J']) and np.array_equal(pol, g['pol'])
    assert E.min() > -1e-9 and E.max() < 2 + 1e-9 and np.abs(P_sto).max() <= 1.0


def test_models_with_lookup_tables_np_interp(gpu):
    """np.interp inside dyn / cost (an efficiency curve, a tariff) runs on the
    fused kernels: results equal the numpy oracle, which calls np.interp itself,
    bit for bit -- column and generic kernels, points on, between and outside the
    table abscissae"""
    eff_x = np.array([-1., -0.5, 0., 0.25, 1.])
    eff_y = np.array([0.80, 0.92, 1.0, 0.95, 0.85])
    s = SysDescription((2, 1, 1), name='lookup tables')

    def dyn(e, p, u, w):
        return (e + 0.5 * u * np.interp(u, eff_x, eff_y), 0.7 * p + w)

    def cost(e, p, u, w):
        tariff = np.interp(p, [-0.5, 0., 0.5], [0.2, 1.0, 3.0], left=0.1, right=5.0)
        return tariff * (p - u) * (p - u) + np.interp(e, [0.5, 3.5], [1., 0.], left=2., right=2.) + 0. * w
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda e, p: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.3)]
    ref = DPSolver(s)
    ref.discretize_state(0, 4, 21, -1, 1, 13)
    ref.discretize_perturb(-0.9, 0.9, 5)
    ref.control_steps = (0.125,)          # hits the abscissae -1, -0.5, 0, 0.25, 1 exactly
    model = ref._traced()
    assert not isinstance(model, TraceError) and model.bit_exact and model.storage_separable
    V = np.random.default_rng(23).standard_normal((21, 13))
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(ref), V)
    for kernel in ('column', 'generic'):
        sol = _clone_with_kernel(s, ref, kernel)
        J, u = sol.value_iteration(V, report_time=False)
        assert sol.backend_info['mode'] == 'traced' and sol.backend_info['kernel'] == kernel
        assert np.array_equal(J, Jo) and np.array_equal(u, uo), kernel
        assert np.array_equal(sol.last_policy_index, io)
    f32 = _clone_with_kernel(s, ref, 'column', np.float32)
    Jf, _ = f32.value_iteration(V, report_time=False)
    assert np.abs(Jf - Jo).max() / np.abs(Jo).max() < 1e-5


def test_column_kernel_four_state_axes(gpu):
    """d = 4 (the largest dimension the reference's interpolation dispatches,
    multilinear_cython.pyx:211-300): a stock next to a 3-axis exogenous process;
    the column kernel tabulates a trilinear partial interpolation per (w, row)"""
    s = SysDescription((4, 1, 1), name='four axes')

    def dyn(e, a, b, c, u, w):
        return (e + 0.5 * u - 0.02 * abs(u),
                0.7 * a + 0.2 * b + w,
                0.6 * b - 0.3 * c + 0.5 * w,
                0.5 * c + 0.1 * a - 0.25 * w)

    def cost(e, a, b, c, u, w):
        return (a + 0.5 * b - u) * (a + 0.5 * b - u) + 0.1 * c * u + 0.05 * (e - 1.0) * (e - 1.0)
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda e, a, b, c: ((np.max((-e / 0.5, -1.)), np.min(((2. - e) / 0.5, 1.))),)
    s.perturb_laws = [models.NormalLaw(0, 0.2)]
    ref = DPSolver(s)
    ref.discretize_state(0, 2, 11, -1, 1, 7, -1, 1, 6, -1, 1, 5)
    ref.discretize_perturb(-0.6, 0.6, 6)
    ref.control_steps = (0.1,)
    assert ref._traced().storage_separable
    V = np.random.default_rng(17).standard_normal((11, 7, 6, 5))
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(ref), V)
    for kernel in ('column', 'generic'):
        sol = _clone_with_kernel(s, ref, kernel)
        J, u = sol.value_iteration(V, report_time=False)
        assert sol.backend_info['kernel'] == kernel
        assert np.array_equal(J, Jo) and np.array_equal(u, uo), kernel
        assert np.array_equal(sol.last_policy_index, io)
    E1 = quiet(_clone_with_kernel(s, ref, 'column').eval_policy, uo, 3, True, V, J_ref_full=True)
    E2 = quiet(_clone_with_kernel(s, ref, 'generic').eval_policy, uo, 3, True, V, J_ref_full=True)
    assert np.array_equal(E1[0], E2[0]) and np.array_equal(E1[1], E2[1])
    f32 = _clone_with_kernel(s, ref, 'column', np.float32)
    Jf, _ = f32.value_iteration(V, report_time=False)
    assert np.abs(Jf - Jo).max() / np.abs(Jo).max() < 1e-5


def test_column_kernel_deterministic_and_two_controls(gpu):
    """column kernel corner cases: no perturbation (W = 0, dims of length 2),
    two multi-point controls (Cartesian lattice, control 0 slowest), 4-D state"""
    # deterministic, 2-D
    s = SysDescription((2, 1), name='det')

    def dyn(x, y, u):
        return (x + 0.3 * u, 0.9 * y + 0.05)

    def cost(x, y, u):
        return (x - y) * (x - y) + 0.1 * abs(u)
    s.dyn, s.cost = dyn, cost
    s.control_box = lambda x, y: ((-1., 1.),)
    det = DPSolver(s)
    det.discretize_state(-1, 1, 21, 0, 1, 9)
    det.control_steps = (0.25,)
    assert det._kernel_plan()['column'] and det._kernel_plan()['W'] == 0
    V = np.random.default_rng(1).standard_normal((21, 9))
    J, u = det.value_iteration(V, report_time=False)
    assert det.backend_info['kernel'] == 'column'
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(det), V)
    assert np.array_equal(J, Jo) and np.array_equal(u, uo)
    assert np.array_equal(det.last_policy_index, io)

    # two real controls + 4-D state, stochastic
    s2 = SysDescription((4, 2, 1), name='two controls')

    def dyn2(a, b, c, d, u, v, w):
        return (a + 0.2 * u - 0.1 * v, 0.9 * b + w, 0.8 * c - 0.1 * b + 0.5 * w, 0.7 * d + 0.2 * c)

    def cost2(a, b, c, d, u, v, w):
        return (a - 0.1) * (a - 0.1) + 0.2 * u * u + 0.3 * abs(v) + 0.05 * u * v + 0.01 * b * w

    def box2(a, b, c, d):
        return ((-1., 1.), (0., 0.5 + 0.5 * (a > 0)))
    s2.dyn, s2.cost, s2.control_box = dyn2, cost2, box2
    s2.perturb_laws = [models.NormalLaw(0, 0.1)]
    two = DPSolver(s2)
    two.discretize_state(-1, 1, 9, -1, 1, 5, -1, 1, 4, -1, 1, 3)
    two.discretize_perturb(-0.3, 0.3, 5)
    two.control_steps = (0.25, 0.2)
    plan = two._kernel_plan()
    assert plan['column'] and plan['per_node'] and plan['max_u'] == 9 * 6
    V = np.random.default_rng(3).standard_normal((9, 5, 4, 3))
    J, u = two.value_iteration(V, report_time=False)
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(two), V)
    assert np.array_equal(J, Jo) and np.array_equal(u, uo)
    assert np.array_equal(two.last_policy_index, io)
    E = quiet(two.eval_policy, u, 3, False, V)
    Eo = vi_numpy.eval_policy(vi_numpy.Spec.from_solver(two), uo, 3, False, V)
    assert np.array_equal(E, Eo)


def test_system_parameters_are_forwarded(gpu):
    """sys.params reach the callables as keyword arguments (sdp.py:70-73,
    439-440, 669-676) and are baked into the compiled model"""
    s = SysDescription((2, 1, 1), params={'gain': 0.3, 'target': 0.2})

    def dyn(x, y, u, w, **p):
        return (x + p['gain'] * u, 0.9 * y + w)

    def cost(x, y, u, w, **p):
        return (x - p['target']) ** 2 + 0.1 * u * u + 0.01 * y * w

    def box(x, y, **p):
        return ((-1., 1.),)
    s.dyn, s.cost, s.control_box = dyn, cost, box
    s.perturb_laws = [models.NormalLaw(0, 0.1)]
    solver = DPSolver(s)
    solver.discretize_state(-1, 1, 17, -1, 1, 9)
    solver.discretize_perturb(-0.3, 0.3, 5)
    solver.control_steps = (0.2,)
    V = np.random.default_rng(4).standard_normal((17, 9))
    J, u = solver.value_iteration(V, report_time=False)
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    assert np.array_equal(J, Jo) and np.array_equal(u, uo)
    # changing a parameter changes the model (and its code object)
    s.params['gain'] = 0.6
    J2, _ = solver.value_iteration(V, report_time=False)
    Jo2, _, _, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    assert np.array_equal(J2, Jo2) and not np.array_equal(J2, J)


# ---------------------------------------------------------------- randomised models
def _random_expr(rng, leaves, depth):
    """random numpy expression over the given leaf names, built from operators
    whose device results are correctly rounded (bit-exact against numpy)"""
    if depth == 0 or rng.random() < 0.2:
        if rng.random() < 0.3:
            return repr(float(np.round(rng.uniform(-2, 2), 3)))
        return leaves[rng.integers(len(leaves))]
    a = _random_expr(rng, leaves, depth - 1)
    b = _random_expr(rng, leaves, depth - 1)
    kind = rng.integers(9)
    if kind == 0:
        return '({} + {})'.format(a, b)
    if kind == 1:
        return '({} - {})'.format(a, b)
    if kind == 2:
        return '({} * {})'.format(a, b)
    if kind == 3:
        return '({} / (1.5 + np.abs({})))'.format(a, b)
    if kind == 4:
        return 'np.where({} > {}, {}, {})'.format(a, b, a, _random_expr(rng, leaves, depth - 1))
    if kind == 5:
        return 'np.minimum({}, {})'.format(a, b)
    if kind == 6:
        return 'np.maximum({}, {})'.format(a, b)
    if kind == 7:
        return 'np.sqrt(np.abs({}))'.format(a)
    return '(-{}) ** 2'.format(a)


@pytest.mark.parametrize('seed', range(8))
def test_random_models_match_numpy_bit_for_bit(gpu, seed):
    """tracer + code generator fuzz: random dyn / cost expressions (2 states,
    1 control, 1 perturbation), one sweep on the GPU against the numpy oracle
    calling the very same Python callables"""
    rng = np.random.default_rng(1000 + seed)
    separable = seed % 2 == 0                  # exercise both kernel families
    lead = _random_expr(rng, ['x', 'y', 'u'] if separable else ['x', 'y', 'u', 'w'], 3)
    trail = _random_expr(rng, ['y', 'w'] if separable else ['x', 'y', 'u', 'w'], 3)
    cst = _random_expr(rng, ['x', 'y', 'u', 'w'], 4)
    ns = {'np': np}
    exec('def dyn(x, y, u, w):\n    return (0.5 * x + 0.2 * ({}), 0.5 * y + 0.2 * ({}))\n'
         'def cost(x, y, u, w):\n    return {} + 0.0 * u\n'.format(lead, trail, cst), ns)
    s = SysDescription((2, 1, 1), name='fuzz %d' % seed)
    s.dyn, s.cost = ns['dyn'], ns['cost']
    s.control_box = lambda x, y: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.3)]
    solver = DPSolver(s)
    solver.discretize_state(-1, 1, 13, -1, 1, 11)
    solver.discretize_perturb(-0.6, 0.6, 5)
    solver.control_steps = (0.25,)
    model = solver._traced()
    assert not isinstance(model, TraceError)
    # a power of a state-only sub-expression is a numpy SCALAR power (libm pow) in the
    # reference: flagged by the tracer, one ulp off in ~0.1 % of the values
    assert model.inexact_ops() in ([], ['scalar_pow'])
    V = rng.standard_normal((13, 11))
    with np.errstate(all='ignore'):
        J, u = solver.value_iteration(V, report_time=False)
        Jo, uo, io, mo = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    # storage-separable -> column kernel; trailing axes that depend on the control but not on
    # the leading state -> column kernel with a table per control; anything else -> staged tiles
    # (round 3: state variables the perturbation does not reach, all of them controlled or followed by an
    # exogenous rest -> the reduced-array sweep, csrc/sdp_lead_kernel.h)
    from stodynprog_amd import codegen
    lead_family = bool(codegen.lead_filter_applies(model, np.float64)) and not model.storage_separable
    # (round 4: a stock that is not listed first -- the exogenous variable comes before it -- takes the same
    # sweep on a permuted view of the axes instead of a warning and the slow families)
    permuted = (not lead_family and not model.storage_separable and codegen.lead_order(model, np.float64) is not None)
    lead_family = lead_family or permuted
    assert (solver.backend_info.get('controlled_order') is not None) == permuted
    # (round 5: on these small grids the direct kernel takes what the staged tiles and the table per control took --
    # DPSolver.STAGED_MIN_NODES, PERCONTROL_MIN_NODES; those two families have their own files)
    assert solver.backend_info['kernel'] == ('lead' if lead_family else
                                             ('column' if model.storage_separable else 'generic'))
    assert not solver.backend_info.get('table_per_control')
    if model.bit_exact:
        assert np.array_equal(J, Jo, equal_nan=True), (lead, trail, cst)
        assert np.array_equal(solver.last_policy_index, io)
        assert np.array_equal(u, uo)
    else:
        assert np.allclose(J, Jo, rtol=1e-13, atol=1e-15, equal_nan=True), (lead, trail, cst)
        clear = mo > 1e-12 * max(1.0, np.nanmax(np.abs(Jo)))
        assert np.array_equal(solver.last_policy_index[clear], io[clear])


def _what_wrong_entries_hold(A, ref, n0):
    """for a failure report of the host-array path: where the entries of A that differ from `ref` lie (quarter of
    the column range = phase of sdp_problem_backup_host) and what they hold -- the 0xA5 bytes the host array was
    filled with before the call (never written), the 0xF1 bytes the device J was filled with (copied before the
    kernel wrote it), zeros, or anything else (a conversion buffer read before it was written, ...)"""
    A, ref = np.ascontiguousarray(A), np.ascontiguousarray(ref)
    u = np.uint64 if A.dtype.itemsize == 8 else np.uint32
    bad = (A.view(u) != ref.view(u)).reshape(n0, -1)
    if not bad.any():
        return 'identical'
    per_col = A[0].size
    cols = np.nonzero(bad)[1] // (bad.shape[1] // per_col)
    quarters = np.bincount(cols * 4 // per_col, minlength=4).tolist()
    b = A.reshape(-1).view(np.uint8).reshape(-1, A.dtype.itemsize)[bad.reshape(-1)]
    held = {name: int((b == byte).all(axis=1).sum()) for name, byte in (('host sentinel 0xA5', 0xA5), ('device J poison 0xF1', 0xF1), ('zero', 0))}
    held['other'] = len(b) - sum(held.values())
    rows = np.nonzero(bad)[0]
    return '{} wrong entries, rows {}..{}, by quarter of the columns {}, holding {}'.format(
        int(bad.sum()), rows.min(), rows.max(), quarters, held)


def _host_path_pair(make, monkeypatch, rel_dp=False, poison=True):
    """a.value_iteration (ONE library call: sdp_problem_backup_host, phased on a large grid) against
    b.value_iterations (set_value / sweep / get_value / get_policy).  So that an entry which reaches the host
    without having been written cannot look right by accident, the arrays on the way are filled first: the
    page-locked result arrays with 0xA5 bytes, the device's J with 0xF1 bytes."""
    from stodynprog_amd import _native as nat
    orig = nat.pinned_empty

    def filled(shape, dtype):
        A = orig(shape, dtype)
        A.view(np.uint8).reshape(-1)[:] = 0xA5
        return A
    filled.fills = True
    if not getattr(orig, 'fills', False):
        monkeypatch.setattr(nat, 'pinned_empty', filled)
    _, a = make()
    _, b = make()
    V = np.random.default_rng(21).standard_normal(a._state_grid_shape).astype(a.dtype)
    if rel_dp:
        V = V - V[a._state_ref_ind]
    arg = (V, 0.) if rel_dp else V
    if poison:
        prob = a._problem()
        prob.set_value(np.frombuffer(b'\xf1' * 8, dtype=a.dtype)[:1].repeat(V.size).reshape(V.shape))
        prob.swap()                                          # (the poisoned buffer is J now; V is uploaded by the call)
    Ja, pa = quiet(a.value_iteration, arg, rel_dp)
    Jb, pb = quiet(b.value_iterations, arg, 1, rel_dp)          # set_value / sweep / get_value / get_policy
    if rel_dp:
        assert Ja[1] == Jb[1]
        Ja, Jb = Ja[0], Jb[0]
    assert a._state_grid_shape == Ja.shape and Ja.nbytes >= 8 << 20
    n0 = Ja.shape[0]
    assert np.array_equal(Ja, Jb), 'J: ' + _what_wrong_entries_hold(Ja, Jb, n0)
    assert np.array_equal(pa, pb), 'policy: ' + _what_wrong_entries_hold(pa, pb, n0)
    assert np.array_equal(a.last_policy_index, b.last_policy_index)
    return a


def test_host_array_path_in_phases_equals_the_plain_downloads(gpu, monkeypatch):
    """value_iteration on a large grid runs the backup in phases and downloads the finished
    rows under the next phase's kernel (strided 2-D copies of transposed sub-blocks in the
    column layout; sdp_problem_backup_host).  Same arrays as the step-by-step entry points
    -- two controls (16-byte policy rows), relative DP, float32, a node-layout problem"""
    def ar1():
        return models.storage_ar1(n_E=1200, n_P=1000, steps=(1.0, 0.1))
    s = _host_path_pair(ar1, monkeypatch)
    assert s.backend_info['kernel'] == 'column' and len(s.sys.control) == 2
    _host_path_pair(ar1, monkeypatch, rel_dp=True)

    def f32():
        sysd, ref = models.synthetic3d(N=144)
        ref.dtype = np.dtype(np.float32)
        return sysd, ref
    _host_path_pair(f32, monkeypatch)

    def staged():
        sysd, ref = models.synthetic3d_coupled(N=104, cross=0.1)
        ref.control_steps = (0.5,)
        return sysd, ref
    s = _host_path_pair(staged, monkeypatch)
    assert s.backend_info['kernel'] == 'staged'
    # the switch of the overlap (DPSolver.host_overlap = False: one launch, then the downloads): same arrays
    def plain():
        sysd, ref = ar1()
        ref.host_overlap = False
        return sysd, ref
    _host_path_pair(plain, monkeypatch)


def test_host_array_path_again_and_again_in_the_state_the_suite_left(gpu, monkeypatch):
    """Round 4 saw this path return a J with wrong entries ONCE, in a full-suite run on a work-in-progress tree,
    and never in loops of the test alone.  Here the call is repeated with fresh problems --
    fresh device buffers, fresh conversion buffers, whatever streams and allocations the tests before this one
    left behind -- and with poisoned arrays, so that a recurrence says what it copied and from where."""
    def ar1():
        return models.storage_ar1(n_E=1200, n_P=1000, steps=(1.0, 0.1))

    def f64():
        return models.synthetic3d(N=144)
    for k in range(24):
        _host_path_pair(ar1 if k % 3 else f64, monkeypatch)
