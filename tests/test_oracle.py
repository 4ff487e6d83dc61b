"""The CPU oracle (oracle/) pinned against golden vectors produced by the real
reference (tests/golden/make_golden.py) and against the reference's own known
answers.  No GPU needed."""
import numpy as np
import pytest

from conftest import golden, assert_sweep_parity
from oracle import c_oracle, vi_numpy
from stodynprog_amd import models


def test_interp_bit_exact_all_cases():
    g = golden('g1_interp')
    n = int(g['n_cases'])
    assert n == 16
    for c in range(n):
        p = 'c{:02d}_'.format(c)
        ref = g[p + 'out']
        for impl in (c_oracle.mlinterp, vi_numpy.mlinterp_np):
            out = impl(g[p + 'smin'], g[p + 'smax'], g[p + 'orders'], g[p + 'values'], g[p + 's'])
            assert out.dtype == ref.dtype
            assert np.array_equal(out, ref, equal_nan=True), (c, impl.__name__)


def test_interp_known_answer_of_reference_test():
    # reference stodynprog/tests/test_dolointerp.py:17-40
    s = np.linspace(0., 2., 5)[None, :]
    vals = np.array([[0., 1., 4.]])
    for impl in (c_oracle.mlinterp, vi_numpy.mlinterp_np):
        out = impl(np.array([0.]), np.array([2.]), np.array([3]), vals, s)
        assert np.all(np.abs(out - np.array([0, 0.5, 1, 2.5, 4])) < 1e-10)
    g = golden('g1_interp')
    assert np.array_equal(g['ka_out'], out)


def test_interp_extrapolation_and_cast_edge_cases():
    g = golden('g1_interp')
    vals = np.array([[0., 1., 4.]])
    for impl in (c_oracle.mlinterp, vi_numpy.mlinterp_np):
        with np.errstate(all='ignore'):
            out = impl(np.array([0.]), np.array([2.]), np.array([3]), vals, g['ex_s'])
        assert np.array_equal(out, g['ex_out'], equal_nan=True)
    assert g['ex_out'][0, 0] == -1.0 and g['ex_out'][0, 1] == 7.0


def test_interp_object_api_case():
    # reference tests/test_dolointerp.py:45-93: exact at the grid corners
    g = golden('g1_interp')
    out = c_oracle.mlinterp([1., 1.], [2., 2.], [5, 5], g['mli_values'], g['mli_pts'])
    assert np.array_equal(out, g['mli_out'])
    x = g['mli_pts']
    exact = np.vstack([np.sqrt(x[0] ** 2 + x[1] ** 2), np.power(x[0] ** 3 + x[1] ** 3, 1 / 3.)])
    assert np.all(np.abs(out[:, :4] - exact[:, :4]) < 1e-9)
    assert np.all(np.abs(out - exact) < 0.01)


def test_interp_dimension_5_raises():
    with pytest.raises(Exception):
        c_oracle.mlinterp(np.zeros(5), np.ones(5), [2] * 5, np.zeros((1, 32)), np.zeros((5, 1)))


def test_inventory_known_answers_and_golden():
    # doc/example_inventory.rst:220-239
    g = golden('g2_inventory')
    _, solver = models.inventory()
    spec = vi_numpy.Spec.from_solver(solver)
    assert np.array_equal(solver.state_grid[0], g['state_grid'])
    assert np.array_equal(solver.perturb_proba[0], g['perturb_proba'])
    J = np.zeros(10)
    expected_pol = {0: [0] * 10, 1: [4, 3, 2, 1, 0, 0, 0, 0, 0, 0],
                    2: [5, 4, 3, 2, 1, 0, 0, 0, 0, 0], 3: [5, 4, 3, 2, 1, 0, 0, 0, 0, 0]}
    for k in range(6):
        J, pol, idx, mar = vi_numpy.value_iteration(spec, J)
        assert_sweep_parity(J, idx, g['J'][k], g['idx'][k], g['margin'][k], 'inventory %d' % k)
        assert np.array_equal(pol, g['pol'][k])
        if k in expected_pol:
            assert np.array_equal(pol[:, 0], expected_pol[k])
        if k == 0:
            assert np.allclose(J, [9, 6, 3, 0, 0.5, 1, 1.5, 2, 2.5, 3], rtol=0, atol=1e-14)


def test_nas_demo_golden():
    g = golden('g7_nas')
    _, solver = models.nas_demo()
    spec = vi_numpy.Spec.from_solver(solver)
    J1, pol1, idx1, _ = vi_numpy.value_iteration(spec, np.zeros(spec.shape))
    assert_sweep_parity(J1, idx1, g['J1'], g['idx1'], g['margin1'], 'nas 1')
    J2, pol2, idx2, _ = vi_numpy.value_iteration(spec, g['J1'])
    assert_sweep_parity(J2, idx2, g['J2'], g['idx2'], g['margin2'], 'nas 2')


def test_storage_ar1_sampled_nodes_golden():
    """reference-size problem (41x61, up to 8001 controls): sampled nodes only,
    the numpy oracle costs ~2 ms per node like the reference."""
    g = golden('g3_ar1_ref')
    _, solver = models.storage_ar1()
    spec = vi_numpy.Spec.from_solver(solver)
    assert np.array_equal(solver.perturb_proba[0], g['perturb_proba'])
    rng = np.random.default_rng(3)
    nodes = np.unique(np.concatenate([rng.integers(0, 41 * 61, 120), [0, 41 * 61 - 1]]))
    J, pol, idx, _ = vi_numpy.value_iteration(spec, g['J1'], nodes=nodes)
    assert_sweep_parity(J, idx, g['J2'].ravel()[nodes], g['idx2'].ravel()[nodes],
                        g['margin2'].ravel()[nodes], 'ar1 sweep 2')
    # control counts quoted by the notebook (AR1.ipynb:362-365): 4001 .. 8001
    assert g['npts'][..., 0].min() == 4001 and g['npts'][..., 0].max() == 8001
    assert abs(g['npts'][..., 0].mean() - 6342.5) < 0.5
    assert (g['npts'][..., 1] == 1).all()


def test_searev_sampled_golden_with_c_tabulated_oracle():
    g = golden('g4_searev')
    _, solver = models.searev(n_E=128, n_S=128, n_A=128, step=2.2 / 31)
    spec = vi_numpy.Spec.from_solver(solver)
    assert np.array_equal(solver.perturb_proba[0], g['perturb_proba'])
    E, S, A = solver.state_grid_full
    V0 = np.ascontiguousarray(0.02 * (E - 5.) * (E - 5.) + 1.5 * (S * S) + 0.7 * (A * A)
                              + 0.1 * S * A - 0.01 * E)
    nodes = g['nodes'][::8]
    J, pol, idx, _ = vi_numpy.value_iteration(spec, V0, nodes=nodes)
    assert_sweep_parity(J, idx, g['J'][::8], g['idx'][::8], g['margin'][::8], 'searev c3')
    assert np.array_equal(pol, g['pol'][::8])


def test_synthetic_c_oracle_vs_reference_golden():
    """the hand-written C model of the benchmark problem against the reference
    run on 4099 sampled nodes of the 256^3 grid and on the full 20^3 grid"""
    g = golden('g5_synth')
    _, solver = models.synthetic3d()
    V0 = models.synthetic3d_V0(solver.state_grid)
    import zlib
    assert zlib.crc32(V0.tobytes()) == int(g['V0_crc'])
    assert np.array_equal(solver.perturb_proba[0], g['perturb_proba'])
    J, idx, mar = c_oracle.vi_synth3d(solver.state_grid, V0, models.SYNTH_PAR, -1., 1., 64,
                                      solver.perturb_grid[0], solver.perturb_proba[0],
                                      node_ids=g['nodes'], n_threads=4)
    err, ndiff = assert_sweep_parity(J, idx, g['J'], g['idx'], g['margin'], 'synthetic 256^3')
    assert ndiff == 0                      # strictly convex cost: unique minimiser
    _, small = models.synthetic3d(N=20)
    V0s = models.synthetic3d_V0(small.state_grid)
    J1, idx1, _ = c_oracle.vi_synth3d(small.state_grid, V0s, models.SYNTH_PAR, -1., 1., 64,
                                      small.perturb_grid[0], small.perturb_proba[0],
                                      n_nodes=V0s.size)
    assert_sweep_parity(J1.reshape(V0s.shape), idx1.reshape(V0s.shape), g['s_J1'], g['s_idx1'],
                        g['s_margin1'], 'synthetic 20^3')
    # numpy oracle with the Python callables agrees with the C model bit for bit
    spec = vi_numpy.Spec.from_solver(small)
    nodes = np.arange(0, V0s.size, 37)
    Jn, _, idxn, _ = vi_numpy.value_iteration(spec, V0s, nodes=nodes)
    assert np.array_equal(Jn, J1[nodes]) and np.array_equal(idxn, idx1[nodes])


def test_c_tabulated_oracle_matches_numpy_oracle():
    _, solver = models.nas_demo(n_E=9, n_P=7, n_w=5)
    spec = vi_numpy.Spec.from_solver(solver)
    rng = np.random.default_rng(0)
    V = rng.standard_normal(spec.shape)
    interp = vi_numpy.Interp(*spec.state_grid)
    interp.set_values(V)
    import itertools
    offs, xs, gs, Js, idxs = [0], [], [], [], []
    for x_k in itertools.product(*spec.state_grid):
        J_opt, u, flat, mar = vi_numpy.backup_node(spec, x_k, interp)
        grids, dims = vi_numpy.control_grids(spec, x_k)
        u0 = grids[0].reshape(-1, 1)
        args = tuple(x_k) + (u0,) + tuple(spec.perturb_grid)
        xn = spec.dyn(*args)
        lattice = dims + (len(spec.perturb_grid[0]),)
        xs.append(np.vstack([np.broadcast_to(a, lattice).ravel() for a in xn]))
        gs.append(np.broadcast_to(spec.cost(*args), lattice).ravel())
        offs.append(offs[-1] + gs[-1].size)
        Js.append(J_opt); idxs.append(flat)
    J, idx, _ = c_oracle.vi_tab(interp._xmin, interp._xmax, interp._xshape, V, offs,
                                len(spec.perturb_grid[0]), spec.perturb_proba[0],
                                np.concatenate(xs, axis=1), np.concatenate(gs))
    assert np.array_equal(J, np.array(Js)) and np.array_equal(idx, np.array(idxs))


def test_eval_policy_oracle_vs_reference():
    g = golden('g6_policy')
    _, solver = models.storage_ar1()
    spec = vi_numpy.Spec.from_solver(solver)
    pol = models.storage_ar1_empirical_policy(solver)
    assert np.array_equal(pol, g['ar1_pol_ini'])
    J, J_ref = vi_numpy.eval_policy(spec, pol, 50, rel_dp=True, J_ref_full=True)
    assert np.allclose(J_ref, g['ar1_J_ref'], rtol=1e-12, atol=1e-14)
    assert np.abs(J - g['ar1_J']).max() < 1e-12
    # published value (AR1.ipynb:594): reference cost of the empirical policy 0.105724
    assert '{:g}'.format(J_ref[-1]) == '0.105724'
    J7 = vi_numpy.eval_policy(spec, pol, 7)
    assert np.abs(J7 - g['ar1_J7']).max() < 1e-12


def test_bellman_recursion_oracle_vs_reference():
    g = golden('g8_bellman')
    _, solver = models.finite_horizon()
    spec = vi_numpy.Spec.from_solver(solver)
    nxt = g['J_fin']
    for t in (4, 3, 2, 1, 0):
        J, pol, _, _ = vi_numpy.value_iteration(spec, nxt, t_k=t)
        assert np.abs(J - g['J'][t]).max() <= 1e-13 * max(1.0, np.abs(g['J'][t]).max())
        assert np.array_equal(pol, g['pol'][t])
        nxt = J


def test_pv_storage_time_indexed_data_oracle_vs_reference():
    """finite horizon whose cost looks data up by time index (the shape of the
    reference's examples/01 .../pv_storage_control.py): numpy oracle against
    the reference's bellman_recursion, all 48 steps, bit for bit"""
    g = golden('g9_pv_storage')
    _, solver = models.pv_storage()
    assert np.array_equal(solver.P_prod_data, g['P_prod'])
    spec = vi_numpy.Spec.from_solver(solver)
    nxt = np.zeros(50)
    for t in range(47, -1, -1):
        J, pol, _, _ = vi_numpy.value_iteration(spec, nxt, t_k=t)
        assert np.array_equal(J, g['J'][t]), t
        assert np.array_equal(pol, g['pol'][t]), t
        nxt = J
