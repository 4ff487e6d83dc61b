"""Batched closed-loop simulation on the device (DPSolver.simulate ->
sdp_problem_simulate) against golden trajectories produced by the REFERENCE's
own loop (examples/20 Searev storage control/storage_control.py:242-251, run by
tests/golden/make_golden.py g10) and against the hand-written host loop."""
import numpy as np
import pytest

from conftest import golden
from stodynprog_amd import models

pytestmark = pytest.mark.gpu


def test_searev_trajectories_match_the_reference_loop(gpu):
    g = golden('g10_simulation')
    wec, solver = models.searev(n_E=11, n_S=15, n_A=13)
    assert np.array_equal(models.searev_linear_policy(solver), g['pol'])
    x, u, cost = solver.simulate(g['pol'], g['x0'], g['w'])
    assert solver.backend_info['mode'] == 'traced'
    assert x.shape == (401, 5, 3) and u.shape == (400, 5, 1) and cost.shape == (400, 5)
    # the searev cost squares a state-only sub-expression: numpy scalar pow in the reference
    # loop (1 ulp, flagged `scalar_pow`); states and controls involve no such operation
    assert np.array_equal(x, g['x'])
    assert np.array_equal(u, g['u'])
    assert np.allclose(cost, g['g'], rtol=1e-15, atol=0)
    # one trajectory: the example's start state
    x1, u1, c1 = solver.simulate(g['pol'], g['x0'][0], g['w'][:, 0])
    assert x1.shape == (401, 3) and np.array_equal(x1, g['x'][:, 0]) and np.array_equal(u1, g['u'][:, 0])


def test_two_controls_match_the_reference_loop(gpu):
    g = golden('g10_simulation')
    _, ar1 = models.storage_ar1(n_E=21, n_P=25)
    y, v, _ = ar1.simulate(g['pol2'], g['y0'], g['w2'])
    assert np.array_equal(y, g['y'])
    assert np.array_equal(v, g['v'])


def test_simulation_equals_the_hand_written_host_loop(gpu):
    """the loop of examples/searev_storage.py, step by step through interp_on_state"""
    wec, solver = models.searev(n_E=9, n_S=11, n_A=10)
    J, pol = solver.value_iteration(np.zeros((9, 11, 10)), report_time=False)
    rng = np.random.default_rng(3)
    T = 150
    w = rng.normal(0., models.SEAREV['innov_std'], T)
    law = solver.interp_on_state(np.ascontiguousarray(pol[..., 0]))
    E = np.zeros(T + 1); S = np.zeros(T + 1); A = np.zeros(T + 1)
    E[0] = 10 / 3.
    P = np.zeros(T)
    for k in range(T):
        P[k] = law(E[k], S[k], A[k])
        E[k + 1], S[k + 1], A[k + 1] = wec.dyn(E[k], S[k], A[k], P[k], w[k])
    x, u, _ = solver.simulate(pol, (10 / 3., 0., 0.), w)
    assert np.array_equal(x, np.column_stack([E, S, A])) and np.array_equal(u[:, 0], P)
    # float32 solver: same loop through the float32 interpolator and numpy float32 arithmetic
    xb, ub, _ = solver.simulate(pol, np.tile([10 / 3., 0., 0.], (70, 1)), np.tile(w[:, None], (1, 70)))
    assert all(np.array_equal(xb[:, b], x) for b in range(70))     # more lanes than a wavefront


def test_deterministic_time_dependent_system(gpu):
    """finite-horizon model with a time index: u_k from the policy of step k is not what
    simulate does (one policy array), but the time index reaches dyn and cost"""
    fh, solver = models.finite_horizon()
    pol = np.linspace(-0.5, 0.5, 17)[:, None]
    w = np.random.default_rng(0).normal(0, 0.1, 12)
    x, u, g = solver.simulate(pol, (0.3,), w, t0=2)
    law = solver.interp_on_state(pol[:, 0])
    xs = [0.3]
    for k in range(12):
        uk = float(law(xs[-1]))
        assert uk == u[k, 0]
        assert g[k] == fh.cost(2 + k, xs[-1], uk, w[k])
        xs.append(fh.dyn(2 + k, xs[-1], uk, w[k])[0])
    assert np.array_equal(x[:, 0], xs)


def test_untraceable_model_runs_the_host_loop(gpu):
    """callables that branch on values cannot be traced: simulate then runs the
    reference's loop itself (interpolation on the device, callbacks on the host)"""
    from stodynprog_amd import SysDescription, DPSolver
    sysd = SysDescription((1, 1, 1), name='branchy')

    def dyn(x, u, w):
        return (np.array([xi + ui if xi > 0 else xi - ui for xi, ui in zip(np.atleast_1d(x), np.atleast_1d(u))]) + w,)
    sysd.dyn = dyn
    sysd.cost = lambda x, u, w: x * x + u * u + 0 * w
    sysd.control_box = lambda x: ((-1., 1.),)
    sysd.perturb_laws = [models.NormalLaw(0, 0.1)]
    s = DPSolver(sysd)
    s.discretize_state(-2, 2, 9)
    s.discretize_perturb(-0.2, 0.2, 3)
    pol = (0.1 * s.state_grid[0])[:, None]
    x, u, g = s.simulate(pol, [[0.5], [-0.5]], np.zeros((4, 2)))
    assert x.shape == (5, 2, 1) and np.allclose(x[1, :, 0], [0.55, -0.45])


def test_empty_batch_and_zero_steps(gpu):
    wec, solver = models.searev(n_E=5, n_S=5, n_A=5)
    pol = models.searev_linear_policy(solver)
    x, u, g = solver.simulate(pol, np.zeros((3, 3)) + [[5., 0., 0.]], np.zeros((0, 3)))
    assert x.shape == (1, 3, 3) and u.shape == (0, 3, 1) and g.shape == (0, 3)
    assert np.array_equal(x[0], np.zeros((3, 3)) + [[5., 0., 0.]])
