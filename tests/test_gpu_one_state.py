"""ONE state variable (the reference's tutorial doc/example_inventory.py:29-33 and its `det/` examples): since round 5
such a problem runs as the same problem with a second, inert state variable wherever that gets the filtered column kernel
(DPSolver._lifted_1d, solver._Embedded1D; VERDICT r04: "d = 1 problems never enter the column family").  The claim: J,
policy and policy index equal those of the problem as written (`embed_1d = False`: the LDS-staged tile kernel) and of the
numpy oracle -- every entry point that carries arrays, special values, per-node boxes, relative DP, a finite horizon."""
import contextlib
import io

import numpy as np
import pytest

from stodynprog_amd import models, SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw

pytestmark = pytest.mark.gpu


def _inventory(n_x=600, n_u=257, n_w=16, box_on_state=False):
    sysd = SysDescription((1, 1, 1), name='inventory')
    sysd.dyn = lambda x, u, w: (x + u - w,)
    sysd.cost = lambda x, u, w: 0.5 * u + 0.1 * (x - 2.0) * (x - 2.0)
    sysd.control_box = (lambda x: ((0., 8. + 0.1 * x),)) if box_on_state else (lambda x: ((0., 8.),))
    sysd.perturb_laws = [NormalLaw(2.0, 0.8)]
    s = DPSolver(sysd)
    s.discretize_state(-8., 24., n_x)
    s.discretize_perturb(0., 4., n_w)
    s.control_steps = (8. / (n_u - 1),)
    return s


def _storage(n_x=512, n_u=129):
    """nothing random in the dynamics"""
    sysd = SysDescription((1, 1, 1), name='storage')
    sysd.dyn = lambda x, u, w: (x + 0.9 * u,)
    sysd.cost = lambda x, u, w: (u - 0.3 * w) * (u - 0.3 * w) + 0.05 * x
    sysd.control_box = lambda x: ((-1., 1.),)
    sysd.perturb_laws = [NormalLaw(0., 1.)]
    s = DPSolver(sysd)
    s.discretize_state(0., 10., n_x)
    s.discretize_perturb(-2., 2., 8)
    s.control_steps = (2. / (n_u - 1),)
    return s


def _both(make, V, **kw):
    out = []
    for embed in (True, False):
        s = make()
        s.embed_1d = embed
        for k, v in kw.items():
            setattr(s, k, v)
        with np.errstate(all='ignore'):
            J, pol = s.value_iteration(V, report_time=False)
        out.append((J, pol, s.last_policy_index, s))
    return out


def _same(a, b):
    assert a[0].shape == b[0].shape and a[1].shape == b[1].shape and a[2].shape == b[2].shape
    assert np.array_equal(a[0], b[0], equal_nan=True), 'J differs'
    assert np.array_equal(a[2], b[2]), 'policy index differs'
    assert np.array_equal(a[1], b[1], equal_nan=True), 'policy differs'


@pytest.mark.parametrize('make', [_inventory, lambda: _inventory(box_on_state=True), _storage, lambda: _inventory(100, 65, 9)])
def test_the_lifted_problem_gives_the_values_of_the_problem_as_written(gpu, make):
    from oracle import vi_numpy
    V = np.random.default_rng(41).standard_normal(make()._state_grid_shape)
    emb, plain = _both(make, V)
    assert emb[3].backend_info['kernel'] == 'column' and emb[3].backend_info['certified_filter'] and emb[3].backend_info['embedded_1d']
    assert plain[3].backend_info['kernel'] == 'staged' and not plain[3].backend_info.get('embedded_1d')
    assert emb[0].shape == V.shape and emb[1].shape == V.shape + (1,)
    _same(emb, plain)
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(make()), V)
    assert np.array_equal(emb[0], Jo) and np.array_equal(emb[2], io) and np.array_equal(emb[1], uo)


@pytest.mark.parametrize('case', ['nan', 'inf', '-inf', 'huge', 'subnormal', 'zeros', 'minus zero'])
def test_special_values(gpu, case):
    V = np.random.default_rng(42).standard_normal(600)
    if case == 'nan':
        V[100:120] = np.nan
    elif case == 'inf':
        V[:50] = np.inf
    elif case == '-inf':
        V[500:] = -np.inf
    elif case == 'huge':
        V *= 1e302
    elif case == 'subnormal':
        V *= 1e-310
    elif case == 'zeros':
        V[:] = 0.0
    else:
        V[::2] = -0.0
    emb, plain = _both(_inventory, V)
    _same(emb, plain)          # (-0.0 == 0.0: where the cost-to-go is a signed zero, the sign is all that may differ)


def test_loops_on_the_device_relative_dp_and_policy_evaluation(gpu):
    V = np.zeros(600)
    out = []
    for embed in (True, False):
        s = _inventory()
        s.embed_1d = embed
        with contextlib.redirect_stdout(io.StringIO()):
            (J, ref), pol = s.value_iterations((V, 0.), 30, rel_dp=True, report_time=False)
            idx = s.last_policy_index
            (J1, ref1), pol1 = s.value_iteration((J, ref), rel_dp=True, report_time=False)
            E, Eref = s.eval_policy(pol, 25, True)
            (Jp, refp), polp = s.policy_iteration(pol, 20, 3, rel_dp=True)
        out.append((J, ref, pol, idx, J1, ref1, pol1, E, Eref, Jp, refp, polp, s.backend_info.get('embedded_1d', False)))
    assert out[0][-1] and not out[1][-1]
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert np.array_equal(a, b)
    assert out[0][0][_inventory()._state_ref_ind] == 0.0


def test_a_finite_horizon_with_time_dependent_callables(gpu):
    """models.finite_horizon (golden G8's problem): dynamics, cost and box depend on the time index"""
    out = []
    for embed in (True, False):
        _, s = models.finite_horizon(n_x=65)
        s.embed_1d = embed
        with contextlib.redirect_stdout(io.StringIO()):
            J, pol = s.bellman_recursion(6, np.zeros(65))
        out.append((J, pol, s.backend_info.get('embedded_1d', False)))
    assert out[0][2] and not out[1][2]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_where_the_lifted_problem_is_not_used(gpu):
    """a grid whose table no longer fits the LDS, a forced kernel, 4-byte reals without a filter for the shape: the
    problem runs as written; `simulate` always does (its trajectories are the 1-D system's)"""
    s = _inventory(65536, 1025, 16)
    assert s._lifted_1d(None) is None
    s = _inventory()
    s.kernel = 'staged'
    assert s._lifted_1d(None) is None
    s = _inventory()
    J, pol = s.value_iteration(np.zeros(600), report_time=False)
    assert s.backend_info['embedded_1d']
    rng = np.random.default_rng(5)
    x, u, g = s.simulate(pol, np.array([[1.0], [3.0]]), rng.uniform(0, 4, size=(20, 2)))
    assert x.shape == (21, 2, 1) and not s.backend_info.get('embedded_1d')
    s2 = _inventory()
    s2.embed_1d = False
    x2, u2, g2 = s2.simulate(pol, np.array([[1.0], [3.0]]), np.random.default_rng(5).uniform(0, 4, size=(20, 2)))
    assert np.array_equal(x, x2) and np.array_equal(u, u2) and np.array_equal(g, g2)


def test_4_byte_reals(gpu):
    """the storage shape in 4-byte reals takes the wide pass of the column kernel through the lifted problem; the inventory
    shape has no 4-byte filter and runs as written"""
    V = np.random.default_rng(43).standard_normal(512).astype(np.float32)
    emb, plain = _both(_storage, V, dtype=np.dtype(np.float32))
    assert emb[0].dtype == np.float32 and emb[3].backend_info.get('embedded_1d') and emb[3].backend_info['kernel'] == 'column'
    _same(emb, plain)
    s = _inventory()
    s.dtype = np.dtype(np.float32)
    assert s._lifted_1d(None) is None
