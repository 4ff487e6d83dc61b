"""Host-side drop-in API: same behaviour as the reference's description layer
and discretisation helpers.  The first block restates the reference's own unit
tests (stodynprog/tests/test_stodynprog.py) against this package.  No GPU."""
import io
import contextlib
import pickle

import numpy as np
import pytest

import stodynprog_amd
from stodynprog_amd import SysDescription, DPSolver, models
from stodynprog_amd.sysdesc import _zero_cost, _enforce_sig_len
from conftest import golden


# ---- reference tests/test_stodynprog.py:15-65 --------------------------------
def test_zero_cost():
    assert _zero_cost() == 0.
    assert _zero_cost(1) == 0.
    assert _zero_cost(1, 2, 3) == 0.


def test_enforce_sig_len():
    def f0():
        pass

    def f1(x):
        pass

    def f2(x, y):
        pass
    arg0, arg1, arg2 = [], ['x'], ['x', 'y']
    assert _enforce_sig_len(f0, arg0, False)
    assert _enforce_sig_len(f1, arg1, False)
    assert _enforce_sig_len(f2, arg2, False)
    for f, a in ((f0, arg1), (f0, arg2), (f1, arg0), (f1, arg2), (f2, arg1)):
        with pytest.raises(ValueError):
            _enforce_sig_len(f, a, False)
    with pytest.raises(ValueError) as e:
        _enforce_sig_len(f1, arg2, False)
    assert e.value.args[0] == "'f1' should accept 2 args (x, y), not 1"   # :58

    def f1p(x, **params):
        pass
    assert _enforce_sig_len(f1p, arg1, with_params=True)
    with pytest.raises(ValueError):
        _enforce_sig_len(f1p, arg1, with_params=False)
    with pytest.raises(ValueError):
        _enforce_sig_len(f1, arg1, with_params=True)
    with pytest.raises(ValueError) as e:
        _enforce_sig_len(f1, arg2, False, 'dynamics function')
    assert e.value.args[0].startswith("dynamics function'f1' should accept")


# ---- reference tests/test_stodynprog.py:69-107 ---------------------------------
def test_sysdescription_attributes_and_names():
    sys110 = SysDescription((1, 1, 0), stationnary=True, name='sys110')
    sys111 = SysDescription((1, 1, 1), stationnary=True, name='sys111')
    assert sys111.stationnary and sys111.stochastic and not sys110.stochastic
    assert sys111.name == 'sys111'
    assert sys111.state == ['x1'] and sys111.control == ['u1'] and sys111.perturb == ['w1']

    def dyn3(my_state, my_control, my_perturb):
        pass
    sys111.dyn = dyn3
    assert sys111.state == ['my_state']
    assert sys111.control == ['my_control']
    assert sys111.perturb == ['my_perturb']

    def dyn2(x, u):
        pass

    def dyn4(x, y, u, w):
        pass
    with pytest.raises(ValueError):
        sys111.dyn = dyn2
    with pytest.raises(ValueError):
        sys111.dyn = dyn4
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sys111.print_summary()
    assert 'Dynamical system "sys111" description' in buf.getvalue()
    assert repr(sys111).startswith('<SysDescription "sys111" at 0x')


def test_sysdescription_dims_and_laws():
    with pytest.raises(ValueError) as e:
        SysDescription((1,))
    assert e.value.args[0] == 'dims tuple should be of len 2 or 3'
    s = SysDescription((2, 1))
    assert s.perturb == [] and not s.stochastic
    s = SysDescription((1, 1, 1))
    with pytest.raises(ValueError):
        s.perturb_laws = []
    with pytest.raises(ValueError):
        s.perturb_laws = [object()]
    s.perturb_laws = [models.NormalLaw(0, 1)]
    assert s.perturb_types == ['continuous']
    s.perturb_laws = [models.DiscreteLaw([0, 1], [.5, .5])]
    assert s.perturb_types == ['discrete']

    def tc(a, b):
        return 0
    with pytest.raises(ValueError):
        s.terminal_cost = tc
    # non-stationnary systems take the time index first (sdp.py:89-91)
    ns = SysDescription((1, 1, 0), stationnary=False)

    def dyn_t(k, x, u):
        return (x + u,)
    ns.dyn = dyn_t
    assert ns.state == ['x'] and ns.control == ['u']

    def box_t(k, x):
        return ((0, 1),)
    ns.control_box = box_t
    # parameters are forwarded as keyword arguments and must be accepted
    ps = SysDescription((1, 1, 0), params={'a': 2.})

    def dyn_np(x, u):
        return (x + u,)
    with pytest.raises(ValueError):
        ps.dyn = dyn_np


def test_scipy_laws_give_the_same_weights():
    stats = pytest.importorskip('scipy.stats')
    _, solver = models.searev()
    grid = solver.perturb_grid[0]
    p = stats.norm(loc=0, scale=models.SEAREV['innov_std']).pdf(grid)
    p /= p.sum()
    assert np.array_equal(p, solver.perturb_proba[0])
    law = stats.rv_discrete(values=([0, 1, 2, 3], [0.2, 0.4, 0.3, 0.1])).freeze()
    assert np.allclose(law.pmf(np.linspace(0, 3, 4)), models.inventory()[1].perturb_proba[0])


# ---- discretisation helpers ------------------------------------------------------
def test_inventory_discretisation_matches_the_tutorial():
    # doc/example_inventory.rst:182,188
    _, solver = models.inventory()
    assert np.array_equal(solver.state_grid[0], np.arange(-3., 7.))
    assert np.array_equal(solver.perturb_grid[0], [0., 1., 2., 3.])
    assert np.allclose(solver.perturb_proba[0], [0.2, 0.4, 0.3, 0.1])
    assert solver._state_grid_shape == (10,)
    assert solver._state_ref_ind == (5,) and solver._state_ref == (2.0,)
    grids, dims = solver.control_grids((0.,))
    assert dims == (11,) and np.array_equal(grids[0], np.arange(0., 11.))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        solver.print_summary()
    out = buf.getvalue()
    assert '* state space discretized on a 10 points grid' in out
    assert 'yields 11 possible values' in out


def test_control_grids_rules():
    _, solver = models.storage_ar1()
    # width/step < 0.1 -> one point at the centre (sdp.py:449-453)
    grids, dims = solver.control_grids((0., 0.))
    assert dims == (4001, 1)
    assert grids[1][0] == 0.0
    assert grids[0][0] == 0.0 and grids[0][-1] == 4.0
    grids, dims = solver.control_grids((5., 1.))
    assert dims == (8001, 1)
    # the vectorised box table agrees with control_grids on every node
    lo, hi, n = solver._box_table()
    g = golden('g3_ar1_ref')
    assert np.array_equal(n.T.reshape(41, 61, 2), g['npts'])
    import itertools
    for flat, x in enumerate(itertools.product(*solver.state_grid)):
        if flat % 97:
            continue
        grids, dims = solver.control_grids(x)
        assert tuple(n[:, flat]) == dims
        assert lo[0, flat] == grids[0][0] and hi[0, flat] == grids[0][-1]
        assert lo[1, flat] == grids[1][0]


def test_box_table_falls_back_to_scalar_calls():
    calls = []
    s = SysDescription((1, 1, 0))

    def dyn(x, u):
        return (x + u,)
    s.dyn = dyn

    def box(x):
        calls.append(x)
        if x > 0.5:                       # only works on scalars
            return ((0., 1.),)
        return ((0., 2.),)
    s.control_box = box
    solver = DPSolver(s)
    solver.discretize_state(0, 1, 5)
    solver.control_steps = (0.5,)
    lo, hi, n = solver._box_table()
    assert list(n[0]) == [5, 5, 5, 3, 3]
    assert list(hi[0]) == [2., 2., 2., 1., 1.]


def test_interp_on_state_errors_and_state_grid_full():
    _, solver = models.nas_demo()
    with pytest.raises(ValueError) as e:
        solver.interp_on_state(np.zeros((3, 3)))
    assert e.value.args[0] == 'array `A` should be of shape (51, 41), not (3, 3)'
    full = solver.state_grid_full
    assert full[0].shape == (51, 41) and full[1].shape == (51, 41)
    assert np.array_equal(full[0][:, 0], solver.state_grid[0])
    it = solver.interp_on_state(np.zeros((51, 41)))
    assert it.ndim == 2 and it.values.shape == (1, 51 * 41)
    # pickles with the reference's attribute names (searev/P_sto_law.dat format)
    it2 = pickle.loads(pickle.dumps(it))
    assert set(vars(it2)) == {'ndim', '_xmin', '_xmax', '_xshape', 'values'}
    assert np.array_equal(it2._xshape, [51, 41])


def test_value_iteration_argument_checks_need_no_gpu():
    _, solver = models.inventory()
    with pytest.raises(ValueError):
        solver.value_iteration(np.zeros(9), report_time=False)
    with pytest.raises(AssertionError):       # rel_dp needs a differential cost (sdp.py:488)
        solver.value_iteration((np.ones(10), 0.), rel_dp=True, report_time=False)
    with pytest.raises(AssertionError):
        solver.discretize_state(0, 1)
    with pytest.raises(AssertionError):
        solver.eval_policy(np.zeros((10, 2)), 1)


def test_package_surface():
    assert stodynprog_amd.SysDescription is SysDescription
    from stodynprog_amd.dolointerpolation import (MultilinearInterpolator,
                                                  multilinear_interpolation, mlinspace)
    g = mlinspace([0, 0], [1, 2], [2, 3])
    assert g.shape == (2, 6) and np.array_equal(g[1], [0, 1, 2, 0, 1, 2])
    assert mlinspace([0.], [1.], [3]).shape == (1, 3)
    mli = MultilinearInterpolator([0, 0], [1, 2], [2, 3])
    assert mli.d == 2 and mli.grid.shape == (2, 6)
    assert callable(multilinear_interpolation)
    for name in ('discretize_perturb', 'discretize_state', 'state_grid_full', 'interp_on_state',
                 'control_grids', 'value_iteration', 'bellman_recursion', '_value_at_state_loop',
                 '_value_at_state_vect', 'eval_policy', 'policy_iteration', 'print_summary'):
        assert hasattr(DPSolver, name), name


def test_reference_pickle_format_round_trip(tmp_path):
    """interpolator pickles carry the reference's class path and attribute
    names (searev/P_sto_law.dat format) in both directions"""
    import pickletools
    from stodynprog_amd import compat, MlinInterpolator
    _, solver = models.nas_demo(n_E=5, n_P=4)
    it = solver.interp_on_state(np.arange(20.).reshape(5, 4))
    path = tmp_path / 'law.dat'
    compat.dump_interpolator(it, str(path))
    raw = path.read_bytes()
    names = [arg for op, arg, _ in pickletools.genops(raw) if op.name == 'GLOBAL']
    assert 'stodynprog.stodynprog MlinInterpolator' in names
    assert not any(n.startswith('stodynprog_amd') for n in names)
    back = compat.load_interpolator(str(path))
    assert isinstance(back, MlinInterpolator)
    assert set(vars(back)) == {'ndim', '_xmin', '_xmax', '_xshape', 'values'}
    assert back.ndim == 2 and np.array_equal(back.values, it.values)
    assert np.array_equal(back._xshape, [5, 4]) and back._xshape.dtype == np.int64
    import sys
    assert 'stodynprog.stodynprog' not in sys.modules      # the writer cleans up its shim


@pytest.mark.skipif(not __import__('os').path.exists(
    '/root/reference/examples/20 Searev storage control/P_sto_law.dat'),
    reason='reference checkout not present (GPU box)')
def test_load_the_reference_searev_policy_pickle():
    from stodynprog_amd import compat
    it = compat.load_interpolator('/root/reference/examples/20 Searev storage control/P_sto_law.dat')
    assert it.ndim == 3 and list(it._xshape) == [31, 61, 61]
    assert it.values.shape == (1, 31 * 61 * 61)
    assert np.allclose(it._xmax, [10., 1.016, 0.908])


def test_control_box_table_vectorises_the_tuple_min_max_idiom():
    """the reference's control_box callbacks use np.max((a, b)) on scalars
    (AR1 notebook cell 15, searev/storage_control.py:76-78): scalar-only as
    written.  The table builder TRACES them (np.max / np.min of a tuple become
    the DAG's max / min) and evaluates the DAG on the whole grid: the
    node-by-node table bit for bit; numpy itself is never touched."""
    import itertools
    from stodynprog_amd import models, solver as solver_mod
    for name, kw, t in (('searev', dict(n_E=12, n_S=7, n_A=5), None), ('storage_ar1', {}, None),
                        ('pv_storage', {}, 5), ('nas_demo', {}, None)):
        _, s = getattr(models, name)(**kw)
        box = s.sys.control_box
        lo, hi, n = s._box_table(t)
        S = lo.shape[1]
        lead = () if t is None else (t,)
        for flat, x in enumerate(itertools.product(*s.state_grid)):
            for c, (a, b) in enumerate(box(*(lead + x))):
                n_interv = (b - a) / s.control_steps[c]
                col = flat if S > 1 else 0
                if n_interv < 0.1:
                    assert n[c, col] == 1 and lo[c, col] == (a + b) / 2
                else:
                    assert n[c, col] == int(np.ceil(n_interv) + 1)
                    assert lo[c, col] == a and hi[c, col] == b
    assert np.max((1.0, 3.0)) == 3.0 and np.max.__name__ in ('max', 'amax')
    assert np.max(np.array([[1., 5.], [2., 0.]]), axis=0).tolist() == [2., 5.]
    # a box that is NOT what the elementwise reading gives must not be accepted
    _, s = models.storage_ar1()
    s.sys.control_box = lambda E, P: ((float(np.max((np.min((E, 2.)), 0.5))) - 3.0, 1.0), (0., 0.))
    lo, hi, n = s._box_table()
    E = s.state_grid[0]
    assert np.array_equal(lo[0].reshape(41, 61)[:, 0], np.maximum(np.minimum(E, 2.), 0.5) - 3.0)


def test_interpolator_loader_refuses_anything_but_data(tmp_path):
    """compat.load_interpolator reads third-party policy files: a pickle that names any
    global besides the interpolator class and numpy's array reconstruction is refused"""
    import os
    import pickle
    from stodynprog_amd.compat import load_interpolator

    class Evil(object):
        def __reduce__(self):
            return (os.system, ('echo pwned',))
    path = tmp_path / 'evil.dat'
    path.write_bytes(pickle.dumps(Evil(), protocol=2))
    with pytest.raises(pickle.UnpicklingError):
        load_interpolator(str(path))


@pytest.mark.parametrize('grid', [(41, 61), (200, 200), (500, 500)])
def test_a_cached_control_box_table_notices_data_the_callback_reads(grid):
    """The reference calls control_box at every node of every sweep (stodynprog.py:440); here the table is cached, and the
    callback is TRACED again on every call: data it reads are constants of its DAG, the table stays while the DAG is the
    same (a proof for every node: no scalar call, no sample) and is rebuilt when module-level data changed."""
    from stodynprog_amd import SysDescription, DPSolver
    rated = {'P': 1.0}
    sysd = SysDescription((2, 1, 1), name='storage')
    sysd.dyn = lambda E, P, u, w: (E + u, 0.8 * P + w)
    sysd.cost = lambda E, P, u, w: (P - u) * (P - u)
    calls = []

    def box(E, P):
        calls.append(type(E).__name__)
        return ((np.max((-E, -rated['P'])), np.min((10. - E, rated['P']))),)
    sysd.control_box = box
    sysd.perturb_laws = [__import__('stodynprog_amd').models.NormalLaw(0, 0.5)]
    s = DPSolver(sysd)
    s.discretize_state(0, 10, grid[0], -2, 2, grid[1])
    s.discretize_perturb(-1, 1, 5)
    s.control_steps = (0.1,)
    bp = s._box_plan()
    assert bp['mode'] == 'traced' and bp['per_node']
    # building the table: one symbolic call, plus the tracer's self-check at the corners and the centre
    assert calls.count('Sym') == 1 and len(calls) == 1 + 5
    del calls[:]
    assert s._box_plan() is bp                                   # unchanged data: the cached table
    assert calls == []                                           # .. the callback's fingerprint (code, closure cells, globals) is what it was: not even a trace
    s.trace_cache = False
    assert s._box_plan() is bp and calls == ['Sym']              # (without fingerprints: one trace per call, whatever the grid's size)
    s.trace_cache = True
    del calls[:]
    rated['P'] = 0.5                                             # the data changes: the table is rebuilt
    bp2 = s._box_plan()
    assert bp2 is not bp and bp2['hi'].max() == 0.5 and bp['hi'].max() == 1.0
    assert s._box_plan() is bp2


def _box_by_scalar_calls(s, t=None):
    """the reference's control_grids at every node (stodynprog.py:432-463)"""
    import itertools
    lead = () if t is None else (t,)
    shape = tuple(len(g) for g in s.state_grid)
    nu = len(s.sys.control)
    lo = np.empty((nu,) + shape)
    hi = np.empty((nu,) + shape)
    n = np.empty((nu,) + shape, dtype=np.int32)
    for ind in itertools.product(*[range(k) for k in shape]):
        x = tuple(g[i] for g, i in zip(s.state_grid, ind))
        for c, (a, b) in enumerate(s.sys.control_box(*(lead + x), **s.sys.params)):
            n_interv = (b - a) / s.control_steps[c]
            if n_interv < 0.1:
                lo[(c,) + ind] = hi[(c,) + ind] = (a + b) / 2
                n[(c,) + ind] = 1
            else:
                lo[(c,) + ind], hi[(c,) + ind], n[(c,) + ind] = a, b, int(np.ceil(n_interv) + 1)
    return lo.reshape(nu, -1), hi.reshape(nu, -1), n.reshape(nu, -1)


def test_a_box_with_a_branch_in_one_small_region_is_exact_at_every_node():
    """a whole-grid evaluation validated on a SAMPLE accepts a box whose data-dependent branch
    bites in a small interior region.  The trace follows every path of the callback (`if`, the builtins max / min) and
    merges them with selects on the recorded conditions: the table is the scalar calls' at EVERY node, the odd one too."""
    from stodynprog_amd import SysDescription, DPSolver
    sysd = SysDescription((3, 1, 0), name='one odd node')
    sysd.dyn = lambda a, b, c, u: (a + u, b, c)
    sysd.cost = lambda a, b, c, u: u * u

    def box(a, b, c):
        hi = min(1.0, 2.0 - a)                                   # builtin min: a branch on the state
        if 0.49 < a < 0.51 and 0.29 < b < 0.31 and abs(c - 0.7) < 0.01:      # one node of the 21^3 grid
            hi = 0.25
        return ((max(-a, -1.0), hi),)
    sysd.control_box = box
    s = DPSolver(sysd)
    s.discretize_state(0, 1, 21, 0, 1, 21, 0, 1, 21)
    s.control_steps = (0.05,)
    lo, hi, n = s._box_table()
    assert s._box_mode == 'traced'
    lo_r, hi_r, n_r = _box_by_scalar_calls(s)
    assert np.array_equal(lo, lo_r) and np.array_equal(hi, hi_r) and np.array_equal(n, n_r)
    assert (hi == 0.25).sum() == 1                               # the odd node is there, once
    # the naive whole-grid call of the same callback does not even run (Python `if` on an array), and what round 5
    # accepted instead -- the branch-free part, validated on a sample -- misses the node:
    naive_hi = np.minimum(1.0, 2.0 - s.state_grid[0])
    assert (np.broadcast_to(naive_hi[:, None, None], (21, 21, 21)).ravel() != hi[0]).sum() == 1


def test_closure_data_that_changes_for_one_node_rebuilds_the_table():
    from stodynprog_amd import SysDescription, DPSolver
    odd = {'at': (0.5, 0.3), 'hi': 0.25}
    sysd = SysDescription((2, 1, 0), name='closure')
    sysd.dyn = lambda a, b, u: (a + u, b)
    sysd.cost = lambda a, b, u: u * u

    def box(a, b):
        hi = np.min((1.0, 2.0 - a))
        hi = np.where((a == odd['at'][0]) & (b == odd['at'][1]), odd['hi'], hi)
        return ((np.max((-a, -1.0)), hi),)
    sysd.control_box = box
    s = DPSolver(sysd)
    s.discretize_state(0, 1, 201, 0, 1, 101)                   # 20 301 nodes: round 5 re-checked such a grid at 24 of them
    s.control_steps = (0.05,)
    bp = s._box_plan()
    assert bp['mode'] == 'traced' and (bp['hi'] == 0.25).sum() == 1
    assert s._box_plan() is bp
    odd['hi'] = 0.125                                            # one node's box changes between two calls
    bp2 = s._box_plan()
    assert bp2 is not bp and (bp2['hi'] == 0.125).sum() == 1 and (bp2['hi'] == 0.25).sum() == 0
    lo_r, hi_r, n_r = _box_by_scalar_calls(s)
    assert np.array_equal(bp2['lo'], lo_r) and np.array_equal(bp2['hi'], hi_r) and np.array_equal(bp2['n'], n_r)
    odd['at'] = (0.25, 0.5)                                      # .. and moves to another node
    bp3 = s._box_plan()
    lo_r, hi_r, n_r = _box_by_scalar_calls(s)
    assert bp3 is not bp2 and np.array_equal(bp3['hi'], hi_r) and np.array_equal(bp3['n'], n_r)


def test_tracing_a_box_leaves_numpy_alone():
    """np.max / np.min of a tuple are read elementwise by a stand-in for the numpy module in the GLOBALS OF A COPY of the
    callback; numpy's own attributes are never rebound (round 5 patched them process-wide while the callback ran)."""
    import numpy
    from stodynprog_amd import models
    from stodynprog_amd import trace
    seen = []
    _, s = models.storage_ar1()
    inner = s.sys.control_box
    orig = (numpy.max, numpy.min, numpy.amax, numpy.amin)

    def spy(E, P):
        import numpy as real                                     # (what any other module, or thread, sees meanwhile)
        seen.append((real.max, real.min, real.amax, real.amin) == orig and real.max((1.0, 3.0)) == 3.0)
        return inner(E, P)
    s.sys._control_box = spy
    bp = s._box_plan()
    assert bp['mode'] == 'traced' and seen and all(seen)
    assert (numpy.max, numpy.min, numpy.amax, numpy.amin) == orig
    assert not hasattr(__import__('stodynprog_amd').solver, '_TupleMinMax')
    # the callback itself is untouched too: the stand-in lives in the globals of a copy
    assert inner.__globals__['np'] is numpy
    tb = trace.trace_box(inner, 2, 2)
    assert tb.paths == 1 and not tb.inexact_ops()


def test_a_box_that_cannot_be_traced_is_called_node_by_node():
    from stodynprog_amd import SysDescription, DPSolver
    table = np.linspace(1.0, 2.0, 11)
    sysd = SysDescription((1, 1, 0))
    sysd.dyn = lambda x, u: (x + u,)
    sysd.control_box = lambda x: ((0.0, float(table[int(round(x * 10))])),)       # an index from the state: no trace
    s = DPSolver(sysd)
    s.discretize_state(0, 1, 11)
    s.control_steps = (0.5,)
    bp = s._box_plan()
    assert bp['mode'] is None and bp['sig'] is None
    assert np.array_equal(bp['hi'][0], table)
    assert s._box_plan() is bp                                   # re-checked at a sample of nodes
    table[5] = 5.0                                               # (11 nodes: the sample is the whole grid)
    assert s._box_plan()['hi'][0, 5] == 5.0
    s.box_recheck = 'every node'
    bp = s._box_plan()
    assert s._box_plan() is not bp                               # rebuilt on every call, as the reference does
    # transcendental functions in a box: traceable, but numpy's whole-grid loops need not repeat the scalar call's bits
    sysd.control_box = lambda x: ((0.0, 1.0 + np.exp(-x)),)
    s2 = DPSolver(sysd)
    s2.discretize_state(0, 1, 11)
    s2.control_steps = (0.5,)
    assert s2._box_plan()['mode'] is None


def test_a_trace_is_kept_while_the_callables_fingerprint_stands():
    """dyn and cost are traced again on a call only when something their result can depend on has changed: their code,
    their defaults, the contents of their closure cells and of the globals they name (by VALUE).  What cannot be
    fingerprinted -- an object with attributes, an array beyond 64 KiB -- is traced on every call, as before."""
    from stodynprog_amd import SysDescription, DPSolver
    from stodynprog_amd.trace import callable_fingerprint
    par = {'gain': 0.5, 'table': np.linspace(0., 1., 8)}
    sysd = SysDescription((1, 1, 1))
    sysd.dyn = lambda x, u, w: (x + par['gain'] * u - w,)
    sysd.cost = lambda x, u, w: u * u + np.interp(x, np.linspace(0., 1., 8), par['table'])
    sysd.control_box = lambda x: ((0., 1.),)
    s = DPSolver(sysd)
    s.discretize_state(0, 1, 9)
    m1 = s._trace_now()
    assert s._trace_now() is m1                                  # same fingerprint: the same trace object
    par['gain'] = 0.25                                           # a mutated closure value invalidates
    m2 = s._trace_now()
    assert m2 is not m1 and m2.param_values() != m1.param_values()
    assert s._trace_now() is m2
    par['table'][3] = 7.0                                        # .. an array mutated in place too (hashed by content)
    m3 = s._trace_now()
    assert m3 is not m2
    s.trace_cache = False
    assert s._trace_now() is not s._trace_now()
    s.trace_cache = True

    class Holder(object):
        gain = 0.5
    h = Holder()
    sysd.dyn = lambda x, u, w: (x + h.gain * u - w,)              # an object's attribute: out of the fingerprint's sight
    assert callable_fingerprint(sysd.dyn) is None
    a = s._trace_now()
    assert s._trace_now() is not a                               # traced on every call
    h.gain = 0.125
    assert s._trace_now().param_values() != a.param_values()
    big = np.zeros(20000)
    sysd.dyn = lambda x, u, w: (x + big[0] * u - w,)
    assert callable_fingerprint(sysd.dyn) is None                # 160 kB: not hashed on every call
    import math
    sysd.dyn = lambda x, u, w: (x + math.sqrt(2.0) * u - w,)      # a library module and a builtin are fine
    assert callable_fingerprint(sysd.dyn) is not None
