import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# parity bars (BASELINE.json north_star): fp64 |dV|/|V| < 1e-10, policy indices
# exact.  The reference's expectation is np.inner (BLAS summation order), ours
# is sequential in w, so J agrees to ~1e-15 and the argmin can only differ
# where two controls tie to that level (SURVEY section 7 "hard parts").
RTOL_J = 1e-10
TIE_RTOL = 1e-12


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box)')
    # a fresh checkout has no libsdp_hip.so yet: build it once (hipcc cross-compiles
    # gfx950 without a GPU); the tests themselves never fall back to anything else
    from stodynprog_amd import _native as nat
    if not os.path.exists(nat.LIB_PATH) and os.path.exists(nat.HIPCC):
        nat.build_library()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def _report(line):
    """append to the parity report (committed as profiles/rNN_parity_report.txt) when
    SDP_PARITY_REPORT names a file"""
    path = os.environ.get('SDP_PARITY_REPORT')
    if path:
        with open(path, 'a') as f:
            f.write(line.rstrip('\n') + '\n')


def prove_ties(solver, J_next, flat_nodes, idx_ours, idx_ref, what, tie_rtol=TIE_RTOL, t_k=None):
    """Per-node proof that an argmin index differing from the reference's is a TIE:
    the complete per-control expected-cost vector of the node is recomputed with
    the CPU oracle (oracle/vi_numpy.backup_node, pinned against the reference's
    golden vectors by tests/test_oracle.py) from the same cost-to-go, and the costs
    of the two indices must agree within tie_rtol * max(1, |J|).  Returns the
    largest cost gap found (0.0 when nothing differs)."""
    from oracle import vi_numpy
    spec = vi_numpy.Spec.from_solver(solver)
    interp = vi_numpy.Interp(*spec.state_grid)
    interp.set_values(np.asarray(J_next, dtype=float))
    worst = 0.0
    for flat, io, ir in zip(flat_nodes, idx_ours, idx_ref):
        ind = np.unravel_index(int(flat), spec.shape)
        x_k = tuple(g[i] for g, i in zip(spec.state_grid, ind))
        J_opt, u_opt, flat_opt, margin, Jfull = vi_numpy.backup_node(spec, x_k, interp, t_k, full=True)
        costs = np.asarray(Jfull, dtype=float).ravel()
        gap = abs(costs[int(io)] - costs[int(ir)])
        tol = tie_rtol * max(1.0, abs(float(J_opt)))
        assert gap <= tol, ('{}: node {} picks control {} (cost {!r}) where the reference picks {} '
                            '(cost {!r}): gap {:.3e} > {:.3e}, not a tie'.format(
                                what, int(flat), int(io), costs[int(io)], int(ir), costs[int(ir)], gap, tol))
        # and neither may be worse than the true minimum by more than the tie level
        assert costs[int(io)] - costs.min() <= tol and costs[int(ir)] - costs.min() <= tol, what
        worst = max(worst, gap)
    return worst


def assert_sweep_parity(J, idx, J_ref, idx_ref, margin_ref, what='', rtol=RTOL_J,
                        tie_rtol=TIE_RTOL, prove=None, nodes=None):
    """J within rtol of the reference everywhere; argmin index identical, or PROVED
    a tie node by node: `prove` = (solver, J_next[, t_k]) lets every differing node be
    re-evaluated with the CPU oracle (prove_ties); `nodes` = flat C-order ids of the
    compared entries when they are a sample of the grid.  Without `prove` any
    difference fails.  Returns (max relative error of J, number of differing indices)."""
    J, J_ref = np.asarray(J, dtype=float), np.asarray(J_ref, dtype=float)
    scale = max(1.0, float(np.abs(J_ref).max()))
    err = np.abs(J - J_ref).max() / scale
    assert err < rtol, '{}: max |dJ|/|J| = {:.3e}'.format(what, err)
    idx, idx_ref = np.asarray(idx).astype(np.int64), np.asarray(idx_ref).astype(np.int64)
    diff = idx != idx_ref
    ndiff = int(diff.sum())
    gap = 0.0
    if ndiff:
        assert prove is not None, ('{}: {} argmin indices differ from the reference and no tie proof '
                                   'was requested'.format(what, ndiff))
        where = np.flatnonzero(diff.ravel())
        flat = where if nodes is None else np.asarray(nodes).ravel()[where]
        gap = prove_ties(prove[0], prove[1], flat, idx.ravel()[where], idx_ref.ravel()[where], what,
                         tie_rtol, prove[2] if len(prove) > 2 else None)
        # the reference's own best / second-best margin must say "tie" as well
        tie = np.asarray(margin_ref).ravel()[where] <= tie_rtol * np.maximum(1.0, np.abs(J_ref.ravel()[where]))
        assert tie.all(), '{}: {} differing nodes have a reference margin above the tie level'.format(
            what, int((~tie).sum()))
    _report('{:32s} nodes {:9d}  max|dJ|/|J| {:.2e}  index differences {:5d}  (all proved ties, '
            'largest cost gap {:.2e})'.format(what, diff.size, err, ndiff, gap))
    return err, ndiff


def have_gpu():
    try:
        from stodynprog_amd import _native as nat
        return nat.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu():
    from stodynprog_amd import _native as nat
    nat.require_gpu()          # loud failure: GPU tests never fall back to anything
    nat.check(nat.lib().sdp_set_device(0))
    return nat


@pytest.fixture
def debug_defines(monkeypatch):
    """Diagnostic switches of the generated kernels (stodynprog_amd.codegen.DEBUG_NAMES) for the
    solvers a test creates: an explicit dict on the DPSolver class for the duration of the test
    (undone by monkeypatch).  The product never reads such switches from the environment --
    tests/test_trace_codegen.py::test_the_environment_does_not_reach_the_generated_source."""
    from stodynprog_amd import DPSolver

    class Switches(object):
        def set(self, **kw):
            cur = dict(DPSolver.debug_defines or {})
            cur.update({k: str(v) for k, v in kw.items()})
            monkeypatch.setattr(DPSolver, 'debug_defines', cur)

        def unset(self, *names):
            cur = {k: v for k, v in (DPSolver.debug_defines or {}).items() if k not in names}
            monkeypatch.setattr(DPSolver, 'debug_defines', cur or None)

    return Switches()
