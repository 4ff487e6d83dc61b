"""Sparse peer exchange (DPSolver.comm_sparse): are the need lists complete?  A rank's backups
must read the cost-to-go array ONLY in the rows `DPSolver._peer_needs` lists for it (plus its
own).  Checked on one GPU without any communicator: every row outside a rank's list and slab
is poisoned with NaN, the whole grid is swept, and the rank's own nodes must come out exactly as
from the clean array -- value, policy and index, for the sweep and for a policy evaluation."""
import io
import contextlib

import numpy as np
import pytest

from stodynprog_amd import models, dist

pytestmark = pytest.mark.gpu


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize('name,kw,nranks,phases', [
    ('synthetic3d', dict(N=32), 4, 3),
    ('synthetic3d', dict(N=24), 3, 2),
    ('storage_ar1', dict(), 2, 4),
    ('nas_demo', dict(), 3, 2),
])
def test_rows_outside_the_need_list_are_never_read(gpu, name, kw, nranks, phases):
    _, s = getattr(models, name)(**kw)
    shape = s._shape()
    n0 = shape[0]
    n_cols = int(np.prod(shape[1:]))
    model = s._traced()
    assert model.storage_separable and s._kernel_plan()['column']
    parts = dist.slab_partition(n_cols, n0, nranks, phases)
    off, ranges = s._peer_needs(model, parts, shape)
    V = np.random.default_rng(7).standard_normal(shape)
    J, pol = s.value_iteration(V, report_time=False)
    idx = s.last_policy_index
    E = _quiet(s.eval_policy, pol, 2, False, V)
    # columns in the order the library stores them: trailing multi-index, C order
    for r in range(nranks):
        readable = np.zeros(n_cols, dtype=bool)
        own = np.zeros(n_cols, dtype=bool)
        for ph in range(parts.shape[0]):
            own[parts[ph, r] // n0:parts[ph, r + 1] // n0] = True
        readable |= own
        for b, e in ranges[off[r]:off[r + 1]]:
            readable[b // n0:e // n0] = True
        assert 0 < readable.sum() < n_cols or nranks == 1
        Vp = V.reshape(n0, n_cols).copy()
        Vp[:, ~readable] = np.nan
        Vp = Vp.reshape(shape)
        with np.errstate(all='ignore'):
            Jp, polp = s.value_iteration(Vp, report_time=False)
            idxp = s.last_policy_index
            Ep = _quiet(s.eval_policy, pol, 1, False, Vp)        # (one step: E of the poisoned array's owners)
        E1 = _quiet(s.eval_policy, pol, 1, False, V)
        sel = own.reshape(shape[1:])
        for a, b in ((J, Jp), (idx, idxp), (E1, Ep)):
            assert np.array_equal(a[:, sel] if a.ndim == len(shape) else a[:, sel], b[:, sel]), (name, r)
        assert np.array_equal(pol[:, sel], polp[:, sel])
        # the test has teeth: without the listed rows (only its own slab left) the rank's nodes change
        Vq = V.reshape(n0, n_cols).copy()
        Vq[:, ~own] = np.nan
        with np.errstate(all='ignore'):
            Jq, _ = s.value_iteration(Vq.reshape(shape), report_time=False)
        assert not np.array_equal(J[:, sel], Jq[:, sel], equal_nan=True), (name, r)
    assert E.shape == V.shape
