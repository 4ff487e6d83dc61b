"""numpy restatement of the reference's value-iteration path, with the user's
Python callables evaluated exactly the way the reference evaluates them.

TEST INFRASTRUCTURE ONLY (see oracle/sdp_oracle.c header).  Slow by design:
one Python iteration per state node, like stodynprog/stodynprog.py:511-515.

Every function cites the reference lines it follows (paths relative to the
reference checkout: sdp.py = stodynprog/stodynprog.py, pyx =
stodynprog/dolointerpolation/multilinear_cython.pyx).

Pinning: tests/test_oracle.py checks this module against tests/golden/*.npz,
which tests/golden/make_golden.py produced by importing the real reference.
"""
import itertools

import numpy as np

_I32_MIN = -2 ** 31


def mlinterp_np(smin, smax, orders, values, s):
    """pyx:17-49 dispatcher + pyx:51-300 point kernels, vectorised over points.

    Each numpy operator below is one IEEE operation per element, in the order
    of the Cython source, so results are bit-identical to the compiled
    reference for float32 and float64.
    """
    values = np.ascontiguousarray(values)
    dt = values.dtype.type
    s = np.ascontiguousarray(s, dtype=dt)
    d, n_s = s.shape
    if d < 1 or d > 4:
        raise Exception("Can't interpolate in dimension strictly greater than 5")  # pyx:47
    orders = [int(o) for o in orders]
    q, lam = [], []
    for k in range(d):
        sn = (s[k] - dt(smin[k])) / (dt(smax[k]) - dt(smin[k]))        # pyx:75
        p = sn * dt(orders[k] - 1)
        with np.errstate(invalid='ignore'):
            ok = np.abs(p) < dt(2147483648.0)
            t = np.where(ok, p, dt(0)).astype(np.int64)                # C truncation
        t = np.where(ok, t, _I32_MIN)                                  # x86 cvtt* indefinite
        qk = np.maximum(np.minimum(t, orders[k] - 2), 0)               # pyx:78
        q.append(qk)
        lam.append(p - qk.astype(dt))                                  # pyx:81
    M = [1] * d
    for k in range(d - 2, -1, -1):
        M[k] = M[k + 1] * orders[k + 1]
    out = np.zeros((values.shape[0], n_s), dtype=dt)
    # Cython emits the `1` of `(1-lam_k)` as the C double literal 1.0
    # (pyx:88,140,208,300), so in the float specialisation the lerp tree is
    # evaluated in double -- except the innermost `lam*v` product, float x
    # float -- and rounded to float once, on the store.
    lam64 = [l.astype(np.float64) for l in lam]
    for v in range(values.shape[0]):
        V = values[v]

        def rec(k, base):
            if k == d - 1:
                lo = V[base + M[k] * q[k]]
                hi = V[base + M[k] * (q[k] + 1)]
                return (1.0 - lam64[k]) * lo.astype(np.float64) + (lam[k] * hi).astype(np.float64)
            lo = rec(k + 1, base + M[k] * q[k])
            hi = rec(k + 1, base + M[k] * (q[k] + 1))
            return (1.0 - lam64[k]) * lo + lam64[k] * hi
        out[v] = rec(0, np.zeros(n_s, dtype=np.int64)).astype(dt)
    return out


class Interp:
    """sdp.py:255-290 (MlinInterpolator) on top of mlinterp_np."""

    def __init__(self, *x_grid):
        self.ndim = len(x_grid)
        self._xmin = np.array([x[0] for x in x_grid], dtype=float)     # sdp.py:263
        self._xmax = np.array([x[-1] for x in x_grid], dtype=float)    # sdp.py:264
        self._xshape = np.array([len(x) for x in x_grid], dtype=np.int64)
        self.values = None

    def set_values(self, values):
        assert values.shape == tuple(self._xshape)
        self.values = np.ascontiguousarray(np.atleast_2d(values.ravel()), dtype=float)

    def __call__(self, *x_interp):
        x_mesh = np.broadcast_arrays(*x_interp)                         # sdp.py:281
        shape = x_mesh[0].shape
        x_stack = np.vstack([x.astype(float).ravel() for x in x_mesh])  # sdp.py:283
        a = mlinterp_np(self._xmin, self._xmax, self._xshape, self.values, x_stack)
        return a.reshape(shape)


class Spec:
    """Plain container for a discretised problem (what DPSolver holds)."""

    def __init__(self, dyn, cost, control_box, state_grid, perturb_grid, perturb_proba,
                 control_steps, params=None, stationnary=True):
        self.dyn, self.cost, self.control_box = dyn, cost, control_box
        self.state_grid = [np.asarray(g, dtype=float) for g in state_grid]
        self.perturb_grid = [np.asarray(g, dtype=float) for g in perturb_grid]
        self.perturb_proba = [np.asarray(g, dtype=float) for g in perturb_proba]
        self.control_steps = tuple(control_steps)
        self.params = params or {}
        self.stationnary = stationnary
        self.shape = tuple(len(g) for g in self.state_grid)
        self.ref_ind = tuple(n // 2 for n in self.shape)                # sdp.py:384

    @classmethod
    def from_solver(cls, dpsolv):
        s = dpsolv.sys
        return cls(s.dyn, s.cost, s.control_box, dpsolv.state_grid, dpsolv.perturb_grid,
                   dpsolv.perturb_proba, dpsolv.control_steps, s.params, s.stationnary)


def control_grids(spec, state_k, t_k=None):
    """sdp.py:432-463."""
    if t_k is not None:
        state_k = (t_k,) + tuple(state_k)
    intervals = spec.control_box(*state_k, **spec.params)               # sdp.py:440
    grids, dims = [], []
    for (u_min, u_max), step in zip(intervals, spec.control_steps):
        width = u_max - u_min
        n_interv = width / step                                         # sdp.py:447
        if n_interv < 0.1:
            npts = 1
            u_grid = np.array([(u_min + u_max) / 2])                    # sdp.py:453
        else:
            npts = int(np.ceil(n_interv) + 1)                           # sdp.py:457
            u_grid = np.linspace(u_min, u_max, npts)
        grids.append(u_grid)
        dims.append(npts)
    return grids, tuple(dims)


def backup_node(spec, x_k, J_next_interp, t_k=None, full=False):
    """sdp.py:639-691 (_value_at_state_vect) plus index and margin."""
    u_grids, control_dims = control_grids(spec, x_k, t_k)
    nb_control = len(u_grids)
    for i in range(nb_control):
        u_grids[i] = u_grids[i].reshape((1,) * i + (-1,) + (1,) * (nb_control - i))
    nb_perturb = len(spec.perturb_grid)
    args = tuple(x_k) + tuple(u_grids) + tuple(spec.perturb_grid)       # sdp.py:668
    if t_k is not None:
        args = (t_k,) + args
    x_next = spec.dyn(*args, **spec.params)                             # sdp.py:674
    g_k_grid = spec.cost(*args, **spec.params)                          # sdp.py:676
    J_k_grid = g_k_grid + J_next_interp(*x_next)                        # sdp.py:677
    if nb_perturb == 0:
        J = J_k_grid
    else:
        # sdp.py:681 uses np.inner (BLAS, summation order not source-defined);
        # the restatement sums sequentially in w order like the HIP kernel.
        w_proba = spec.perturb_proba[0]
        Jb = np.broadcast_to(J_k_grid, control_dims + (len(w_proba),))
        J = np.zeros(control_dims)
        for w in range(len(w_proba)):
            J = J + Jb[..., w] * w_proba[w]
    J = np.asarray(J, dtype=float).reshape(control_dims)
    flat = int(J.argmin())                                              # sdp.py:686
    ind_opt = np.unravel_index(flat, control_dims)
    J_opt = J[ind_opt]
    u_opt = [u_grids[i].flatten()[ind_opt[i]] for i in range(nb_control)]
    Jr = J.ravel()
    if Jr.size > 1:
        margin = np.partition(Jr, 1)[1] - Jr[flat] if not np.isnan(Jr).any() else 0.0
    else:
        margin = np.inf
    if full:
        return J_opt, u_opt, flat, margin, J
    return J_opt, u_opt, flat, margin


def value_iteration(spec, J_next, rel_dp=False, t_k=None, nodes=None):
    """sdp.py:466-534.  Returns (J_k | (J_k, J_ref)), pol_k, idx_k, margin_k.

    `nodes`: optional iterable of flat C-order node ids; then 1-D outputs for
    those nodes only (used for sampled parity on big grids).
    """
    if rel_dp:
        J_next, _ = J_next
        assert J_next[spec.ref_ind] == 0.                               # sdp.py:488
    interp = Interp(*spec.state_grid)
    interp.set_values(np.asarray(J_next, dtype=float))
    nu = len(spec.control_steps)
    if nodes is not None:
        nodes = np.asarray(nodes, dtype=np.int64)
        J = np.zeros(len(nodes)); pol = np.zeros((len(nodes), nu))
        idx = np.zeros(len(nodes), dtype=np.int64); mar = np.zeros(len(nodes))
        for n, flat in enumerate(nodes):
            ind = np.unravel_index(flat, spec.shape)
            x_k = tuple(g[i] for g, i in zip(spec.state_grid, ind))
            J[n], pol[n], idx[n], mar[n] = backup_node(spec, x_k, interp, t_k)
        return J, pol, idx, mar
    J_k = np.zeros(spec.shape)
    pol_k = np.zeros(spec.shape + (nu,))
    idx_k = np.zeros(spec.shape, dtype=np.int64)
    mar_k = np.zeros(spec.shape)
    state_ind = itertools.product(*[range(n) for n in spec.shape])      # sdp.py:480
    for ind_x, x_k in zip(state_ind, itertools.product(*spec.state_grid)):
        J_k[ind_x], pol_k[ind_x], idx_k[ind_x], mar_k[ind_x] = \
            backup_node(spec, x_k, interp, t_k)
    if rel_dp:
        J_ref = J_k[spec.ref_ind]                                       # sdp.py:524
        J_k -= J_ref
        return (J_k, J_ref), pol_k, idx_k, mar_k
    return J_k, pol_k, idx_k, mar_k


def eval_policy(spec, pol, n_iter, rel_dp=False, J_zero=None, J_ref_full=False):
    """sdp.py:693-775 with the expectation summed sequentially in w order."""
    dims = spec.shape
    nb_state = len(dims)
    J_pol = np.zeros(dims) if J_zero is None else J_zero
    J_ref = np.zeros(n_iter)
    nb_control = pol.shape[-1]
    w_k = spec.perturb_grid[0]
    w_proba = spec.perturb_proba[0]
    state_grid = tuple(np.reshape(spec.state_grid[i], (1,) * i + (-1,) + (1,) * (nb_state - i))
                       for i in range(nb_state))                        # sdp.py:732-739
    for k in range(n_iter):
        interp = Interp(*spec.state_grid)
        interp.set_values(J_pol)
        u_k = [pol[..., i].reshape(dims + (1,)) for i in range(nb_control)]
        args = state_grid + tuple(u_k) + (w_k,)
        x_next = spec.dyn(*args, **spec.params)
        g = spec.cost(*args, **spec.params)
        J_k_grid = np.broadcast_to(g + interp(*x_next), dims + (len(w_k),))
        J_pol = np.zeros(dims)
        for w in range(len(w_proba)):
            J_pol = J_pol + J_k_grid[..., w] * w_proba[w]
        if rel_dp:
            J_ref[k] = J_pol[spec.ref_ind]                              # sdp.py:761
            J_pol -= J_ref[k]
    if rel_dp:
        return J_pol, (J_ref if J_ref_full else J_ref[-1])
    return J_pol
