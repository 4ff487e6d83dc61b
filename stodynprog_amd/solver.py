"""Dynamic-programming solver: the drop-in `DPSolver` class.

Host-side mirror of the reference's solver API (reference
stodynprog/stodynprog.py:317-876).  The discretisation helpers stay in numpy
(same operators as the reference, so grids and weights are bit-identical); the
sweeps run on the GPU:

  value_iteration / bellman_recursion
      -> sdp_problem_vi_sweep   (fused node x control x perturbation backup)
  eval_policy / policy_iteration
      -> sdp_problem_eval_policy (device-resident fixed-policy backups)
  interp_on_state(...)(coords)
      -> sdp_mlinterp_f64 / _f32

The user's `dyn` and `cost` callables are traced once (trace.py) and compiled
into the sweep kernel (codegen.py).  Callables that cannot be traced are
evaluated on the host exactly as the reference does and only the gather /
expectation / argmin runs on the device (sdp_tab_backup, "tabulated mode").
There is no CPU compute fallback: without the HIP library and a GPU every
sweep raises.
"""
from __future__ import division, print_function
import ctypes as C
import itertools
from datetime import datetime

import numpy as np

from . import _native as nat
from . import codegen
from .interp import MlinInterpolator
from .trace import TraceError, trace_model, trace_box, callable_fingerprint

__all__ = ['DPSolver']


class _DeviceProblem(object):
    """Owner of one sdp_problem handle (include/sdp_hip.h)."""

    def __init__(self, desc_arrays, module_path, dtype, shape, nu, W, lanes, box_per_node,
                 node_range, comm=None, slab_bounds=None, layout=0, staged=None, col_seg_nodes=0):
        self._keep = desc_arrays            # host arrays referenced by the descriptor
        self.layout = int(layout)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape)
        self.S = int(np.prod(shape))
        self.nu = nu
        # shape of the per-node arrays as the device stores them
        self.dev_shape = (self.shape[1:] + self.shape[:1]
                          if self.layout == nat.LAYOUT_COLUMNS else self.shape)
        d = nat.sdp_problem_desc()
        d.dtype = nat.np_real(dtype)
        d.d, d.nu, d.W = len(shape), nu, W
        for k, n in enumerate(shape):
            d.orders[k] = n
            d.axes[k] = desc_arrays['axes'][k].ctypes.data
        if W > 0:
            d.wgrid = desc_arrays['wgrid'].ctypes.data
            d.proba = desc_arrays['proba'].ctypes.data
        d.box_per_node = int(box_per_node)
        d.lanes_per_node = int(lanes)
        d.layout = int(layout)
        d.col_seg_nodes = int(col_seg_nodes)
        d.variant = nat.VARIANT_STAGED if staged else nat.VARIANT_DIRECT
        if staged:
            for k, n in enumerate(staged['tile']):
                d.tile[k] = int(n)
        d.box_lo = desc_arrays['box_lo'].ctypes.data
        d.box_hi = desc_arrays['box_hi'].ctypes.data
        d.box_n = desc_arrays['box_n'].ctypes.data
        d.node_begin, d.node_end = node_range
        d.module_path = module_path.encode()
        h = C.c_void_p()
        rc = nat.lib().sdp_problem_create(C.byref(d), C.byref(h))
        if comm is not None and comm.nranks > 1:
            # What follows (attaching the communicator, mapping the peers' buffers, need lists) is collective, and so
            # is what the caller does about a failure (DPSolver._problem re-plans without the reduced-array sweep when
            # the device memory does not suffice): the ranks agree HERE whether every one of them has its problem --
            # all go on or all raise (advisor, round 4: one rank out of memory left the others in a collective)
            why = None if rc == 0 else nat.lib().sdp_last_error().decode(errors='replace')
            if comm.allreduce_max(0.0 if rc == 0 else 1.0) > 0:
                if rc == 0:
                    nat.lib().sdp_problem_destroy(h)
                    raise MemoryError('another rank could not create its problem (out of device memory there?)')
                if rc != -4:
                    raise nat._ERR.get(rc, RuntimeError)(why)
                raise MemoryError(why)
        else:
            nat.check(rc)
        self.h = h
        self.node_range = tuple(node_range)
        self.parts = None
        if comm is not None:
            # slab_bounds: [n_phases][nranks+1] partition (dist.phase_partition)
            self.parts = np.ascontiguousarray(slab_bounds, dtype=np.int64)
            nat.check(nat.lib().sdp_problem_attach_comm(self.h, comm.handle,
                                                        int(self.parts.shape[0]),
                                                        nat.ptr(self.parts)))

    def close(self):
        if getattr(self, 'h', None):
            nat.lib().sdp_problem_destroy(self.h)
            self.h = None

    def unmap_peers(self):
        """first half of tearing down a sharded problem: the peers' buffers leave this process (every rank does
        this, then a barrier, then close(): memory a peer still maps must not be freed under it)"""
        if getattr(self, 'h', None):
            nat.check(nat.lib().sdp_problem_disable_peer_exchange(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # per-node arrays cross the C API in the reference's C order; the library
    # keeps them axis-0-fastest on the device when the column kernels are used.
    # These two helpers express that order in numpy (host-side gather, tests).
    def _to_device_order(self, A, extra=(), dtype=None):
        A = np.asarray(A, dtype=dtype or self.dtype).reshape(self.shape + extra)
        if self.layout == nat.LAYOUT_COLUMNS:
            A = np.moveaxis(A, 0, len(self.shape) - 1)
        return np.ascontiguousarray(A)

    def _from_device_order(self, A, extra=()):
        if self.layout == nat.LAYOUT_COLUMNS:
            A = A.reshape(self.shape[1:] + self.shape[:1] + extra)
            A = np.moveaxis(A, len(self.shape) - 1, 0)
        return np.ascontiguousarray(A).reshape(self.shape + extra)

    def set_value(self, V):
        assert np.size(V) == self.S
        V = np.ascontiguousarray(V, dtype=self.dtype)       # reference order; the library converts
        nat.check(nat.lib().sdp_problem_set_value(self.h, nat.ptr(V)))

    def set_policy(self, pol):
        assert np.size(pol) == self.S * self.nu
        pol = np.ascontiguousarray(pol, dtype=self.dtype)
        nat.check(nat.lib().sdp_problem_set_policy(self.h, nat.ptr(pol)))

    def set_params(self, values):
        """lifted model constants of the launches that follow (sdp_problem_set_params)"""
        values = np.ascontiguousarray(values, dtype=self.dtype)
        nat.check(nat.lib().sdp_problem_set_params(self.h, nat.ptr(values), int(values.size)))

    def sweep(self, t_k=0.0, rel_dp=False, ref_index=0):
        ref = C.c_double(0.0)
        nat.check(nat.lib().sdp_problem_vi_sweep(self.h, float(t_k), int(bool(rel_dp)),
                                                 int(ref_index), C.byref(ref)))
        return ref.value

    def eval_policy(self, n_iter, rel_dp=False, ref_index=0):
        refs = np.zeros(max(int(n_iter), 1))
        nat.check(nat.lib().sdp_problem_eval_policy(self.h, int(n_iter), int(bool(rel_dp)),
                                                    int(ref_index), nat.ptr(refs)))
        return refs[:int(n_iter)]

    def backup_host(self, V, t_k=0.0, rel_dp=False, ref_index=0, overlap=True):
        """value_iteration's device work with host arrays in and out in ONE library call
        (sdp_problem_backup_host): upload, sweep, relative-DP shift, download of J and of
        the policy values, one synchronisation.  Large outputs live in page-locked memory
        (ordinary writable ndarrays for the caller) so the copies run at PCIe rate; fed
        back as the next J_next they upload at that rate too.  The lattice indices stay
        on the device until asked for (get_index)."""
        V = np.ascontiguousarray(V, dtype=self.dtype)
        big = self.S * self.dtype.itemsize >= (1 << 20)
        new = nat.pinned_empty if big else np.empty
        J = new(self.shape, self.dtype)
        pol = new(self.shape + (self.nu,), self.dtype)
        ref = C.c_double(0.0)
        nat.check(nat.lib().sdp_problem_set_host_overlap(self.h, int(bool(overlap))))
        nat.check(nat.lib().sdp_problem_backup_host(self.h, nat.ptr(V), float(t_k), int(bool(rel_dp)),
                                                    int(ref_index), nat.ptr(J), nat.ptr(pol), None,
                                                    C.byref(ref)))
        return J, pol, ref.value

    def get_index(self):
        idx = np.empty(self.shape, dtype=np.int32)
        nat.check(nat.lib().sdp_problem_get_policy(self.h, None, nat.ptr(idx)))
        return idx

    def swap(self):
        nat.check(nat.lib().sdp_problem_swap(self.h))

    def complete(self):
        """sparse peer exchange: make J complete on every rank (collective; no-op otherwise)"""
        nat.check(nat.lib().sdp_problem_complete_value(self.h))

    def get_value(self):
        J = np.empty(self.shape, dtype=self.dtype)
        nat.check(nat.lib().sdp_problem_get_value(self.h, nat.ptr(J)))
        return J

    def get_policy(self):
        pol = np.empty(self.shape + (self.nu,), dtype=self.dtype)
        idx = np.empty(self.shape, dtype=np.int32)
        nat.check(nat.lib().sdp_problem_get_policy(self.h, nat.ptr(pol), nat.ptr(idx)))
        return pol, idx

    def last_kernel_ms(self):
        ms = C.c_double(0.0)
        nat.check(nat.lib().sdp_problem_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def bench_sweeps(self, reps, rel_dp=False, ref_index=0):
        loop, kern = C.c_double(0.0), C.c_double(0.0)
        nat.check(nat.lib().sdp_problem_bench_sweeps(self.h, int(reps), int(bool(rel_dp)),
                                                     int(ref_index), C.byref(loop),
                                                     C.byref(kern)))
        return loop.value, kern.value


class DPSolver(object):
    # Diagnostic / A-B switches of the generated kernels (codegen.DEBUG_NAMES): None in the product.
    # Tests and tools/ set a dict here (on an instance, or on the class for a block of solvers);
    # nothing is ever read from the environment.  A radius scale below 1, say, voids the
    # bit-identity guarantee -- which is why it takes an explicit assignment to get one.
    debug_defines = None
    # value_iteration with host arrays in and out on one GPU (sdp_problem_backup_host): True sends the finished
    # rows of a phase to the host under the next phase's kernel (grids of 8 MiB and more), False runs one launch
    # and then the downloads.  Same arrays either way.
    host_overlap = True
    # a control_box callback that cannot be traced (its table is made by scalar calls at every node, like the reference's):
    # 'sample' re-checks the cached table at 24 nodes per call, 'every node' rebuilds it on every call (stodynprog.py:440)
    box_recheck = 'sample'
    # dyn, cost and control_box are traced again on a call only when their fingerprint changed (trace.callable_fingerprint:
    # code, defaults, closure cells and the globals they name, by value; callables that have none are traced on every call);
    # False: trace on every call, as rounds 1-5 did
    trace_cache = True
    _debug_after_create = None
    STAGED_MIN_NODES = 32768          # 'auto': grids of at most this many nodes run the direct kernel, not the staged tiles (see _kernel_plan_now);
    STAGED_MIN_WORK = 1024            #         up to 4 x as many where a node has this many control x perturbation points or more
    PERCONTROL_MIN_NODES = 32768      # 'auto': .. and grids of fewer nodes than this the direct kernel rather than a table per control
    LINE_MIN_CELLS = 1 << 20          # 'auto': one state variable, x' = a(x, u) +- b(w): the filtered line kernel from this many lattice cells per sweep

    def __init__(self, sys, dtype=np.float64, comm=None):
        """Dynamic Programming solver for stochastic dynamic control of `sys`
        (a `SysDescription`).  Implements value iteration, policy evaluation,
        policy iteration and finite-horizon Bellman recursion on an AMD GPU.

        New (optional) arguments, absent from the reference:
        dtype : arithmetic type of the sweep, float64 (default) or float32
        comm  : a `stodynprog_amd.dist.Communicator` to shard the outer state
                axis over several GPUs (one process per GPU)
        """
        self.sys = sys
        self.state_grid = [[0.] for s in self.sys.state]
        self.perturb_grid = [[0.] for p in self.sys.perturb]
        self.perturb_proba = [[1.] for p in self.sys.perturb]
        self.control_steps = (1.,) * len(self.sys.control)
        self.dtype = np.dtype(dtype)
        self.comm = comm
        # 'auto': column kernel for storage-separable models whose table fits LDS, else the
        # LDS-staged tile kernel; 'column' / 'staged' / 'generic' (one global load per
        # vertex, the first kernel) force a family.  All give the same bits.
        self.kernel = 'auto'
        self.comm_phases = 4               # multi-GPU: phases per backup (comm/compute overlap)
        # multi-GPU: how the J rows of a phase reach the other ranks -- 'rccl': in-place all-gather;
        # 'peer': every rank writes its rows straight into the others' buffers (HIP IPC mappings,
        # device-to-device copies over xGMI, no compute units); 'direct': the backup kernel itself
        # stores every J it computes into the buffers of the ranks that read it (xGMI stores spread
        # over the whole sweep: one launch and one rendezvous per backup, nothing left to copy).
        # Both fall back to 'rccl' when the buffers cannot be mapped (backend_info['exchange'] tells
        # which one runs)
        self.comm_exchange = 'rccl'
        self.comm_taper = False            # multi-GPU: shrinking phases (smallest gather exposed)
        # 'peer' / 'direct' exchange only: every rank owns one slab of columns and is sent only the rows
        # of J its own backups read (computed from the model: the cells its trailing next states fall
        # in) -- for a contracting exogenous process a fraction of the array.  J is completed on every
        # rank when the host asks for it.  Full-table column kernels (all filter forms), stationary systems.
        self.comm_sparse = False
        # (every kernel performs the reference's floating-point operations in the reference's order; the opt-in
        # 'fused' arithmetic of rounds 1-5 -- J within ~1e-15, 6.6 ms per benchmark sweep -- went in round 6, when the
        # exact kernel with the certified filter ran the same sweep in 1.05 ms)
        # column kernel: decide all but the near-minimal controls of a node on
        # a table reduced over the perturbation (rigorous error radius) and evaluate only the
        # survivors with the reference's operations -- same bits, ~W times less work per control
        # (csrc/sdp_colfilter_kernel.h, SdpColFilter).  False: every control the long way.
        self.certified_filter = True
        self._cache = {}
        self._idx_cache = None             # see last_policy_index
        self._idx_source = None
        self.backend_info = {}

    @property
    def last_policy_index(self):
        """flat C-order index into the control lattice of the optimal control of every
        node, from the last sweep (int32, shape of the state grid).  After a
        value_iteration call the indices stay on the device and are fetched on first
        access (the reference returns control VALUES only, stodynprog.py:530-533)."""
        if self._idx_cache is None and self._idx_source is not None:
            prob, self._idx_source = self._idx_source, None
            if prob.h:
                self._idx_cache = prob.get_index()
        return self._idx_cache

    @last_policy_index.setter
    def last_policy_index(self, value):
        self._idx_cache, self._idx_source = value, None

    # ------------------------------------------------------------------ grids
    def discretize_perturb(self, *linspace_args):
        """create a regular discrete grid for each perturbation variable;
        grids go to `self.perturb_grid` (may also be set manually), the
        probability weights to `self.perturb_proba` (reference sdp.py:335-362)."""
        assert len(linspace_args) == len(self.sys.perturb) * 3
        self.perturb_grid = []
        self.perturb_proba = []
        for i in range(len(self.sys.perturb)):
            grid_wi = np.linspace(*linspace_args[i * 3:i * 3 + 3])
            law = self.sys.perturb_laws[i]
            if self.sys.perturb_types[i] == 'continuous':
                proba_wi = law.pdf(grid_wi)
                proba_wi /= proba_wi.sum()
            else:
                proba_wi = law.pmf(grid_wi)
                assert np.allclose(proba_wi.sum(), 1.)
            self.perturb_grid.append(grid_wi)
            self.perturb_proba.append(proba_wi)
        return self.perturb_grid, self.perturb_proba

    def discretize_state(self, *linspace_args):
        """create a regular discrete grid for each state variable, stored in
        `self.state_grid` (reference sdp.py:364-389)."""
        assert len(linspace_args) == len(self.sys.state) * 3
        self.state_grid = [np.linspace(*linspace_args[i * 3:i * 3 + 3])
                           for i in range(len(self.sys.state))]
        shape = tuple(len(g) for g in self.state_grid)
        self._state_grid_shape = shape
        # reference state of the relative DP algorithm: the middle of the grid
        self._state_ref_ind = tuple(n // 2 for n in shape)
        self._state_ref = tuple(g[i] for g, i in zip(self.state_grid, self._state_ref_ind))
        return self.state_grid

    @property
    def state_grid_full(self):
        """broadcasted state grid (self.state_grid is flat)"""
        d = len(self.state_grid)
        parts = []
        for i, g in enumerate(self.state_grid):
            shape = [1] * d
            shape[i] = -1
            parts.append(np.asarray(g).reshape(shape))
        return np.broadcast_arrays(*parts)

    def interp_on_state(self, A):
        """returns an interpolating function of array A, assumed to be given
        on the state grid (reference sdp.py:405-430)."""
        expect_shape = self._state_grid_shape
        if A.shape != expect_shape:
            raise ValueError('array `A` should be of shape {:s}, not {:s}'.format(
                str(expect_shape), str(A.shape)))
        if len(expect_shape) <= 5:
            A_interp = MlinInterpolator(*self.state_grid)
            A_interp.set_values(A)
            return A_interp
        raise NotImplementedError('interpolation for state dimension >5'
                                  ' is not implemented.')

    def _check_state_array(self, A):
        expect_shape = self._state_grid_shape
        if A.shape != expect_shape:
            raise ValueError('array `A` should be of shape {:s}, not {:s}'.format(
                str(expect_shape), str(A.shape)))

    def control_grids(self, state_k, t_k=None):
        """grid on the box of admissible controls at state `state_k`, using
        self.control_steps as hints (reference sdp.py:432-463).
        Returns (list of 1-D arrays, tuple of their lengths)."""
        if t_k is not None:
            state_k = (t_k,) + tuple(state_k)
        intervals = self.sys.control_box(*state_k, **self.sys.params)
        grids, dims = [], []
        for (u_min, u_max), step in zip(intervals, self.control_steps):
            n_interv = (u_max - u_min) / step
            if n_interv < 0.1:
                npts = 1
                u_grid = np.array([(u_min + u_max) / 2])
            else:
                npts = int(np.ceil(n_interv) + 1)
                u_grid = np.linspace(u_min, u_max, npts)
            grids.append(u_grid)
            dims.append(npts)
        return grids, tuple(dims)

    # ------------------------------------------------- host-side preparation
    def _shape(self):
        return tuple(len(g) for g in self.state_grid)

    def _trace_box_now(self, t_k=None):
        """The control_box callback traced for ONE call (trace.trace_box): a TracedBox whose whole-grid evaluation is,
        node for node, what the reference's scalar calls return (stodynprog.py:438-440) -- or a TraceError, and the
        table is then made by those scalar calls themselves.  Traced afresh on every call, like dyn and cost: data the
        callback reads (a rated power in a closure, a module-level capacity) are constants of the DAG, so data that
        changed show up as a changed signature.  A time index (t_k not None) is traced as a symbol, or failing that as
        the concrete step (`data[k]`)."""
        sd = self.sys
        nx, nu = len(sd.state), len(sd.control)
        try:
            tb = trace_box(sd.control_box, nx, nu, sd.params, stationnary=t_k is None)
        except TraceError as e:
            if t_k is None:
                return e
            try:
                tb = trace_box(sd.control_box, nx, nu, sd.params, stationnary=False, t_value=t_k)
            except TraceError:
                return e
        bad = tb.inexact_ops()
        if bad:
            return TraceError('control_box uses operations whose whole-grid evaluation need not repeat the scalar call '
                              'bit for bit: {}'.format(', '.join(bad)))
        return tb

    def _box_from_trace(self, tb, t_k=None):
        """(lo, hi) of shape (nu, S) -- (nu, 1) when the box ignores the state -- from a traced box, on the open grid"""
        shape = self._shape()
        d = len(shape)
        S = int(np.prod(shape))
        nu = len(self.sys.control)
        open_grid = [np.asarray(g, dtype=float).reshape((1,) * k + (-1,) + (1,) * (d - k - 1))
                     for k, g in enumerate(self.state_grid)]
        ends = tb.evaluate(open_grid, t_k)
        constant = all(np.ndim(a) == 0 and np.ndim(b) == 0 for a, b in ends)
        cols = 1 if constant else S
        lo_v = np.empty((nu, cols))
        hi_v = np.empty((nu, cols))
        for c, (a, b) in enumerate(ends):
            if constant:
                lo_v[c, 0], hi_v[c, 0] = a, b
            else:
                lo_v[c] = np.broadcast_to(a, shape).ravel()
                hi_v[c] = np.broadcast_to(b, shape).ravel()
        return lo_v, hi_v

    def _box_table(self, t_k=None, traced=None):
        """control_grids() for every node at once: (lo, hi, n) arrays of shape (nu, S) -- (nu, 1) when the box ignores
        the state -- with the same numpy operators as control_grids (sdp.py:432-463).

        The ends of the boxes come from the callback's TRACE evaluated on the whole grid (`_trace_box_now`): per node
        the same IEEE operations as the reference's scalar call, so the table is exact at every node by construction
        -- no sampling.  (As a check of the tracer itself the corners and the centre of the grid are compared with
        scalar calls whenever a table is built.)  A callback that cannot be traced is called node by node, as the
        reference does."""
        shape = self._shape()
        S = int(np.prod(shape))
        nu = len(self.sys.control)
        params = self.sys.params
        lead = () if t_k is None else (t_k,)
        tb = self._trace_box_now(t_k) if traced is None else traced
        self._box_mode = None           # how the table was made: 'traced', or None (node by node)
        self._box_sig = None
        lo = hi = None
        if not isinstance(tb, TraceError):
            lo, hi = self._box_from_trace(tb, t_k)
            anchors = sorted({int(np.ravel_multi_index(ind, shape))
                              for ind in itertools.product(*[(0, n - 1) for n in shape])} | {S // 2})
            for flat in anchors:
                ind = np.unravel_index(flat, shape)
                x = tuple(g[i] for g, i in zip(self.state_grid, ind))
                col = flat if lo.shape[1] > 1 else 0
                for c, (a, b) in enumerate(self.sys.control_box(*(lead + x), **params)):
                    if not (_same(lo[c, col], a) and _same(hi[c, col], b)):
                        raise AssertionError('the traced control_box differs from the callback at node {}: the callback is '
                                             'not a pure function of its arguments, or the tracer is wrong'.format(ind))
            self._box_mode = 'traced'
            self._box_sig = tb.signature()
        else:
            lo = np.empty((nu, S))
            hi = np.empty((nu, S))
            for flat, x in enumerate(itertools.product(*self.state_grid)):
                box = self.sys.control_box(*(lead + x), **params)
                for c, (a, b) in enumerate(box):
                    lo[c, flat] = a
                    hi[c, flat] = b
        return self._box_lattice(lo, hi)

    def _box_lattice(self, lo, hi):
        """(lo, hi, n) of the control lattice from the ends of the boxes, with control_grids' operators (sdp.py:446-453)"""
        nu = len(self.sys.control)
        n = np.empty(lo.shape, dtype=np.int32)
        with np.errstate(all='ignore'):
            for c in range(nu):
                step = self.control_steps[c]
                n_interv = (hi[c] - lo[c]) / step                   # sdp.py:446-447
                single = n_interv < 0.1                              # sdp.py:449
                npts = np.where(single, 1, np.ceil(np.where(single, 0., n_interv)) + 1)
                n[c] = npts.astype(np.int32)
                mid = (lo[c] + hi[c]) / 2                            # sdp.py:453
                lo[c] = np.where(single, mid, lo[c])
                hi[c] = np.where(single, mid, hi[c])
        return lo, hi, n

    def _fingerprint(self, t_k):
        s = self.sys
        # the callables themselves (hashed by identity and kept alive by the cache
        # key, so a recycled id() can never alias a stale entry)
        parts = [s.dyn, s.cost, s.control_box, _params_key(s.params),
                 tuple(float(x) for x in self.control_steps), str(self.dtype), t_k,
                 id(self.comm), self.comm_phases, self.comm_taper, self.comm_exchange,
                 getattr(self, 'comm_sparse', False), self.kernel,
                 getattr(self, 'certified_filter', True),
                 tuple(sorted((codegen.check_debug(self.debug_defines) or {}).items()))]
        for g in list(self.state_grid) + list(self.perturb_grid) + list(self.perturb_proba):
            parts.append(np.asarray(g, dtype=float).tobytes())
        # the tuple itself is the cache key (not its hash): the callables stay alive as
        # long as the entry does, so a recycled id() can never alias a stale entry
        return tuple(parts)

    def _traced(self):
        key = ('trace', self.sys.dyn, self.sys.cost, repr(sorted(self.sys.params.items())))
        if key not in self._cache:
            s = self.sys
            try:
                model = trace_model(s.dyn, s.cost, len(s.state), len(s.control),
                                    len(s.perturb), s.params, s.stationnary)
            except TraceError as e:
                model = e
            self._cache[key] = model
        return self._cache[key]

    def _trace_now(self, t_k=None):
        """Trace the callables for ONE call (value_iteration, a step of
        bellman_recursion, eval_policy).  Traced afresh every time, like the
        reference evaluates the callables at call time (stodynprog.py:674-676):
        module-level data they read may have changed since the last call.

        A time-dependent system whose callables need a concrete time index
        (`data[k]`) cannot be traced with a symbolic k; it is traced for the
        step `t_k` alone and the constants of the step are lifted into kernel
        parameters (trace.TracedModel.lift_constants), so the steps of a
        horizon share one code object.  Returns a TracedModel or a TraceError."""
        s = self.sys
        # (round 6: a callable whose FINGERPRINT -- code, defaults, closure cells and the globals it names, by value -- is
        # what it was at the last call traces to the same graph: the trace is kept.  No fingerprint (an object, a large
        # array, a module of the user's in sight): traced afresh, as before.  trace.callable_fingerprint)
        fp = callable_fingerprint(s.dyn, s.cost, s.params) if self.trace_cache else None
        if fp is not None:
            kept = self._cache.get(('trace now', s.stationnary, None if s.stationnary else t_k))
            if kept is not None and kept[0] == fp:
                return kept[1]
        model = self._trace_now_uncached(t_k)
        if fp is not None and (s.stationnary or getattr(model, 't_value', None) is None):
            self._cache[('trace now', s.stationnary, None if s.stationnary else t_k)] = (fp, model)
        return model

    def _trace_now_uncached(self, t_k=None):
        s = self.sys
        try:
            model = trace_model(s.dyn, s.cost, len(s.state), len(s.control), len(s.perturb),
                                s.params, s.stationnary)
            # Constants stay literals in the generated source (the compiler folds
            # them) until the SAME expression structure shows up with other values
            # -- a parameter study looping over a cost coefficient, say; from then on
            # they are lifted too, so the study compiles at most two code objects.
            key = ('struct', model.structure_key())
            bits = np.asarray(model.param_values(), dtype=float).tobytes()
            seen = self._cache.setdefault(key, dict(bits=bits, lifted=False))
            if not seen['lifted'] and seen['bits'] != bits:
                seen['lifted'] = True
            if seen['lifted']:
                model.lift_constants()
            return model
        except TraceError as e:
            if s.stationnary or t_k is None:
                return e
            try:
                model = trace_model(s.dyn, s.cost, len(s.state), len(s.control),
                                    len(s.perturb), s.params, s.stationnary, t_value=int(t_k))
            except TraceError:
                return e
            model.lift_constants()
            return model

    def _check_supported(self):
        if len(self.perturb_grid) > 1:
            raise NotImplementedError('only one perturbation variable is supported '
                                      '(as in the reference, sdp.py:664-666)')
        d = len(self.state_grid)
        if d > 5:
            raise NotImplementedError('interpolation for state dimension >5'
                                      ' is not implemented.')
        if d > 4:
            raise Exception("Can't interpolate in dimension strictly greater than 5")

    def _box_plan(self, box_t=None):
        """Control-box table of the current discretisation, cached.

        The reference calls control_box at every node of every sweep (stodynprog.py:440), so data the callback reads (a
        rated power in a closure, a module-level capacity) may change between two calls.  The callback is therefore
        TRACED AGAIN on every call -- as dyn and cost are --: what it reads are constants of its DAG, and the cached
        table stays exactly as long as the new trace has the structure and the constants of the one the table was made
        from (`TracedBox.signature`): a proof for every node, not a probe of some.  Only a callback that cannot be traced
        (its table was made by scalar calls at every node) is re-checked by scalar calls at `n_probe` nodes -- the
        corners, the centre and a different sample every time -- which is all a cached table of an opaque callable can
        offer; `DPSolver.box_recheck = 'every node'` rebuilds such a table on every call instead, as the reference does."""
        key = ('box', self._fingerprint(box_t))
        bp = self._cache.get(key)
        # (a callback whose fingerprint is what it was when the table was made would trace to the same graph: no trace)
        fp = callable_fingerprint(self.sys.control_box, self.sys.params) if self.trace_cache else None
        if bp is not None and fp is not None and bp.get('fp') == fp and bp.get('sig') is not None:
            return bp
        tb = self._trace_box_now(box_t)
        if bp is not None:
            if not isinstance(tb, TraceError):
                if bp.get('sig') != tb.signature():
                    bp = None                   # the callback reads data that changed (or is no longer what it was): rebuild
                else:
                    bp['fp'] = fp               # (the same graph under another fingerprint: data it does not depend on)
            elif bp.get('sig') is not None or self.box_recheck == 'every node' or not self._box_still_valid(bp, box_t):
                bp = None
        if bp is None:
            lo, hi, n = self._box_table(box_t, traced=tb)
            per_node = not (np.all(lo == lo[:, :1]) and np.all(hi == hi[:, :1])
                            and np.all(n == n[:, :1]))
            max_u = int(np.prod(n.astype(np.int64), axis=0).max())
            if max_u >= 2 ** 31:
                raise ValueError('control lattice too large')
            if not per_node:
                lo, hi, n = lo[:, :1], hi[:, :1], n[:, :1]
            lo, hi, n = (np.ascontiguousarray(a) for a in (lo, hi, n))
            digest = hash((lo.tobytes(), hi.tobytes(), n.tobytes()))
            bp = dict(lo=lo, hi=hi, n=n, per_node=per_node, max_u=max_u,
                      lanes=codegen.lanes_for(max_u), digest=digest, mode=getattr(self, '_box_mode', None),
                      sig=getattr(self, '_box_sig', None), fp=fp)
            if box_t is not None:           # one table per time step: keep only the latest
                for k in [k for k in self._cache if k[0] == 'box']:
                    del self._cache[k]
            self._cache[key] = bp
        return bp

    def _box_still_valid(self, bp, box_t, n_probe=21):
        """A table made by scalar calls at every node (the callback could not be traced): the callback again at
        `n_probe` nodes plus the corners and the centre, compared with the table; any difference rebuilds it."""
        shape = self._shape()
        S = int(np.prod(shape))
        lo, hi, n = bp['lo'], bp['hi'], bp['n']
        self._box_probe_round = getattr(self, '_box_probe_round', 0) + 1
        rng = np.random.default_rng(self._box_probe_round)
        probe = set(rng.integers(0, S, size=min(S, n_probe)).tolist())
        probe.update([0, S - 1, S // 2])
        lead = () if box_t is None else (box_t,)
        try:
            for flat in probe:
                ind = np.unravel_index(flat, shape)
                x = tuple(g[i] for g, i in zip(self.state_grid, ind))
                box = self.sys.control_box(*(lead + x), **self.sys.params)
                col = flat if bp['per_node'] else 0
                for c, (a, b) in enumerate(box):
                    a, b = float(a), float(b)
                    with np.errstate(all='ignore'):
                        n_interv = (b - a) / self.control_steps[c]
                    if n_interv < 0.1:
                        a = b = (a + b) / 2
                        npts = 1
                    else:
                        npts = int(np.ceil(n_interv) + 1)
                    if not (_same(lo[c, col], a) and _same(hi[c, col], b) and n[c, col] == npts):
                        return False
        except Exception:
            return False
        return True

    def _kernel_plan(self, box_t=None, model=None):
        """Everything that determines the model code object of the current
        discretisation (no GPU needed): control-box table, lanes per node,
        kernel family, generated source.  `model`: the trace to plan for
        (default: the cached symbolic trace of `_traced`)."""
        if model is None:
            model = self._traced()
        if isinstance(model, TraceError):
            raise model
        shape = self._shape()
        dt = self.dtype
        bp = self._box_plan(box_t)
        lanes = bp['lanes']
        debug = codegen.check_debug(self.debug_defines)
        W = len(self.perturb_grid[0]) if self.perturb_grid else 0
        # The callables are traced afresh on every call (_trace_now), but a trace with the structure and the constants
        # of the last one plans -- and generates -- the same unit: the plan is kept (the source text of a call was a
        # fifth of a millisecond, as much as the kernels of the reference's own problem sizes).
        # (a trace that was kept -- trace_cache -- keeps what it contributes to the key, too)
        mk = getattr(model, '_plan_key', None)
        if mk is None or mk[0] != (model.param_index is not None):
            mk = (model.param_index is not None, model.structure_key(), np.asarray(model.param_values(), dtype=float).tobytes())
            try:
                model._plan_key = mk
            except AttributeError:
                pass
        memo = ('plan', self._fingerprint(None), bp['digest'], None if box_t is None else float(box_t),
                mk[1], mk[2], model.param_index is not None, model.t_value, bool(self._cache.get('no_lead')))
        kept = self._cache.get(memo)
        if kept is not None:
            return dict(kept, model=model)
        plan = self._kernel_plan_now(box_t, model, bp, lanes, debug, W)
        plan['_memo'] = memo
        for k in [k for k in self._cache if k[0] == 'plan' and k[1:4] != memo[1:4]]:
            del self._cache[k]                  # (another discretisation / box table: its plans are never asked for again)
        self._cache[memo] = plan
        return plan

    def _kernel_plan_now(self, box_t, model, bp, lanes, debug, W):
        shape = self._shape()
        dt = self.dtype
        # storage-separable models on a grid whose (W x N0) table fits the LDS of
        # a CU run the column kernels, with per-node arrays stored axis-0-fastest
        if self.kernel not in ('auto', 'generic', 'column', 'staged', 'lead', 'line'):
            raise ValueError("kernel must be 'auto', 'column', 'lead', 'line', 'staged' or 'generic'")
        may_filter = (getattr(self, 'certified_filter', True)
                      and codegen.column_filter_applies(model, dtype=dt, table=(shape[0], W, len(shape)), debug=debug))
        # The shape of the full-table column kernel is planned ONCE, with everything that sizes its LDS image
        # (the control table included), and handed to the code generator as it is.  A lattice that changes
        # with the time index gets a control table with room to spare (a capacity, checked by the library as
        # controls <= capacity), so that the steps of a horizon keep sharing one code object.
        n_controls = bp['max_u'] if box_t is None else 1 << max(int(bp['max_u']) - 1, 0).bit_length()
        col_cfg, utab, wres = None, None, 0
        if self.kernel in ('auto', 'column') and model.storage_separable:
            wpair = codegen.use_wpair(model, dt, debug)
            shift = bool(may_filter and codegen.column_shift_applies(model, dt, debug=debug))
            fr = codegen.control_table_plan(model, dt, bp['per_node'], n_controls, debug) if may_filter else None
            for frontier in ((fr, None) if fr is not None else (None,)):
                kw = dict(max_controls=n_controls, n_columns=int(np.prod(shape[1:])), shift=shift,
                          utab_values=codegen.utab_reals(len(frontier), n_controls) if frontier else 0, debug=debug)
                col_cfg = codegen.column_config(shape[0], W, len(shape), dt, wpair, may_filter, **kw)
                if col_cfg is not None:                   # (else once more without the control table)
                    utab = (frontier, n_controls) if frontier is not None else None
                    # the table a chunk of perturbation points at a time, where that lets more workgroups share a CU
                    wres = codegen.column_resident_points(model, shape[0], W, len(shape), dt, may_filter, shift, wpair,
                                                          col_cfg[0], kw['utab_values'], debug)
                    if wres:
                        col_cfg = codegen.column_config(shape[0], W, len(shape), dt, wpair, may_filter, wres=wres, **kw)
                    break
        column = col_cfg is not None
        # several controlled state variables next to an exogenous process: the node-order sweep with the
        # certified filter on an array reduced over w (csrc/sdp_lead_kernel.h); one GPU for now
        lead_axes = 0
        if (not column and self.kernel in ('auto', 'lead') and W > 0 and not self._cache.get('no_lead')
                and (self.comm is None or self.comm.is_device)
                and getattr(self, 'certified_filter', True)):
            # (one stock whose table does not fit LDS too: measured 5.9 ms against 12.7 ms of the row-window
            # column kernel at 1024 x 128 x 128 x 64 x 32, tools/window_vs_lead.py)
            lead_axes = codegen.lead_filter_applies(
                model, dt, 1 if (self.kernel == 'lead' or model.storage_separable) else 2, debug)
        # the stocks need not be listed first (the order of the state variables is the user's, reference
        # stodynprog.py:119-131): the filter then works on a permuted view of the axes, the second pass keeps the
        # reference's own axis order
        lead_perm = None
        if (not column and not lead_axes and self.kernel in ('auto', 'lead') and W > 0 and not self._cache.get('no_lead')
                and (self.comm is None or self.comm.is_device)
                and getattr(self, 'certified_filter', True)):
            co = codegen.lead_order(model, dt, debug)
            if co is not None:
                lead_axes, lead_perm = co
        if self.kernel == 'lead' and not lead_axes:
            raise ValueError("kernel = 'lead' needs controlled state variables next to an exogenous process, "
                             'a perturbation that reaches only that process, 8-byte reals and the certified '
                             'filter (several GPUs: a device communicator)')
        if lead_axes:
            lanes = 1                                     # one lane per node, the control loop in-lane
        # trailing next states that depend on the control but not on x0: the nodes of a
        # column still share a table, control by control, provided they share their control
        # values (box independent of x0) -- csrc/sdp_column_kernel.h, SDP_TRAIL_HAS_U
        # (on a grid of fewer than PERCONTROL_MIN_NODES nodes the direct kernel is faster than a table per control: 16^3
        # 0.160 / 0.036 ms, 24^3 0.224 / 0.101 ms, 32^3 0.242 / 0.251 ms, 48^3 0.46 / 0.95 ms -- round 5)
        per_control = (not column and not lead_axes and self.kernel in ('auto', 'column') and model.column_shareable
                       and model.trail_depends_on_u
                       and (self.kernel == 'column' or int(np.prod(shape)) >= self.PERCONTROL_MIN_NODES)
                       and self._box_constant_along_axis0(bp, shape))
        per_control_cfg = None
        if per_control:
            per_control_cfg = codegen.column_percontrol_config(shape[0], W, len(shape), dt, debug)
            column = per_control = per_control_cfg is not None
        window = None
        if (not column and not per_control and not lead_axes and self.kernel in ('auto', 'column')
                and model.storage_separable):
            # the W x N0 table exceeds the LDS of a CU: tabulate a window of rows per
            # segment of the column (csrc/sdp_column_kernel.h, SDP_COL_ROWS)
            window = codegen.column_window_config(shape[0], W, len(shape), dt,
                                                  self._lead_reach_rows(model, bp, box_t))
            column = window is not None
        if (not column or per_control) and not lead_perm and self.kernel in ('auto', 'column'):
            k = model.separable_axis_hint()
            if k is not None and not self._cache.get('hinted'):
                self._cache['hinted'] = True
                import warnings
                warnings.warn('state variable "{}" is the only one driven by the control: listing it '
                              'FIRST in dyn/cost/control_box (and in discretize_state) lets the fast '
                              'column kernel run this model'.format(self.sys.state[k]))
        if self.kernel == 'column' and not column:
            raise ValueError('the column kernel needs a storage-separable model whose '
                             'table fits in LDS')
        staged = None
        # The staged kernel gives a node to a thread, the direct one to `lanes` of them: below ~1e5 nodes the staged tiles
        # do not fill the chip and the value array sits in the caches anyway.  Measured in round 5 (staged / direct):
        # 1-D inventory 600 nodes x 257 controls x 16 w 0.95 / 0.023 ms, 4096 x 1025 x 32 6.9 / 0.23 ms, 65 536 x 1025 x 16
        # 7.8 / 3.3 ms, 262 144 x 257 x 16 5.0 / 4.2 ms; control-coupled 3-D 16^3 0.85 / 0.037 ms, 32^3 0.87 / 0.25 ms,
        # 40^3 0.88 / 0.46 ms, 48^3 0.92 / 0.88 ms, 64^3 1.42 / 2.05 ms; coupled 2-D, 65 controls x 9 w: 128^2 0.196 / 0.069 ms,
        # 256^2 0.218 / 0.288 ms, 512^2 0.34 / 1.26 ms; the same with 1025 controls: 256^2 3.04 / 1.46 ms, 512^2 4.5 / 5.9 ms.
        # So the direct kernel for one state variable, for grids of at most STAGED_MIN_NODES nodes, and up to four times
        # that where a node has STAGED_MIN_WORK control x perturbation points or more (few threads with long loops).
        S_nodes = int(np.prod(shape))
        small = (len(shape) == 1 or S_nodes <= self.STAGED_MIN_NODES
                 or (S_nodes <= 4 * self.STAGED_MIN_NODES and bp['max_u'] * max(W, 1) >= self.STAGED_MIN_WORK))
        if not column and not lead_axes and (self.kernel == 'staged' or (self.kernel == 'auto' and not small)):
            key = ('staged', model.structure_key(), bp['digest'], str(dt), shape, W, _dbg_key(debug))
            staged = self._cache.get(key)
            if staged is None:
                staged = codegen.staged_config(model, self.state_grid, self.perturb_grid, bp, dt,
                                               0.0 if box_t is None else float(box_t), debug=debug)
                self._cache[key] = staged
        # ONE state variable with the perturbation in its sums (`x + u - w`): the certified filter on the shifted lattice with
        # the value array itself as the table (csrc/sdp_line_kernel.h; one GPU).  Two launches per sweep, so only where
        # there is work to save: LINE_MIN_CELLS lattice cells per sweep (below that the direct kernel takes microseconds).
        line = 0
        if (len(shape) == 1 and self.kernel in ('auto', 'line') and self.comm is None and W > 0 and staged is None
                and getattr(self, 'certified_filter', True) and box_t is None
                and model.t_value is None and model.param_index is None
                and (self.kernel == 'line' or S_nodes * int(bp['max_u']) * W >= self.LINE_MIN_CELLS)
                and codegen.line_filter_applies(model, dt, W, debug)):
            line = W
            # lanes = slices of the control lattice per wave (csrc/sdp_line_kernel.h: a workgroup's four waves share a tile of
            # 64 / lanes consecutive nodes): enough of them to fill the chip (8192 waves), at least 8 controls per slice,
            # and a tile's (node, perturbation point) items within the kernel's LDS table of terms (2048)
            pow2 = lambda v: 1 << max(int(np.ceil(np.log2(max(v, 1)))), 0)
            need = pow2(W * 64 / 2048.)
            want = min(pow2(131072. / S_nodes), max(1 << int(np.floor(np.log2(max(bp['max_u'] / 32., 1)))), 1))
            lanes = int(min(64, max(need, want)))
        if self.kernel == 'line' and not line:
            raise ValueError("kernel = 'line' needs one state variable whose perturbation enters x' through final sums "
                             "(x + u - w), a cost that does not see it, 8-byte reals, a stationary system, the certified "
                             'filter, one GPU')
        filtered = bool(column and getattr(self, 'certified_filter', True) and codegen.column_filter_applies(
            model, window, per_control_cfg if per_control else None, dtype=dt,
            table=(shape[0], W, len(shape)), debug=debug))
        if not filtered or window is not None or per_control:
            utab = None
        source = codegen.translation_unit(model, dt, lanes,
                                          column=(shape[0], W, n_controls, int(np.prod(shape[1:]))) if column else None,
                                          staged=staged,
                                          window=window, per_control=per_control_cfg if per_control else None,
                                          filtered=filtered, utab=utab, lead_axes=lead_axes,
                                          col_cfg=col_cfg, debug=debug, wres=wres if filtered else 0,
                                          lead_perm=lead_perm, line=line)
        filtered = filtered or bool(lead_axes) or bool(line)
        return dict(model=model, source=source, column=column, lanes=lanes, staged=staged, filtered=filtered,
                    lead_axes=lead_axes, lead_perm=lead_perm, line=bool(line),
                    window=window, per_control=per_control,
                    col_seg_nodes=(window[3] if window else (per_control_cfg[0] if per_control else 0)),
                    per_node=bp['per_node'], lo=bp['lo'], hi=bp['hi'], n=bp['n'],
                    max_u=bp['max_u'], W=W, box_digest=bp['digest'], box_mode=bp.get('mode'))

    @staticmethod
    def _box_constant_along_axis0(bp, shape):
        """do all nodes of every column along axis 0 have the same control box?"""
        if not bp['per_node']:
            return True
        n0 = shape[0]
        for arr in (bp['lo'], bp['hi'], bp['n']):
            a = arr.reshape(arr.shape[0], n0, -1)
            if not (a == a[:, :1, :]).all():
                return False
        return True

    def _lead_reach_rows(self, model, bp, box_t=None, n_samples=4096, around=False):
        """Rows of axis 0 the controls (and perturbation points) of ONE node span:
        max over sampled nodes of the distance, in grid rows, between the next
        values of the leading state variable at the ends of the node's control
        lattice -- the same prediction the windowed column kernel makes per unit."""
        from .trace import evaluate
        shape = self._shape()
        S = int(np.prod(shape))
        rng = np.random.default_rng(7)
        flat = np.unique(np.concatenate([rng.integers(0, S, size=min(S, n_samples)), [0, S - 1]]))
        idx = np.unravel_index(flat, shape)
        x = [np.asarray(g, dtype=float)[i] for g, i in zip(self.state_grid, idx)]
        col = flat if bp['per_node'] else np.zeros_like(flat)
        nu = len(self.sys.control)
        ends = []
        for first in (True, False):
            ends.append([bp['lo'][c, col] if first else bp['hi'][c, col] for c in range(nu)])
        wg = np.asarray(self.perturb_grid[0], dtype=float) if (self.perturb_grid and model.n_perturb) else None
        ws = [None] if wg is None else ([wg[0], wg[-1]] if model.lead_depends_on_w else [wg[0]])
        g0 = np.asarray(self.state_grid[0], dtype=float)
        rows = []
        for u in ends:
            for w in ws:
                with np.errstate(all='ignore'):
                    xn, _ = evaluate(model, x, u, [] if w is None else [w],
                                     0.0 if box_t is None else float(box_t))
                p = (np.asarray(xn[0], dtype=float) - g0[0]) / (g0[-1] - g0[0]) * (len(g0) - 1)
                rows.append(np.clip(np.nan_to_num(p, nan=0.0, posinf=1e9, neginf=-1e9), 0, len(g0) - 2))
        rows = np.array([np.broadcast_to(r, flat.shape) for r in rows])
        if around:          # farthest row from the node's own, either side
            return int(np.ceil(np.abs(rows - idx[0][None, :]).max())) + 1
        return int(np.ceil((rows.max(axis=0) - rows.min(axis=0)).max())) + 1

    def _problem(self, t_k=None, model=None):
        """Device problem for the current discretisation and callables.  The
        handle is reused from call to call (and from time step to time step)
        as long as the generated source and the control-box table are the
        same; lifted constants are re-sent on every call."""
        self._check_supported()
        box_t = None if self.sys.stationnary else t_k
        if model is None:
            model = self._trace_now(t_k)
        if isinstance(model, TraceError):
            raise model
        plan = self._kernel_plan(box_t, model)
        # (the key of the unit's source: a hash of the text and of the kernel headers -- once per plan, not per call)
        skey = plan.get('_source_key')
        if skey is None:
            skey = codegen.source_key(plan['source'])
            kept_plan = self._cache.get(plan.get('_memo'))
            if kept_plan is not None:
                kept_plan['_source_key'] = skey
        fp = ('problem', self._fingerprint(None), plan['box_digest'], skey)
        prob = self._cache.get(fp)
        if prob is None:
            try:
                prob = self._create_problem(fp, plan)
            except MemoryError:
                # the reduced-array sweep keeps two more arrays of the size of the grid (the reduced array and a
                # plane-major copy of the cost-to-go): where they no longer fit, the families that need no extra
                # memory take over (row window / table per control / staged tiles) -- slower, same bits
                if not plan.get('lead_axes') or self.kernel == 'lead':
                    raise
                import warnings
                warnings.warn('not enough device memory for the reduced-array sweep (2 extra arrays of the grid\'s '
                              'size): using the kernels that need none')
                self._cache['no_lead'] = True
                plan = self._kernel_plan(box_t, model)
                fp = ('problem', self._fingerprint(None), plan['box_digest'], codegen.source_key(plan['source']))
                prob = self._create_problem(fp, plan)
        if model.param_index is not None:
            prob.set_params(model.param_values())
        self.backend_info = dict(prob.info, time_specialized=model.t_value is not None,
                                 lifted_constants=len(model.param_index or ()),
                                 # how the table of admissible boxes was made: 'traced' (the callback's DAG on the whole grid: exact
                                 # at every node by construction) or 'node by node' (scalar calls, like the reference)
                                 box_mode=plan.get('box_mode') or 'node by node')
        return prob

    def _create_problem(self, fp, plan):
        nat.require_gpu()
        model, source, column, lanes = plan['model'], plan['source'], plan['column'], plan['lanes']
        per_node, lo, hi, n, max_u, W = (plan[k] for k in ('per_node', 'lo', 'hi', 'n', 'max_u', 'W'))
        shape = self._shape()
        S = int(np.prod(shape))
        dt = self.dtype
        layout = nat.LAYOUT_COLUMNS if column else nat.LAYOUT_NODES
        arrays = dict(
            axes=[np.ascontiguousarray(g, dtype=dt) for g in self.state_grid],
            box_lo=np.ascontiguousarray(lo, dtype=dt),
            box_hi=np.ascontiguousarray(hi, dtype=dt),
            box_n=np.ascontiguousarray(n, dtype=np.int32))
        if W:
            arrays['wgrid'] = np.ascontiguousarray(self.perturb_grid[0], dtype=dt)
            arrays['proba'] = np.ascontiguousarray(self.perturb_proba[0], dtype=dt)
        module = nat.compile_model(source)
        # slabs: whole hyperplanes of the outermost axis of the device layout
        dev_shape = (shape[1:] + shape[:1]) if column else shape
        if self.comm is not None and self.comm.is_device:
            # RCCL: the library shares out phases of the node range and overlaps
            # each phase's all-gather with the next phase's kernel
            from .dist import phase_partition, slab_partition
            unit = shape[0] if column else 1
            bounds = phase_partition(S // unit, unit, self.comm.nranks, self.comm_phases,
                                     self.comm_taper)
            sparse = ((getattr(self, 'comm_sparse', False) or self.comm_exchange == 'sendrecv')
                      and self.comm_exchange in ('peer', 'direct', 'sendrecv')
                      and self.comm.nranks > 1 and column and not plan['per_control']
                      and not plan['window'] and model.storage_separable and self.sys.stationnary
                      and len(shape) >= 2)
            if sparse:
                dense_bounds = bounds
                bounds = slab_partition(S // unit, unit, self.comm.nranks, self.comm_phases)
            elif plan.get('lead_axes'):
                # reduced-array sweep: a rank reduces what its nodes read -- its own rows of the first
                # stock and a few around them -- so it gets ONE slab of whole rows
                row = int(np.prod(shape[1:])) if shape[0] >= self.comm.nranks else 1
                bounds = slab_partition(S // row, row, self.comm.nranks, self.comm_phases)
            node_range = (0, S)
        elif self.comm is not None:
            bounds = self.comm.slab_bounds(dev_shape)
            node_range = (int(bounds[self.comm.rank]), int(bounds[self.comm.rank + 1]))
        else:
            bounds, node_range = None, (0, S)
        # drop other cached problems: they hold large device buffers (indices of the last
        # sweep that nobody has looked at yet are fetched first)
        if self._idx_source is not None:
            self.last_policy_index
        old = [k for k in self._cache if k[0] == 'problem']
        if old and self.comm is not None and self.comm.is_device and self.comm.nranks > 1:
            # peer exchange: the other ranks may have this rank's old buffers mapped (HIP IPC).  Every
            # rank first closes ITS mappings, then all meet, and only then are the buffers freed -- a
            # freed buffer that is still mapped elsewhere poisons the export of the next one
            # allocated at its address (hipIpcGetMemHandle: invalid argument, seen with 8 ranks).
            for k in old:
                if self._cache[k].h:
                    nat.check(nat.lib().sdp_problem_attach_comm(self._cache[k].h, None, 0, None))
            self.comm.barrier()
        for k in old:
            self._cache.pop(k).close()
        prob = _DeviceProblem(arrays, module, dt, shape, len(self.sys.control), W, lanes,
                              per_node, node_range,
                              self.comm if (self.comm is not None and self.comm.is_device) else None,
                              bounds, layout, plan['staged'], plan['col_seg_nodes'])
        self._cache[fp] = prob
        if self._debug_after_create is not None:        # (tools/host_phase_stress.py: poison the fresh buffers)
            self._debug_after_create(prob)
        exchange = None
        if self.comm is not None and self.comm.is_device and self.comm.nranks > 1:
            exchange = 'rccl'
            if self.comm_exchange in ('peer', 'direct'):
                try:                                        # collective: all ranks succeed or none
                    nat.check(nat.lib().sdp_problem_enable_peer_exchange(prob.h))
                    exchange = 'peer'
                except RuntimeError as e:
                    # only the outcome the ranks AGREED on is a reason to fall back (all of them
                    # do); anything else is this rank's problem alone and must not be papered over
                    if 'peer exchange not available' not in str(e):
                        raise
                    import warnings
                    warnings.warn('peer exchange unavailable, using the RCCL all-gather: {}'.format(e))
                    prob.peer_failure = str(e)
                    if sparse:                              # back to interleaved phases for the gathers
                        prob.parts = np.ascontiguousarray(dense_bounds, dtype=np.int64)
                        nat.check(nat.lib().sdp_problem_attach_comm(prob.h, self.comm.handle,
                                                                    int(prob.parts.shape[0]), nat.ptr(prob.parts)))
                if exchange == 'peer' and sparse:
                    off, ranges = self._peer_needs(model, prob.parts, shape)
                    nat.check(nat.lib().sdp_problem_set_peer_needs(prob.h, nat.ptr(off), nat.ptr(ranges)))
                    exchange = 'peer-sparse'
                    prob.need_fraction = float((ranges[:, 1] - ranges[:, 0]).sum()) / S / self.comm.nranks
                if exchange.startswith('peer') and self.comm_exchange == 'direct':
                    if self.comm.nranks <= 8:
                        nat.check(nat.lib().sdp_problem_set_direct_exchange(prob.h, 1))
                        exchange = exchange.replace('peer', 'direct')
                    else:
                        import warnings
                        warnings.warn('the direct exchange serves the (at most 8) GPUs of one node: peer copies instead')
            elif self.comm_exchange == 'sendrecv':
                # the sparse exchange through the collective library alone: grouped ncclSend / ncclRecv of the bounding
                # range of what each rank reads (csrc/sdp_hip.hip, sendrecv_phase); where the model gives no need lists
                # it is the RCCL all-gather
                if sparse:
                    nat.check(nat.lib().sdp_problem_set_sendrecv_exchange(prob.h, 1))
                    off, ranges = self._peer_needs(model, prob.parts, shape)
                    nat.check(nat.lib().sdp_problem_set_peer_needs(prob.h, nat.ptr(off), nat.ptr(ranges)))
                    exchange = 'sendrecv'
                    prob.need_fraction = float((ranges[:, 1] - ranges[:, 0]).sum()) / S / self.comm.nranks
            elif self.comm_exchange != 'rccl':
                raise ValueError("comm_exchange must be 'rccl', 'sendrecv', 'peer' or 'direct'")
        if plan.get('lead_axes') and self.comm is not None and self.comm.is_device and self.comm.nranks > 1:
            # rows of the first stock a node's controls reach (sampled: the kernel notices a node that
            # reaches further and evaluates it from the value array itself)
            nat.check(nat.lib().sdp_problem_set_lead_halo(prob.h, self._lead_reach_rows(model, plan, None, around=True) + 1))
        prob.info = dict(mode='traced', exchange=exchange,
                         kernel='column' if column else ('staged' if plan['staged'] else
                                                         ('lead' if plan.get('lead_axes') else ('line' if plan.get('line') else 'generic'))),
                         controlled_axes=int(plan.get('lead_axes') or (1 if column else 0)),
                         # state variables in the order the filter sees them (stocks first) when they are not listed first
                         controlled_order=(list(plan['lead_perm']) if plan.get('lead_perm') else None),
                         staged=plan['staged'],
                         row_window=(dict(rows=plan['window'][2], segment_nodes=plan['window'][3])
                                     if plan['window'] else None),
                         table_per_control=bool(plan['per_control']),
                         certified_filter=bool(plan.get('filtered')),
                         # 'shifted lattice': the perturbation reaches x0' through a final sum (SDP_COL_SHIFT)
                         filter_form=(None if not plan.get('filtered') else
                                      ('shifted lattice' if ('#define SDP_COL_SHIFT 1' in plan['source'] or plan.get('line')) else
                                       ('reduced array' if plan.get('lead_axes') else 'reduced table'))),
                         # x0' = a chain of sums in another nesting than ((a +- b) +- ..), x + (w - u): regrouped for the first pass
                         regrouped_sums=bool(plan.get('filtered') and '#define SDP_COL_SHIFT 1' in plan['source']
                                             and '#define SDP_COL_SHIFT_CHAIN 0' not in plan['source']),
                         module=module, lanes_per_node=lanes,
                         max_controls=max_u, box_per_node=bool(per_node),
                         bit_exact_model=model.bit_exact,
                         inexact_ops=model.inexact_ops(),
                         # diagnostic switches this code object was built with (None in the product)
                         debug_defines=codegen.check_debug(self.debug_defines))
        return prob

    def _peer_needs(self, model, parts, shape):
        """For the sparse peer exchange: per rank the node ranges (device order, whole columns) of
        the cost-to-go array its backups READ -- the 2^(d-1) vertex columns of the cell each
        perturbation point sends each of its columns to, one more cell on every side (the device
        locates the cells itself; a next state on a cell boundary may round the other way there),
        and the column of the relative-DP reference node.  Returns (offsets[nranks+1], ranges[:, 2])."""
        from .trace import evaluate
        from .dist import intervals_of
        n0, trail = shape[0], shape[1:]
        nranks = parts.shape[1] - 1
        grids = [np.asarray(g, dtype=float) for g in self.state_grid]
        wg = (np.asarray(self.perturb_grid[0], dtype=float)
              if (self.perturb_grid and model.n_perturb) else None)
        ws = [None] if wg is None else list(wg)
        n_cols = int(np.prod(trail))
        idx = np.unravel_index(np.arange(n_cols), trail)
        xcols = [grids[k + 1][idx[k]] for k in range(len(trail))]
        owner = np.empty(n_cols, dtype=np.int64)
        for r in range(nranks):
            for ph in range(parts.shape[0]):
                owner[parts[ph, r] // n0:parts[ph, r + 1] // n0] = r
        nu = len(self.sys.control)
        u0 = [np.zeros(n_cols) for _ in range(nu)]
        x0 = np.full(n_cols, grids[0][0])
        cells = []
        for w in ws:
            with np.errstate(all='ignore'):
                xn, _ = evaluate(model, [x0] + xcols, u0, [] if w is None else [np.full(n_cols, w)], 0.0)
            q = []
            for k, g in enumerate(grids[1:]):
                nk = len(g)
                p = (np.broadcast_to(np.asarray(xn[k + 1], dtype=float), (n_cols,)) - g[0]) / (g[-1] - g[0]) * (nk - 1)
                p = np.nan_to_num(p, nan=0.0, posinf=0.0, neginf=0.0)
                p = np.where(np.abs(p) < 2147483648.0, p, 0.0)        # sdp_trunc_i32
                q.append(np.clip(np.trunc(p).astype(np.int64), 0, nk - 2))
            cells.append(q)
        ref_col = int(np.ravel_multi_index(self._state_ref_ind[1:], trail)) if len(trail) else 0
        offs, out = [0], []
        span = (-1, 0, 1, 2)
        for r in range(nranks):
            mine = owner == r
            need = np.zeros(trail, dtype=bool)
            for q in cells:
                for delta in np.ndindex(*([len(span)] * len(trail))):
                    at = tuple(np.clip(q[k][mine] + span[delta[k]], 0, trail[k] - 1) for k in range(len(trail)))
                    need[at] = True
            need = need.reshape(-1)
            need[ref_col] = True
            need[mine] = False                               # its own rows never travel
            iv = intervals_of(need) * n0
            out.append(iv)
            offs.append(offs[-1] + len(iv))
        ranges = (np.concatenate(out) if out else np.zeros((0, 2))).astype(np.int64).reshape(-1, 2)
        if not len(ranges):
            ranges = np.zeros((1, 2), dtype=np.int64)        # (a valid pointer for the C call)
        return np.ascontiguousarray(offs, dtype=np.int64), np.ascontiguousarray(ranges)

    def _ref_flat(self, prob=None):
        """flat C-order index of the relative-DP reference node (sdp.py:384)"""
        return int(np.ravel_multi_index(self._state_ref_ind, self._shape()))

    # ------------------------------------------------------------ value iteration
    def value_iteration(self, J_next, rel_dp=False, report_time=True):
        """solve one DP step on the entire state space grid, given the
        cost-to-go array `J_next` discretized over the state space grid
        (reference sdp.py:466-534).

        If rel_dp is True, J_next should be a (J_next, J_ref) tuple.

        Returns (J_k, pol_k); J_k is a tuple (J_diff, J_ref) if `rel_dp` is True.
        pol_k holds the optimal control VALUES, shape state_dims + (nb_control,).
        """
        t_start = datetime.now()
        ref_ind = self._state_ref_ind if rel_dp else None
        if rel_dp:
            J_next, J_ref = J_next
            # the cost-to-go must be a *differential* cost, zero at the reference state
            assert J_next[ref_ind] == 0.
        J_next = np.asarray(J_next)
        self._check_state_array(J_next)         # same ValueError as interp_on_state (sdp.py:412-415)
        if report_time:
            print('value iteration...', end='')
        J_k, pol_k, J_ref = self._backup(J_next, None, rel_dp)
        exec_time = (datetime.now() - t_start).total_seconds()
        if report_time:
            print('\rvalue iteration run in {:.2f} s'.format(exec_time))
        if rel_dp:
            J_k = J_k, J_ref
        return J_k, pol_k

    def value_iterations(self, J_next, n_iter, rel_dp=False, report_time=True, J_ref_full=False):
        """`n_iter` successive calls of `value_iteration`, each fed with the
        result of the previous one -- the loop every user of the reference
        writes by hand (doc/example_inventory.py:98-109, AR1 notebook) -- but
        with the cost-to-go kept on the device between sweeps: one upload, one
        download.  NOT in the reference API.  Same results, bit for bit, as

            for k in range(n_iter): J, pol = self.value_iteration(J, rel_dp)

        Returns (J_k, pol_k) of the last sweep; J_k is (J_diff, J_ref) with
        `rel_dp` (J_ref: last reference cost, or all of them if J_ref_full)."""
        t_start = datetime.now()
        assert n_iter >= 1
        if rel_dp:
            J_next, J_ref = J_next
            assert J_next[self._state_ref_ind] == 0.
        J_next = np.asarray(J_next)
        self._check_state_array(J_next)
        host_comm = self.comm is not None and not self.comm.is_device
        refs = np.zeros(n_iter)
        model = self._trace_now(None)
        if isinstance(model, TraceError) or host_comm:
            J_k = J_next                    # host callbacks / host exchange: sweep by sweep
            for k in range(n_iter):
                J_k, pol_k, r = self._backup(J_k, None, rel_dp)
                refs[k] = 0.0 if r is None else r
        else:
            prob = self._problem(None, model)
            prob.set_value(J_next)
            ref_flat = self._ref_flat(prob) if rel_dp else 0
            for k in range(n_iter):
                if k:
                    prob.swap()             # J of the previous sweep becomes J_next
                refs[k] = prob.sweep(0.0, rel_dp, ref_flat)
            J_k = prob.get_value()
            pol_k, self.last_policy_index = prob.get_policy()
        if report_time:
            exec_time = (datetime.now() - t_start).total_seconds()
            print('{:d} value iterations run in {:.2f} s'.format(n_iter, exec_time))
        if rel_dp:
            return (J_k, refs if J_ref_full else refs[-1]), pol_k
        return J_k, pol_k

    def _backup(self, J_next, t_k, rel_dp):
        """One sweep: fused kernel when the model is traceable, else tabulated."""
        model = self._trace_now(t_k)
        if isinstance(model, TraceError):
            return self._backup_tabulated(J_next, t_k, rel_dp)
        prob = self._problem(t_k, model)
        if self.comm is None:
            # single GPU: one library call, arrays through page-locked memory
            J_k, pol_k, J_ref = prob.backup_host(J_next, 0.0 if t_k is None else t_k, rel_dp,
                                                 self._ref_flat(prob) if rel_dp else 0,
                                                 overlap=self.host_overlap)
            self._idx_cache, self._idx_source = None, prob
            return J_k, pol_k, J_ref
        prob.set_value(J_next)
        # with a host-side (gloo) communicator the slabs are exchanged through
        # host memory after the sweep; with RCCL the library does it on device
        host_comm = self.comm is not None and not self.comm.is_device
        J_ref = prob.sweep(0.0 if t_k is None else t_k, rel_dp and not host_comm,
                           self._ref_flat(prob) if rel_dp else 0)
        J_k = prob.get_value()
        if host_comm:
            Jd = prob._to_device_order(J_k).reshape(-1)
            self.comm.all_gather_slabs(Jd, self.comm.slab_bounds(prob.dev_shape))
            J_k = prob._from_device_order(Jd)
            if rel_dp:
                J_ref = J_k[self._state_ref_ind]              # sdp.py:523-525
                J_k -= J_ref
        pol_k, idx = prob.get_policy()
        if host_comm:
            # every rank returns the complete policy, like the single-process call
            nu = len(self.sys.control)
            bounds = self.comm.slab_bounds(prob.dev_shape)
            pd = prob._to_device_order(pol_k, (nu,)).reshape(-1)
            self.comm.all_gather_slabs(pd, bounds * nu)
            pol_k = prob._from_device_order(pd, (nu,))
            idd = prob._to_device_order(idx, (), np.asarray(idx).dtype).reshape(-1)
            self.comm.all_gather_slabs(idd, bounds)
            idx = prob._from_device_order(idd)
        self.last_policy_index = idx
        return J_k, pol_k, J_ref

    def _backup_tabulated(self, J_next, t_k, rel_dp, chunk_cells=4_000_000):
        """Tabulated mode: callbacks evaluated on the host node by node exactly
        like reference sdp.py:651-676; gather + expectation + argmin on the
        device (sdp_tab_backup)."""
        nat.require_gpu()
        self._check_supported()
        shape = self._shape()
        S = int(np.prod(shape))
        nu = len(self.sys.control)
        W = len(self.perturb_grid[0]) if self.perturb_grid else 0
        d = len(shape)
        smin = np.array([g[0] for g in self.state_grid], dtype=float)
        smax = np.array([g[-1] for g in self.state_grid], dtype=float)
        orders = np.array(shape, dtype=np.int64)
        V = np.ascontiguousarray(J_next, dtype=float)
        h = C.c_void_p()
        nat.check(nat.lib().sdp_tab_create(d, nat.ptr(smin), nat.ptr(smax), nat.ptr(orders),
                                           nat.ptr(V), C.byref(h)))
        self.backend_info = dict(mode='tabulated', reason=str(self._trace_now(None)))
        J_k = np.zeros(S)
        idx_k = np.zeros(S, dtype=np.int64)
        pol_k = np.zeros((S, nu))
        proba = np.ascontiguousarray(self.perturb_proba[0], dtype=float) if W else None
        try:
            batch = []          # (flat, u_grids, dims, x_next[d, cells], g[cells])
            cells = 0

            def flush():
                nonlocal batch, cells
                if not batch:
                    return
                off = np.zeros(len(batch) + 1, dtype=np.int64)
                for i, b in enumerate(batch):
                    off[i + 1] = off[i] + b[4].size
                xn = np.ascontiguousarray(np.concatenate([b[3] for b in batch], axis=1))
                g = np.ascontiguousarray(np.concatenate([b[4] for b in batch]))
                Jb = np.zeros(len(batch))
                ib = np.zeros(len(batch), dtype=np.int64)
                nat.check(nat.lib().sdp_tab_backup(h, len(batch), nat.ptr(off), W,
                                                   nat.ptr(proba), nat.ptr(xn), nat.ptr(g),
                                                   nat.ptr(Jb), nat.ptr(ib)))
                for (flat, u_grids, dims, _, _), Jv, iv in zip(batch, Jb, ib):
                    J_k[flat] = Jv
                    idx_k[flat] = iv
                    ind = np.unravel_index(iv, dims)
                    pol_k[flat] = [u_grids[c].ravel()[ind[c]] for c in range(nu)]
                batch, cells = [], 0

            # one rule in every path (traced or not): the callables of a STATIONARY system
            # are never handed a time index.  (The reference's bellman_recursion passes t_k
            # regardless, sdp.py:582, and so raises TypeError on a stationary system.)
            if self.sys.stationnary:
                t_k = None
            for flat, x_k in enumerate(itertools.product(*self.state_grid)):
                u_grids, dims = self.control_grids(x_k, t_k)
                lattice = dims + ((W,) if W else ())
                for i in range(nu):
                    u_grids[i] = u_grids[i].reshape((1,) * i + (-1,) + (1,) * (nu - i))
                args = tuple(x_k) + tuple(u_grids) + tuple(self.perturb_grid)
                if t_k is not None:
                    args = (t_k,) + args
                x_next = self.sys.dyn(*args, **self.sys.params)
                g_grid = self.sys.cost(*args, **self.sys.params)
                xn = np.vstack([np.broadcast_to(np.asarray(x, dtype=float), lattice).ravel()
                                for x in x_next])
                gg = np.broadcast_to(np.asarray(g_grid, dtype=float), lattice).ravel()
                batch.append((flat, u_grids, dims, xn, gg))
                cells += gg.size
                if cells >= chunk_cells:
                    flush()
            flush()
        finally:
            nat.lib().sdp_tab_destroy(h)
        J_k = J_k.reshape(shape)
        J_ref = 0.0
        if rel_dp:
            J_ref = J_k[self._state_ref_ind]
            J_k -= J_ref
        self.last_policy_index = idx_k.reshape(shape).astype(np.int32)
        return J_k, pol_k.reshape(shape + (nu,)), J_ref

    def bellman_recursion(self, t_fin, J_fin, t_ini=0, report_time=True):
        """solve the Bellman backward recursion of a *finite horizon problem*
        from `t_fin` (positive int) down to `t_ini` (reference sdp.py:536-591).
        Supports non-stationnary problems.  Returns (J, pol) with a leading
        time axis."""
        t_start = datetime.now()
        state_dims = tuple(len(grid) for grid in self.state_grid)
        nb_control = len(self.sys.control)
        stationnary = self.sys.stationnary
        print('time-dependent problem: {:s}'.format('no' if stationnary else 'yes'))
        assert t_ini == 0       # t_ini > 0 not tested (as in the reference)
        J = np.zeros((t_fin - t_ini,) + state_dims)
        pol = np.zeros((t_fin - t_ini,) + state_dims + (nb_control,))
        if report_time:
            print('bellman recursion...', end='')
        for t_k in range(t_ini, t_fin)[::-1]:
            print('\rtk = {:3d}...'.format(t_k), end='')
            k = t_k - t_ini
            J_next = J_fin if t_k == (t_fin - 1) else J[k + 1]
            self._check_state_array(np.asarray(J_next))
            # the callables get t_k when the system is time dependent (see _backup_tabulated)
            J[k], pol[k], _ = self._backup(np.asarray(J_next), t_k, False)
        exec_time = (datetime.now() - t_start).total_seconds()
        if report_time:
            print('\rvalue iteration run in {:.2f} s'.format(exec_time))
        return J, pol

    # ----------------------------------------------------- per-node entry points
    def _node_lattice(self, x_k, t_k):
        u_grids, dims = self.control_grids(x_k, t_k)
        nu = len(u_grids)
        for i in range(nu):
            u_grids[i] = u_grids[i].reshape((1,) * i + (-1,) + (1,) * (nu - i))
        args = tuple(x_k) + tuple(u_grids) + tuple(self.perturb_grid)
        if t_k is not None:
            args = (t_k,) + args
        return u_grids, dims, args

    def _value_at_state_vect(self, x_k, J_next_interp, t_k=None):
        """optimal cost and control at one state point `x_k` (reference
        sdp.py:639-691): callbacks on the host, gather / expectation / argmin
        on the device.  Returns (J_xk_opt, u_xk_opt)."""
        u_grids, dims, args = self._node_lattice(x_k, t_k)
        nu = len(u_grids)
        W = len(self.perturb_grid[0]) if self.perturb_grid else 0
        lattice = dims + ((W,) if W else ())
        x_next = self.sys.dyn(*args, **self.sys.params)
        g_grid = self.sys.cost(*args, **self.sys.params)
        xn = np.ascontiguousarray(np.vstack(
            [np.broadcast_to(np.asarray(x, dtype=float), lattice).ravel() for x in x_next]))
        gg = np.ascontiguousarray(np.broadcast_to(np.asarray(g_grid, dtype=float), lattice).ravel())
        # the interpolator's values go to the device once (handle cached on the object,
        # dropped by set_values), not once per node
        from .interp import device_tab
        tab = device_tab(J_next_interp)
        off = np.array([0, gg.size], dtype=np.int64)
        Jb = np.zeros(1)
        ib = np.zeros(1, dtype=np.int64)
        proba = np.ascontiguousarray(self.perturb_proba[0], dtype=float) if W else None
        nat.check(nat.lib().sdp_tab_backup(tab.h, 1, nat.ptr(off), W, nat.ptr(proba),
                                           nat.ptr(xn), nat.ptr(gg), nat.ptr(Jb),
                                           nat.ptr(ib)))
        ind_opt = np.unravel_index(int(ib[0]), dims)
        u_opt = [u_grids[i].flatten()[ind_opt[i]] for i in range(nu)]
        return (Jb[0], u_opt)

    def _value_at_state_loop(self, x_k, J_next_interp):
        """same result as `_value_at_state_vect` (the reference's iterative
        variant, sdp.py:594-636, keeps the first minimum as well)."""
        J, u = self._value_at_state_vect(x_k, J_next_interp)
        return (J, tuple(u))

    # ------------------------------------------------------------ policy evaluation
    def eval_policy(self, pol, n_iter, rel_dp=False, J_zero=None,
                    report_time=True, J_ref_full=False):
        """evaluate the policy `pol`: cost of each state after `n_iter` steps
        (reference sdp.py:693-775).  If rel_dp is True the relative DP
        algorithm is used.

        Returns J_pol (array of shape self._state_grid_shape), or
        (J_pol, J_ref) if `rel_dp` is True (J_ref: the last reference cost,
        or all of them when J_ref_full).
        """
        t_start = datetime.now()
        state_dims = self._state_grid_shape
        if J_zero is None:
            J_zero = np.zeros(state_dims)
        assert J_zero.shape == state_dims
        nb_control = len(self.sys.control)
        assert pol.shape == state_dims + (nb_control,)
        t_pol = None if self.sys.stationnary else 0
        model = self._trace_now(t_pol)
        if isinstance(model, TraceError):
            return self._eval_policy_tabulated(pol, n_iter, rel_dp, J_zero, report_time,
                                               J_ref_full, t_start)
        prob = self._problem(t_pol, model)
        prob.set_value(J_zero)
        prob.set_policy(pol)
        for k in range(n_iter):
            # progress line of the reference; the iterations themselves run in one device call
            print('\rpolicy evaluation: iter. {:d}/{:d}'.format(k, n_iter), end='')
        if self.comm is not None and not self.comm.is_device:
            # host-side (gloo) communicator: one device call per iteration, the
            # slabs meet in host memory in between (test path; RCCL stays on device)
            bounds = self.comm.slab_bounds(prob.dev_shape)
            J_pol, J_ref = np.asarray(J_zero, dtype=self.dtype), np.zeros(n_iter)
            for k in range(n_iter):
                if k:
                    prob.set_value(J_pol)
                prob.eval_policy(1, False, 0)
                Jd = prob._to_device_order(prob.get_value()).reshape(-1)
                self.comm.all_gather_slabs(Jd, bounds)
                J_pol = prob._from_device_order(Jd)
                if rel_dp:
                    J_ref[k] = J_pol[self._state_ref_ind]          # sdp.py:757-760
                    J_pol -= J_pol[self._state_ref_ind]
        else:
            J_ref = prob.eval_policy(n_iter, rel_dp, self._ref_flat(prob) if rel_dp else 0)
            J_pol = prob.get_value()
        exec_time = (datetime.now() - t_start).total_seconds()
        if report_time:
            print('\rpolicy evaluation run in {:.2f} s     '.format(exec_time))
        if rel_dp:
            if not J_ref_full:
                J_ref = J_ref[-1]
            return J_pol, J_ref
        return J_pol

    def _eval_policy_tabulated(self, pol, n_iter, rel_dp, J_zero, report_time, J_ref_full,
                               t_start):
        """eval_policy for callables that cannot be traced: dyn and cost are
        evaluated on the host over the whole (S x W) lattice exactly as the
        reference does (sdp.py:732-754, once: they do not change between
        iterations), the interpolation + expectation of every iteration run on
        the device (sdp_tab_backup with one control per node)."""
        nat.require_gpu()
        self._check_supported()
        dims = self._state_grid_shape
        d = len(dims)
        S = int(np.prod(dims))
        nu = len(self.sys.control)
        w_k = self.perturb_grid[0]
        W = len(w_k)
        state_grid = tuple(np.reshape(self.state_grid[i], (1,) * i + (-1,) + (1,) * (d - i))
                           for i in range(d))
        u_k = [pol[..., i].reshape(dims + (1,)) for i in range(nu)]
        args = state_grid + tuple(u_k) + (w_k,)
        if not self.sys.stationnary:
            args = (0,) + args                  # like the traced path: time index 0
        x_next = self.sys.dyn(*args, **self.sys.params)
        g = self.sys.cost(*args, **self.sys.params)
        lattice = dims + (W,)
        xn = np.ascontiguousarray(np.vstack(
            [np.broadcast_to(np.asarray(x, dtype=float), lattice).ravel() for x in x_next]))
        gg = np.ascontiguousarray(np.broadcast_to(np.asarray(g, dtype=float), lattice).ravel())
        off = np.arange(S + 1, dtype=np.int64) * W
        proba = np.ascontiguousarray(self.perturb_proba[0], dtype=float)
        smin = np.array([gr[0] for gr in self.state_grid], dtype=float)
        smax = np.array([gr[-1] for gr in self.state_grid], dtype=float)
        orders = np.array(dims, dtype=np.int64)
        self.backend_info = dict(mode='tabulated', reason=str(self._trace_now(None)))
        J_pol = np.ascontiguousarray(J_zero, dtype=float)
        J_ref = np.zeros(n_iter)
        idx = np.zeros(S, dtype=np.int64)
        for k in range(n_iter):
            print('\rpolicy evaluation: iter. {:d}/{:d}'.format(k, n_iter), end='')
            h = C.c_void_p()
            nat.check(nat.lib().sdp_tab_create(d, nat.ptr(smin), nat.ptr(smax), nat.ptr(orders),
                                               nat.ptr(J_pol), C.byref(h)))
            try:
                J_new = np.zeros(S)
                nat.check(nat.lib().sdp_tab_backup(h, S, nat.ptr(off), W, nat.ptr(proba),
                                                   nat.ptr(xn), nat.ptr(gg), nat.ptr(J_new),
                                                   nat.ptr(idx)))
            finally:
                nat.lib().sdp_tab_destroy(h)
            J_pol = J_new.reshape(dims)
            if rel_dp:
                J_ref[k] = J_pol[self._state_ref_ind]               # sdp.py:760-762
                J_pol -= J_ref[k]
        exec_time = (datetime.now() - t_start).total_seconds()
        if report_time:
            print('\rpolicy evaluation run in {:.2f} s     '.format(exec_time))
        if rel_dp:
            return J_pol, (J_ref if J_ref_full else J_ref[-1])
        return J_pol

    def policy_iteration(self, pol_init, n_val, n_pol=1, rel_dp=False):
        """policy iteration algorithm (reference sdp.py:777-812).

        pol_init : initial policy to evaluate
        n_val : number of value iterations to evaluate the policy
        n_pol : number of policy iterations (default to 1)

        Returns (J_pol, pol); J_pol is a tuple (J_diff, J_ref) if rel_dp.
        """
        pol = pol_init
        J_pol = self.eval_policy(pol, n_val, rel_dp)
        if rel_dp:
            J_diff, J_ref = J_pol
            print('ref policy cost: {:g}'.format(J_ref))
        for k in range(n_pol):
            print('policy iteration {:d}/{:d}'.format(k + 1, n_pol))
            _, pol = self.value_iteration(J_pol, rel_dp=rel_dp)
            J_pol = self.eval_policy(pol, n_val, rel_dp)
            if rel_dp:
                J_ref = J_pol[1]
                print('ref policy cost: {:g}'.format(J_ref))
        return J_pol, pol

    # ------------------------------------------------------------ closed-loop simulation
    def simulate(self, pol, x0, w=None, n_steps=None, t0=0):
        """Closed-loop trajectories under the policy `pol` -- the loop every example of the
        reference writes by hand (examples/20 Searev storage control/
        storage_control.py:242-251):

            for k in range(T):
                u[k] = [self.interp_on_state(pol[..., c])(*x[k]) for c in range(nb_control)]
                x[k+1] = sys.dyn(*x[k], *u[k], w[k])

        run for a BATCH of trajectories on the GPU without a host round trip per step
        (kernel sdp_simulate: policy lookup by multilinear interpolation + the traced
        dynamics).  NOT in the reference API.

        pol : policy array on the state grid, shape state_dims + (nb_control,)
              (as returned by value_iteration / policy_iteration)
        x0  : start state(s), shape (nb_state,) or (B, nb_state)
        w   : perturbation sequence(s), shape (T,) or (T, B); None for a deterministic
              system (then give n_steps)
        t0  : time index of the first step (non-stationary systems)

        Returns (x, u, g): states (T+1, [B,] nb_state), controls (T, [B,] nb_control) and
        instantaneous costs (T[, B]).  Same bits as the hand-written loop when the
        model is `bit_exact` (backend_info)."""
        dims = self._state_grid_shape
        d, nu = len(dims), len(self.sys.control)
        pol = np.asarray(pol)
        assert pol.shape == dims + (nu,)
        x0 = np.asarray(x0, dtype=float)
        single = x0.ndim == 1
        x0 = np.atleast_2d(x0)
        assert x0.shape[1] == d
        B = x0.shape[0]
        n_w = len(self.sys.perturb)
        if n_w:
            assert w is not None, 'a stochastic system needs the perturbation sequence(s) w'
            w = np.asarray(w, dtype=float)
            w = w.reshape(-1, 1) if w.ndim == 1 else w
            assert w.shape[1] == B
            T = w.shape[0] if n_steps is None else int(n_steps)
            assert w.shape[0] >= T
        else:
            assert n_steps is not None, 'give n_steps for a deterministic system'
            T = int(n_steps)
        t_trace = None if self.sys.stationnary else t0
        model = self._trace_now(t_trace)
        if isinstance(model, TraceError) or (model.t_value is not None):
            x, u, g = self._simulate_host(pol, x0, w, T, t0)
        else:
            prob = self._problem(t_trace, model)
            dt = self.dtype
            pol_d = np.ascontiguousarray(np.moveaxis(pol, -1, 0), dtype=dt)        # [nu][S]
            x0_d = np.ascontiguousarray(x0.T, dtype=dt)                             # [d][B]
            w_d = np.ascontiguousarray(w[:T], dtype=dt) if n_w else None            # [T][B]
            x = np.empty((T + 1, d, B), dtype=dt)
            u = np.empty((T, nu, B), dtype=dt)
            g = np.empty((T, B), dtype=dt)
            nat.check(nat.lib().sdp_problem_simulate(prob.h, nat.ptr(pol_d), B, T, nat.ptr(x0_d),
                                                     nat.ptr(w_d), float(t0), nat.ptr(x), nat.ptr(u),
                                                     nat.ptr(g)))
            x, u = np.moveaxis(x, 1, 2), np.moveaxis(u, 1, 2)
        if single:
            return x[:, 0], u[:, 0], g[:, 0]
        return np.ascontiguousarray(x), np.ascontiguousarray(u), g

    def _simulate_host(self, pol, x0, w, T, t0):
        """the reference's loop as written (callables on the host, one interpolator call
        per step, batched over the trajectories): models that cannot be traced"""
        d, nu = len(self._state_grid_shape), len(self.sys.control)
        B = x0.shape[0]
        laws = [self.interp_on_state(np.ascontiguousarray(pol[..., c])) for c in range(nu)]
        x = np.zeros((T + 1, B, d))
        u = np.zeros((T, B, nu))
        g = np.zeros((T, B))
        x[0] = x0
        for k in range(T):
            xs = tuple(x[k, :, i] for i in range(d))
            for c in range(nu):
                u[k, :, c] = laws[c](*xs)
            args = xs + tuple(u[k, :, c] for c in range(nu)) + ((w[k],) if w is not None else ())
            if not self.sys.stationnary:
                args = (t0 + k,) + args
            xn = self.sys.dyn(*args, **self.sys.params)
            for i in range(d):
                x[k + 1, :, i] = xn[i]
            g[k] = self.sys.cost(*args, **self.sys.params)
        return x, u, g

    # ------------------------------------------------------------------ reporting
    def print_summary(self):
        """summary information about the state of the SDP solver
        (reference sdp.py:814-875)"""
        print('SDP solver for system "{}"'.format(self.sys.name))

        def describe(names, grids):
            for name, grid in zip(names, grids):
                if len(grid) > 1:
                    print('  - Δ{:s} = {:g}'.format(name, grid[1] - grid[0]))
                else:
                    print('  - {:s} fixed at {:g}'.format(name, grid[0]))

        size = 'x'.join(str(len(g)) for g in self.state_grid)
        print('* state space discretized on a {:s} points grid'.format(size))
        describe(self.sys.state, self.state_grid)
        if self.sys.stochastic:
            size = 'x'.join(str(len(g)) for g in self.perturb_grid)
            print('* perturbation discretized on a {:s} points grid'.format(size))
            describe(self.sys.perturb, self.perturb_grid)
        cdim = None
        if self.sys.control_box is not None:
            t_k = None if self.sys.stationnary else 0
            _, _, n = self._box_table(t_k)
            cdim = n.T.astype(np.int64)             # (S, nu)
        else:
            print('Warning: sys.control_box is still to be defined!')
        print('* control discretization steps:')
        for i in range(len(self.sys.control)):
            print('  - Δ{:s} = {:g}'.format(self.sys.control[i], self.control_steps[i]))
            if cdim is not None:
                lo_n, hi_n = cdim[:, i].min(), cdim[:, i].max()
                if lo_n != hi_n:
                    print(('    yields [{:,d} to {:,d}] possible values'
                           ' ({:,.1f} on average)').format(lo_n, hi_n, cdim[:, i].mean()))
                else:
                    print('    yields {:,d} possible values'.format(cdim[0, i]))
        if cdim is not None and len(self.sys.control) >= 2:
            tot = np.prod(cdim, axis=1)
            print('  control combinations:'
                  ' [{:,d} to {:,d}] possible values ({:,.1f} on average)'.format(
                      tot.min(), tot.max(), tot.mean()))


def _params_key(params):
    """hashable, untruncated image of a params dict (repr() shortens large arrays)"""
    out = []
    for k in sorted(params):
        v = params[k]
        if isinstance(v, np.ndarray):
            out.append((k, v.dtype.str, v.shape, v.tobytes()))
        else:
            try:
                hash(v)
                out.append((k, v))
            except TypeError:
                out.append((k, repr(v)))
    return tuple(out)


def _dbg_key(debug):
    return tuple(sorted(debug.items())) if debug else ()


def _same(a, b):
    a, b = float(a), float(b)
    return a == b or (a != a and b != b)
