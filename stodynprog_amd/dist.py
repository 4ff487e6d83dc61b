"""Multi-GPU sharding of the value-iteration sweep: one process per GPU.

The reference is single-process (its only attempt at parallelism is a
commented-out multiprocessing.Pool over state nodes, reference
stodynprog/stodynprog.py:503-509).  Every node's backup reads only J_next and
writes only its own J_k entry, so the C-ordered node range is cut into
contiguous slabs along the OUTER state axis, one per rank; J_next is
replicated on every GPU (256^3 fp64 = 134 MB of 288 GB) and after each sweep
the J_k slabs are exchanged with ONE all-gather:

  * RcclCommunicator  -- device buffers, RCCL over xGMI, called inside
    sdp_problem_vi_sweep on the sweep's stream (include/sdp_hip.h);
  * GlooCommunicator  -- host arrays over torch.distributed/gloo; used for the
    rendezvous of the RCCL unique id and by the CPU tests of the slab logic.

Policies are gathered only on request (sdp_problem_get_policy is then a collective call);
the timed sweep never pays for it.
"""
import os

import numpy as np

__all__ = ['slab_bounds', 'phase_partition', 'RcclCommunicator', 'GlooCommunicator', 'from_env']


def slab_bounds(shape, nranks):
    """Flat C-order node ranges [b[r], b[r+1]) of each rank: whole hyperplanes
    of the outer axis, as even as possible (the first N_0 % nranks ranks get
    one more plane)."""
    shape = tuple(int(n) for n in shape)
    n0 = shape[0]
    plane = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    base, extra = divmod(n0, nranks)
    bounds = np.zeros(nranks + 1, dtype=np.int64)
    for r in range(nranks):
        bounds[r + 1] = bounds[r] + (base + (1 if r < extra else 0)) * plane
    return bounds


def phase_partition(n_units, unit, nranks, n_phases, taper=False):
    """Node bounds [n_phases][nranks+1] for the overlapped multi-GPU backup:
    `n_units` work units of `unit` nodes each (columns of the column layout, or
    single nodes) are cut into `n_phases` contiguous phases and every phase
    into one contiguous part per rank, as even as possible.  Phases let the
    all-gather of one phase run under the kernel of the next; only the LAST
    phase's gather is exposed, so with `taper` the phases shrink linearly
    (weights n, n-1, .., 1: 40/30/20/10 % for four) and the exposed gather
    moves the smallest share."""
    n_phases = max(1, min(int(n_phases), max(1, n_units // max(nranks, 1))))
    weights = [n_phases - k if taper else 1 for k in range(n_phases)]
    total = sum(weights)
    cuts = [0]
    for k in range(n_phases):
        cuts.append(n_units * sum(weights[:k + 1]) // total)
    bounds = np.zeros((n_phases, nranks + 1), dtype=np.int64)
    for ph in range(n_phases):
        lo, hi = cuts[ph], cuts[ph + 1]
        for r in range(nranks + 1):
            bounds[ph, r] = (lo + (hi - lo) * r // nranks) * unit
    return bounds


class _Base(object):
    rank = 0
    nranks = 1
    is_device = False

    def slab_bounds(self, shape):
        return slab_bounds(shape, self.nranks)


class GlooCommunicator(_Base):
    """Host-side collectives over an initialised torch.distributed group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)

    def all_gather_slabs(self, J, bounds):
        """Fill the flat view of `J` (complete array, own slab valid) with the
        slabs of every rank, in place.  Slabs may have different lengths."""
        import torch
        flat = J.reshape(-1)
        mine = torch.from_numpy(np.ascontiguousarray(flat[bounds[self.rank]:bounds[self.rank + 1]]))
        for r in range(self.nranks):
            n = int(bounds[r + 1] - bounds[r])
            buf = mine.clone() if r == self.rank else torch.empty(n, dtype=mine.dtype)
            if n:
                self._dist.broadcast(buf, src=self._global_rank(r), group=self.group)
                flat[bounds[r]:bounds[r + 1]] = buf.numpy()
        return J

    def _global_rank(self, r):
        if self.group is None:
            return r
        return self._dist.get_global_rank(self.group, r)

    def broadcast_bytes(self, payload, src=0):
        obj = [payload if self.rank == src else None]
        self._dist.broadcast_object_list(obj, src=self._global_rank(src), group=self.group)
        return obj[0]

    def allreduce_max(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def barrier(self):
        self._dist.barrier(group=self.group)


class RcclCommunicator(_Base):
    """RCCL communicator owned by libsdp_hip.so (sdp_comm_* of include/sdp_hip.h)."""
    is_device = True

    def __init__(self, rank, nranks, unique_id):
        import ctypes as C
        from . import _native as nat
        self.rank, self.nranks = int(rank), int(nranks)
        runtimes = nat.rocm_runtimes()
        if len(runtimes) > 1:
            raise nat.NativeError(
                'two HIP runtimes are mapped in this process ({}): libsdp_hip.so was loaded '
                'before torch.  RCCL cannot initialise in that state; import torch before '
                'stodynprog_amd (a launcher that sets WORLD_SIZE does it for you)'
                .format(', '.join(runtimes)))
        h = C.c_void_p()
        with _stdout_to_stderr():          # RCCL prints a version banner on stdout at init
            nat.check(nat.lib().sdp_comm_create(self.rank, self.nranks, unique_id, C.byref(h)))
        self.handle = h
        self._nat = nat

    @staticmethod
    def new_unique_id():
        import ctypes as C
        from . import _native as nat
        buf = C.create_string_buffer(128)
        nat.check(nat.lib().sdp_comm_unique_id(buf))
        return buf.raw

    def allreduce_max(self, value):
        import ctypes as C
        v = C.c_double(float(value))
        self._nat.check(self._nat.lib().sdp_comm_allreduce_max(self.handle, C.byref(v)))
        return v.value

    def barrier(self):
        self._nat.check(self._nat.lib().sdp_comm_barrier(self.handle))

    def close(self):
        if getattr(self, 'handle', None):
            self._nat.lib().sdp_comm_destroy(self.handle)
            self.handle = None


def from_env():
    """Communicator of a process started by `python -m torch.distributed.run`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
    environment): selects GPU LOCAL_RANK, exchanges the RCCL unique id through
    a gloo group and returns (RcclCommunicator, GlooCommunicator).  With
    WORLD_SIZE absent or 1 returns (None, None)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None, None
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist        # before the library: one ROCm runtime (see _native)
    from . import _native as nat
    nat.check(nat.lib().sdp_set_device(local))
    if not dist.is_initialized():
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    host = GlooCommunicator()
    with _stdout_to_stderr():          # nothing native may write into the caller's stdout
        uid = RcclCommunicator.new_unique_id() if rank == 0 else None
        uid = host.broadcast_bytes(uid, src=0)
        dev = RcclCommunicator(rank, world, uid)
        dev.barrier()
    return dev, host


class _stdout_to_stderr(object):
    """Send file descriptor 1 to stderr for the duration of the block, so that
    banners printed by native libraries cannot land in a program's stdout
    (bench.py promises exactly one JSON line there)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        import sys
        sys.stdout.flush()
        try:
            # the native text sits in the C library's stdio buffer (stdout is a pipe:
            # fully buffered) and would otherwise be flushed into the REAL stdout at
            # exit, after the program's own output
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
