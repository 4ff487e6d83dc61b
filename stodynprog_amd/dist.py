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
  * a HOST communicator (the `Communicator` interface below with is_device False) -- host arrays; the CPU tests
    of the slab logic bring one over torch.distributed/gloo (tests/gloo_comm.py, world_size 2).  The product never
    imports torch: the RCCL unique id travels through a file (`FileRendezvous`).

Policies are gathered only on request (sdp_problem_get_policy is then a collective call);
the timed sweep never pays for it.
"""
import os

import numpy as np

__all__ = ['slab_bounds', 'phase_partition', 'slab_partition', 'Communicator', 'RcclCommunicator',
           'FileRendezvous', 'from_env']


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


def slab_partition(n_units, unit, nranks, n_phases):
    """Node bounds for the SPARSE peer exchange, in the same [phases][nranks+1] form: every
    rank owns ONE contiguous slab of the units (so that the rows its backups read are few and
    near its own), cut into `n_phases` pieces for the overlap.  Expressed as nranks * n_phases
    "phases" in node order in each of which a single rank has work: phase r * n_phases + k is
    piece k of rank r's slab (bounds equal to its begin for ranks <= r, to its end above)."""
    n_phases = max(1, min(int(n_phases), max(1, n_units // max(nranks, 1))))
    bounds = np.zeros((nranks * n_phases, nranks + 1), dtype=np.int64)
    for r in range(nranks):
        lo, hi = n_units * r // nranks, n_units * (r + 1) // nranks
        for k in range(n_phases):
            a, b = lo + (hi - lo) * k // n_phases, lo + (hi - lo) * (k + 1) // n_phases
            bounds[r * n_phases + k, :r + 1] = a * unit
            bounds[r * n_phases + k, r + 1:] = b * unit
    return bounds


def intervals_of(mask):
    """sorted [begin, end) runs of the True entries of a 1-D boolean array"""
    m = np.concatenate(([False], np.asarray(mask, dtype=bool), [False]))
    d = np.flatnonzero(m[1:] != m[:-1])
    return d.reshape(-1, 2)


class Communicator(object):
    """What DPSolver asks of a communicator.  `is_device` True: the library exchanges device buffers itself
    (RcclCommunicator).  False: a HOST communicator -- the solver sweeps this rank's slab and calls
    `all_gather_slabs(flat_array, bounds)` on host arrays; besides that `allreduce_max(value)` and `barrier()`.
    The product ships the RCCL one only; the gloo one of the CPU tests (tests/gloo_comm.py) implements the host
    interface over torch.distributed -- torch stays out of this package."""
    rank = 0
    nranks = 1
    is_device = False

    def slab_bounds(self, shape):
        return slab_bounds(shape, self.nranks)


_Base = Communicator


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
                'and PyTorch-ROCm (which bundles its own runtime) imported afterwards.  RCCL '
                'cannot initialise in that state; either do not import torch in this process '
                '(stodynprog_amd never needs it) or import it BEFORE stodynprog_amd'
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


def _proc_start_time(pid):
    """start time of process `pid` in clock ticks since boot (/proc/<pid>/stat, field 22), or
    None when there is no such process"""
    try:
        with open('/proc/{}/stat'.format(int(pid)), 'rb') as f:
            raw = f.read()
        return int(raw[raw.rindex(b')') + 2:].split()[19])
    except (OSError, ValueError, IndexError):
        return None


class FileRendezvous(object):
    """Hand-off of the 128-byte RCCL unique id between the ranks of ONE node
    through a file (default directory /dev/shm), with nothing but the Python
    standard library: a process that shards sweeps never imports torch.

    Rank 0 writes the id atomically (temp file + rename) together with its own
    pid and process start time; the others poll for the file and accept it
    only while THAT process is alive (same pid, same start time in
    /proc/<pid>/stat) -- a file left behind by a crashed job names a dead
    process and is ignored, however recent it is.  The file name is built from
    what a launcher gives every rank of a job: MASTER_PORT (one job per port on
    a node), WORLD_SIZE, torchelastic's run id and restart count; rank 0
    removes the file once every rank has joined the communicator.
    SDP_RENDEZVOUS_FILE overrides the path."""

    MAGIC = b'SDPUID1\n'

    def __init__(self, rank, world, timeout_s=None):
        self.rank, self.world = int(rank), int(world)
        self.timeout_s = float(os.environ.get('SDP_RENDEZVOUS_TIMEOUT', 600)
                               if timeout_s is None else timeout_s)
        self.path = os.environ.get('SDP_RENDEZVOUS_FILE') or self.default_path()

    @staticmethod
    def default_path():
        import tempfile
        base = os.environ.get('SDP_RENDEZVOUS_DIR') or (
            '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK)
            else tempfile.gettempdir())
        key = '_'.join(str(v) for v in (
            os.environ.get('MASTER_PORT', '0'), os.environ.get('WORLD_SIZE', '1'),
            os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
            os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')))
        key = ''.join(c if (c.isalnum() or c in '_-') else '-' for c in key)
        return os.path.join(base, 'sdp_rccl_uid_' + key)

    def publish(self, payload):
        """rank 0: make `payload` (bytes) visible to the other ranks"""
        pid = os.getpid()
        head = '{} {} {}\n'.format(pid, _proc_start_time(pid) or 0, len(payload)).encode()
        tmp = '{}.tmp.{}'.format(self.path, pid)
        with open(tmp, 'wb') as f:
            f.write(self.MAGIC + head + payload)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)          # atomic: readers see the old file, or all of the new one

    def _read(self):
        """payload of the file if it is complete and its writer is alive, else None"""
        try:
            with open(self.path, 'rb') as f:
                raw = f.read()
        except OSError:
            return None
        if not raw.startswith(self.MAGIC):
            return None
        try:
            head, rest = raw[len(self.MAGIC):].split(b'\n', 1)
            pid, start, size = (int(x) for x in head.split())
        except ValueError:
            return None
        if len(rest) != size:
            return None
        alive = _proc_start_time(pid)
        if alive is None or (start and alive != start):
            return None                      # written by a process that is gone: a crashed job's file
        return rest

    def fetch(self):
        """other ranks: wait for rank 0's payload"""
        import time
        t0 = time.time()
        delay = 0.002
        while True:
            payload = self._read()
            if payload is not None:
                return payload
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError('rank {}: no RCCL unique id from a live rank 0 at {} after {:.0f} s'
                                   .format(self.rank, self.path, self.timeout_s))
            time.sleep(delay)
            delay = min(delay * 1.5, 0.1)

    def exchange(self, payload=None):
        if self.rank == 0:
            self.publish(payload)
            return payload
        return self.fetch()

    def cleanup(self):
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


def from_env():
    """Communicator of a process started by `python -m torch.distributed.run`
    -- or by anything else that sets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_PORT: selects GPU LOCAL_RANK, hands rank 0's RCCL unique id to the
    other ranks through `FileRendezvous` (no torch, no gloo: the sharded
    process holds ONE ROCm runtime, the system one) and returns
    (RcclCommunicator, FileRendezvous).  With WORLD_SIZE absent or 1 returns
    (None, None)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None, None
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from . import _native as nat
    n_dev = nat.device_count()
    if n_dev < 1:
        raise nat.NativeError('rank {}: no HIP device visible'.format(rank))
    # one GPU per rank; a launcher that narrows each rank's visibility to its own GPU
    # (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES) leaves device 0 as the only choice
    nat.check(nat.lib().sdp_set_device(local % n_dev))
    rdv = FileRendezvous(rank, world)
    with _stdout_to_stderr():          # nothing native may write into the caller's stdout
        uid = rdv.exchange(RcclCommunicator.new_unique_id() if rank == 0 else None)
        dev = RcclCommunicator(rank, world, uid)
        dev.barrier()                  # every rank has read the id and joined
    rdv.cleanup()
    return dev, rdv


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
