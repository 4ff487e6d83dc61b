"""Multi-rank slab logic on CPU: world_size-2 gloo processes exchange J slabs
exactly as the RCCL path does on GPUs.  The per-rank sweep is played by the
CPU oracle (no GPU here); what is under test is the partition, the all-gather
assembly (uneven slabs included) and the relative-DP shift after the gather."""
import importlib.util
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from stodynprog_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds():
    b = dist.slab_bounds((256, 256, 256), 8)
    assert list(np.diff(b)) == [32 * 65536] * 8 and b[-1] == 256 ** 3
    b = dist.slab_bounds((10, 3), 4)
    assert list(b) == [0, 9, 18, 24, 30]
    b = dist.slab_bounds((2, 5), 4)                 # more ranks than planes: empty slabs
    assert list(b) == [0, 5, 10, 10, 10]
    assert list(dist.slab_bounds((7,), 1)) == [0, 7]


def test_phase_partition():
    b = dist.phase_partition(65536, 256, 8, 4)           # 256^3 column layout on 8 GPUs
    assert b.shape == (4, 9) and b[0, 0] == 0 and b[-1, -1] == 256 ** 3
    assert (np.diff(b, axis=1) == 2048 * 256).all()      # equal parts -> one ncclAllGather each
    assert (b[1:, 0] == b[:-1, -1]).all()                # phases are contiguous
    b = dist.phase_partition(10, 1, 3, 4)                 # uneven: 10 nodes, 3 ranks
    assert b[0, 0] == 0 and b[-1, -1] == 10 and (np.diff(b, axis=1) >= 0).all()
    assert (b[1:, 0] == b[:-1, -1]).all()
    assert dist.phase_partition(5, 7, 8, 4).shape[0] == 1   # fewer units than ranks: one phase
    b = dist.phase_partition(61, 41, 2, 4)
    assert (b % 41 == 0).all() and b[-1, -1] == 61 * 41


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, 'tests'))
import torch.distributed as dist_t
from gloo_comm import GlooCommunicator
from stodynprog_amd import dist, models, solver as solver_mod
from oracle import vi_numpy

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist_t.init_process_group('gloo', rank=rank, world_size=world)
comm = GlooCommunicator()
assert comm.nranks == world and not comm.is_device

sysd, ref = models.nas_demo(n_E=7, n_P=5, n_w=5)       # 7 planes over 2 ranks: uneven slabs
spec = vi_numpy.Spec.from_solver(ref)
bounds = comm.slab_bounds(spec.shape)
lo, hi = int(bounds[rank]), int(bounds[rank + 1])


class FakeProblem(object):
    """stands for _DeviceProblem: computes this rank's slab with the oracle"""
    layout = 0
    dev_shape = spec.shape
    def _to_device_order(self, A, extra=(), dtype=None):
        return np.ascontiguousarray(A)
    def _from_device_order(self, A, extra=()):
        return A.reshape(spec.shape + extra)
    def set_value(self, V):
        self.V = np.array(V, dtype=float)
    def sweep(self, t_k, rel_dp, ref_index):
        assert not rel_dp                     # the shift must happen after the gather
        nodes = np.arange(lo, hi)
        self.J = np.full(spec.shape, np.nan)
        self.pol = np.zeros(spec.shape + (1,)); self.idx = np.zeros(spec.shape, dtype=np.int32)
        J, pol, idx, _ = vi_numpy.value_iteration(spec, self.V, nodes=nodes)
        self.J.reshape(-1)[lo:hi] = J
        self.pol.reshape(-1, 1)[lo:hi] = pol
        self.idx.reshape(-1)[lo:hi] = idx
        return 0.0
    def get_value(self):
        return self.J.copy()
    def get_policy(self):
        return self.pol, self.idx

s = solver_mod.DPSolver(sysd, comm=comm)
s.state_grid, s.perturb_grid, s.perturb_proba = ref.state_grid, ref.perturb_grid, ref.perturb_proba
s._state_grid_shape, s._state_ref_ind = ref._state_grid_shape, ref._state_ref_ind
s.control_steps = ref.control_steps
s._problem = lambda t_k=None, model=None: FakeProblem()

V0 = np.zeros(spec.shape)
J1, pol1 = s.value_iteration(V0, report_time=False)
idx1 = s.last_policy_index
Jd = J1 - J1[s._state_ref_ind]
(J2, J2ref), pol2 = s.value_iteration((Jd, 0.), rel_dp=True, report_time=False)

# single-process oracle
E1, _, _, _ = vi_numpy.value_iteration(spec, V0)
(E2, E2ref), _, _, _ = vi_numpy.value_iteration(spec, (E1 - E1[spec.ref_ind], 0.), rel_dp=True)
assert np.array_equal(J1, E1), 'gathered J differs from the single-process sweep'
_, Epol, Eidx, _ = vi_numpy.value_iteration(spec, V0)
assert np.array_equal(pol1, Epol.reshape(pol1.shape)), 'gathered policy differs'
assert np.array_equal(idx1, Eidx.reshape(spec.shape))
assert np.array_equal(J2, E2) and J2ref == E2ref
assert J2[s._state_ref_ind] == 0.0
assert comm.allreduce_max(float(rank)) == world - 1
payload = comm.broadcast_bytes(b'id-from-rank0' if rank == 0 else None)
assert payload == b'id-from-rank0'
comm.barrier()
print('rank', rank, 'ok', flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(300)
def test_two_rank_gloo_sweep_matches_single_process(tmp_path):
    if importlib.util.find_spec('torch') is None:      # not imported here: see _native.py
        pytest.skip('torch not installed')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank {} failed:\n{}'.format(rank, out)
        assert 'rank {} ok'.format(rank) in out


def test_library_load_never_imports_torch():
    """north_star: "no PyTorch".  Loading the library -- also under a launcher
    (WORLD_SIZE > 1) -- must not pull torch in: the sharded process holds ONE ROCm
    runtime, the system one, and the RCCL unique id travels through a file."""
    code = ("import sys; sys.path.insert(0, {!r}); from stodynprog_amd import _native as nat, dist; "
            "nat.lib(); print('torch' in sys.modules, len(nat.rocm_runtimes()))").format(ROOT)
    for world in ('2', '1'):
        env = dict(os.environ, WORLD_SIZE=world, RANK='0')
        out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=200)
        assert out.returncode == 0, out.stdout.decode()
        assert out.stdout.decode().strip().splitlines()[-1] == 'False 1', out.stdout.decode()


def test_file_rendezvous_hands_the_id_to_every_rank(tmp_path, monkeypatch):
    """dist.FileRendezvous: atomic publish by rank 0, polling readers; a file whose
    writer is no longer alive (a crashed job, however recent) is ignored; rank 0
    removes its file"""
    import threading
    import time
    monkeypatch.setenv('SDP_RENDEZVOUS_DIR', str(tmp_path))
    monkeypatch.setenv('MASTER_PORT', '29123')
    monkeypatch.setenv('WORLD_SIZE', '4')
    payload = bytes(range(128))
    path = dist.FileRendezvous.default_path()
    assert path.startswith(str(tmp_path)) and '29123' in path
    # a FRESH file of a dead writer (a job that crashed a moment ago, same port) must not be taken
    dead = subprocess.Popen([sys.executable, '-c', 'pass'])
    dead.wait()
    with open(path, 'wb') as f:
        f.write(dist.FileRendezvous.MAGIC + '{} {} 128\n'.format(dead.pid, 12345).encode() + b'x' * 128)
    got = {}

    def reader(r):
        got[r] = dist.FileRendezvous(r, 4, timeout_s=20).exchange()

    threads = [threading.Thread(target=reader, args=(r,)) for r in (1, 2, 3)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    assert not got                                     # still waiting: the writer of that file is gone
    r0 = dist.FileRendezvous(0, 4)
    assert r0.exchange(payload) == payload
    for t in threads:
        t.join(20)
    assert got == {1: payload, 2: payload, 3: payload}
    r0.cleanup()
    assert not os.path.exists(path)
    # a live pid with the wrong start time (a recycled pid) and a truncated file are refused too
    for content in (dist.FileRendezvous.MAGIC + '{} {} 128\n'.format(os.getpid(), 1).encode() + b'y' * 128,
                    dist.FileRendezvous.MAGIC + '{} {} 128\n'.format(os.getpid(), 0).encode() + b'short'):
        with open(path, 'wb') as f:
            f.write(content)
        with pytest.raises(TimeoutError):
            dist.FileRendezvous(1, 4, timeout_s=0.3).exchange()
    # the ranks need not share a parent process (a launcher may wrap each of them)
    os.unlink(path)
    code = ("import sys; sys.path.insert(0, {!r}); from stodynprog_amd import dist; "
            "print(dist.FileRendezvous(1, 4, timeout_s=20).exchange().hex())").format(ROOT)
    child = subprocess.Popen(['sh', '-c', 'exec "$0" -c "$1"', sys.executable, code],
                             stdout=subprocess.PIPE, env=dict(os.environ))
    time.sleep(0.5)
    dist.FileRendezvous(0, 4).exchange(payload)
    assert child.communicate(timeout=60)[0].decode().strip() == payload.hex()


def test_tapered_phase_partition_covers_every_unit_once():
    for n_units, unit, nranks, n_phases in ((65536, 256, 8, 4), (400, 20, 3, 8), (61, 41, 2, 4), (10, 1, 8, 4)):
        for taper in (False, True):
            b = dist.phase_partition(n_units, unit, nranks, n_phases, taper)
            assert b[0, 0] == 0 and b[-1, -1] == n_units * unit
            assert (np.diff(b, axis=1) >= 0).all() and (b % unit == 0).all()
            assert np.array_equal(b[1:, 0], b[:-1, -1])              # phases are contiguous
            sizes = (b[:, -1] - b[:, 0]) // unit
            if taper and len(sizes) > 1 and n_units >= 8 * len(sizes):
                assert (np.diff(sizes) <= 0).all() and sizes[-1] < sizes[0]
    b = dist.phase_partition(65536, 256, 8, 4, True)
    assert [int(x) for x in (b[:, -1] - b[:, 0]) // 256] == [26214, 19661, 13107, 6554]


def test_slab_partition_and_peer_needs_on_the_host():
    """the sparse peer exchange: one slab of columns per rank (as nranks x phases single-owner
    phases in node order) and, per rank, the rows of J its backups read"""
    import numpy as np
    from stodynprog_amd import models, dist
    b = dist.slab_partition(100, 7, 3, 4)
    assert b.shape == (12, 4) and b[0, 0] == 0 and b[-1, -1] == 700
    assert (b[1:, 0] == b[:-1, -1]).all()                 # phases tile the node range in order
    for mp in range(12):
        r = mp // 4
        sizes = np.diff(b[mp])
        assert (sizes[np.arange(3) != r] == 0).all() and sizes[r] > 0 and (b[mp] % 7 == 0).all()
    assert dist.intervals_of([0, 1, 1, 0, 1]).tolist() == [[1, 3], [4, 5]]
    _, s = models.synthetic3d(N=32)
    m = s._traced()
    shape = s._shape()
    for nranks in (2, 5):
        parts = dist.slab_partition(32 * 32, 32, nranks, 3)
        off, ranges = s._peer_needs(m, parts, shape)
        assert off[0] == 0 and len(off) == nranks + 1 and off[-1] == len(ranges)
        ref_node = int(np.ravel_multi_index(s._state_ref_ind[1:], shape[1:])) * 32
        for r in range(nranks):
            iv = ranges[off[r]:off[r + 1]]
            assert (iv[:, 0] < iv[:, 1]).all() and (iv[1:, 0] >= iv[:-1, 1]).all() and (iv % 32 == 0).all()
            own_lo, own_hi = parts[r * 3, r], parts[r * 3 + 2, r + 1]
            assert all(e <= own_lo or b0 >= own_hi for b0, e in iv)          # its own rows never travel
            inside_own = own_lo <= ref_node < own_hi
            assert inside_own or any(b0 <= ref_node < e for b0, e in iv)      # the reference node is read
            assert 0 < (iv[:, 1] - iv[:, 0]).sum() < 32 ** 3
        # the dynamics contract towards the middle: far less than everybody else's rows
        assert (ranges[:, 1] - ranges[:, 0]).sum() / nranks < 0.5 * 32 ** 3 * (nranks - 1) / nranks
