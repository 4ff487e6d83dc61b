"""Sharded sweeps with the real HIP kernels: two processes share the one GPU of
the test box, each computes its slab of every backup through the C ABI and
the slabs meet over gloo (host memory).  The gathered results must equal the
single-process sweep bit for bit -- value, policy values, policy indices,
relative-DP reference cost, fixed-policy evaluation -- for the column layout
(slabs of columns) and for the node layout (slabs of rows).

RCCL itself needs one GPU per rank, which the test box does not have: the
device-side exchange is covered on one rank (test_gpu_sweep.py) and by
bench.py's `sharded_matches_single_gpu` self-check on the multi-GPU node."""
import importlib.util
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, 'tests'))
import numpy as np
import torch.distributed as tdist
tdist.init_process_group('gloo', init_method='env://')
from stodynprog_amd import models
from gloo_comm import GlooCommunicator

comm = GlooCommunicator()
rank = comm.rank
rng = np.random.default_rng(11)                     # same stream on both ranks

CASES = [('synthetic3d', dict(N=20), 'column'),
         ('storage_ar1', dict(), 'column'),          # 61 column planes: uneven slabs
         ('nas_demo', dict(), None),
         ('inventory', dict(), None),
         ('synthetic3d_coupled', dict(N=20), 'column'),              # table per control
         ('synthetic3d_coupled', dict(N=18, cross=0.2), 'generic'),  # a small grid: the direct kernel (round 5)
         ('synthetic3d_coupled', dict(N=18, cross=0.2), 'staged')]   # LDS-staged tiles (asked for: the planner keeps them for large grids)
for name, kw, kernel in CASES:
    _, one = getattr(models, name)(**kw)
    _, two = getattr(models, name)(**kw)
    two.comm = comm
    if kernel == 'staged' or name == 'synthetic3d_coupled' and kernel == 'column':
        one.kernel = two.kernel = kernel              # (small grids: the planner itself picks the direct kernel)
    shape = one._state_grid_shape
    V0 = rng.standard_normal(shape)
    J1, p1 = one.value_iteration(V0, report_time=False)
    i1 = one.last_policy_index
    J2, p2 = two.value_iteration(V0, report_time=False)
    i2 = two.last_policy_index
    if kernel is not None:
        assert one.backend_info['kernel'] == kernel == two.backend_info['kernel'], one.backend_info
    assert two._cache and any(k[0] == 'problem' for k in two._cache)
    prob = [v for k, v in two._cache.items() if k[0] == 'problem'][0]
    lo, hi = prob.node_range
    assert hi - lo < V0.size, 'the sharded solver must sweep a strict slab'
    assert np.array_equal(J1, J2), name
    assert np.array_equal(p1, p2), name
    assert np.array_equal(i1, i2), name
    # relative DP: shift after the gather
    ref = one._state_ref_ind
    Jd = J1 - J1[ref]
    (Ja, ra), _ = one.value_iteration((Jd, 0.), rel_dp=True, report_time=False)
    (Jb, rb), _ = two.value_iteration((Jd, 0.), rel_dp=True, report_time=False)
    assert np.array_equal(Ja, Jb) and ra == rb and Jb[ref] == 0.0, name
    # fixed-policy evaluation, plain and relative
    Ea = one.eval_policy(p1, 3, report_time=False)
    Eb = two.eval_policy(p1, 3, report_time=False)
    assert np.array_equal(Ea, Eb), name
    Ea, fa = one.eval_policy(p1, 4, rel_dp=True, report_time=False, J_ref_full=True)
    Eb, fb = two.eval_policy(p1, 4, rel_dp=True, report_time=False, J_ref_full=True)
    assert np.array_equal(fa, fb), (name, fa, fb)
    assert np.array_equal(Ea, Eb), name
    print('rank', rank, name, 'ok', flush=True)
comm.barrier()
print('rank', rank, 'all ok', flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.timeout(600)
def test_two_processes_one_gpu_sharded_sweeps_match_single_process(gpu, tmp_path):
    if importlib.util.find_spec('torch') is None:      # not imported here: see _native.py
        pytest.skip('torch not installed')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=560)[0].decode() for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank {} failed:\n{}'.format(rank, out)
        assert 'rank {} all ok'.format(rank) in out


LOAD_ORDER = r'''
import os, sys
sys.path.insert(0, {root!r})
from stodynprog_amd import _native as nat
nat.lib()                                   # WORLD_SIZE=2 in the environment
assert 'torch' not in sys.modules           # north_star: no PyTorch in the product
assert len(nat.rocm_runtimes()) == 1, nat.rocm_runtimes()
from stodynprog_amd.dist import RcclCommunicator
nat.check(nat.lib().sdp_set_device(0))
c = RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())     # real RCCL, the system runtime
c.barrier()
assert c.allreduce_max(3.5) == 3.5
c.close()
assert 'torch' not in sys.modules
# a user process that mixes in PyTorch AFTER the library holds two runtimes: refused, with the reason
import torch
if len(nat.rocm_runtimes()) > 1:
    try:
        RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
        raise SystemExit('two runtimes were not refused')
    except nat.NativeError as e:
        assert 'two HIP runtimes' in str(e)
print('load order ok', flush=True)
# two runtimes in one process also collide in their exit handlers (measured: 'double free
# or corruption' at interpreter shutdown) -- one more reason the product never imports torch
os._exit(0)
'''


@pytest.mark.timeout(300)
def test_real_rccl_without_torch_and_two_runtime_guard(gpu, tmp_path):
    """The sharded process never imports torch: real RCCL (system librccl + system
    HIP runtime) initialises with the library alone.  Regression of round 1:
    libsdp_hip.so mapped before torch left two HIP runtimes in the process and
    ncclCommInitRank failed ('no ROCm-capable device') -- now refused up front."""
    script = tmp_path / 'order.py'
    script.write_text(LOAD_ORDER.format(root=ROOT))
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=280)
    assert out.returncode == 0 and b'load order ok' in out.stdout, out.stdout.decode()


# ---------------------------------------------------------------------------
# several ranks through the LIBRARY's exchange path (sdp_comm_*, phased backups,
# in-place gathers, policy gather) on one GPU: RCCL refuses two ranks on one
# device, so the nccl* entry points come from tests/mock_rccl.cpp (host-staged,
# blocking) through SDP_RCCL_LIBRARY.  Everything above the collective calls is
# the product code that runs on the multi-GPU node.
# ---------------------------------------------------------------------------
def _build_mock(tmp_path, asynchronous=False, slot_mb=None):
    """the collective stand-in (tests/mock_rccl.cpp); `asynchronous`: the build whose calls only
    enqueue work on the stream, like the real library"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available to build the collective stand-in')
    out = str(tmp_path / ('libmock_rccl_async.so' if asynchronous else 'libmock_rccl.so'))
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-fPIC', '-shared', '-std=c++17']
                          + (['-DSDP_MOCK_ASYNC'] if asynchronous else [])
                          + (['-DSDP_MOCK_SLOT_MB={}'.format(int(slot_mb))] if slot_mb else [])
                          + ['-o', out, os.path.join(ROOT, 'tests', 'mock_rccl.cpp'), '-lrt'])
    return out


def _hooks_library(tmp_path):
    """the TEST build of the library (-DSDP_TEST_HOOKS): the only one that honours SDP_RCCL_LIBRARY"""
    from stodynprog_amd import _native as nat
    return nat.build_library(test_hooks_to=str(tmp_path / 'libsdp_hip_testhooks.so'))


def _with_hooks(tmp_path, script):
    """a launcher script that makes `script` run on the test build of the library: the product
    library has no way of being pointed at a stand-in for librccl"""
    hooks = _hooks_library(tmp_path)
    wrapper = tmp_path / ('hooks_' + os.path.basename(str(script)))
    wrapper.write_text(
        'import runpy, sys\nsys.path.insert(0, {root!r})\n'
        'from stodynprog_amd import _native as nat\nnat.LIB_PATH = {hooks!r}\n'
        'assert nat.lib().sdp_test_hooks() == 1\n'
        'sys.argv[0] = {script!r}\nrunpy.run_path({script!r}, run_name="__main__")\n'.format(
            root=ROOT, hooks=hooks, script=str(script)))
    return str(wrapper)


def _run_ranks(script, world, extra_env, timeout=300, argv=(), ok_status=(0,)):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1',
                   HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
        procs.append(subprocess.Popen([sys.executable, str(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    try:
        outs = [p.communicate(timeout=timeout) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:                                       # (exactly the processes started above)
            p.kill()
        tails = [[b.decode()[-1500:] for b in p.communicate()] for p in procs]
        raise AssertionError('{} ranks did not finish within {} s:\n{}'.format(
            world, timeout, '\n'.join('rank {}: {} | {}'.format(r, so, se) for r, (so, se) in enumerate(tails))))
    if os.environ.get('SDP_TEST_DUMP'):                   # (where the time of a multi-rank run goes: the ranks' own lines)
        with open(os.environ['SDP_TEST_DUMP'], 'a') as f:
            f.write('--- {} ranks, {}\n{}\n'.format(world, extra_env, outs[0][0].decode()))
    bad = [(rank, p.returncode, so.decode()[-1500:], se.decode()[-2500:])
           for rank, (p, (so, se)) in enumerate(zip(procs, outs)) if p.returncode not in ok_status]
    assert not bad, '\n'.join('rank {} failed (status {}):\n{}\n{}'.format(*b) for b in bad)
    return [so.decode() for so, _ in outs]


LIB_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
from stodynprog_amd import models, dist
dev, rdv = dist.from_env()
assert 'torch' not in sys.modules                   # the rendezvous is a file, not gloo
assert dev.is_device and dev.nranks == int(os.environ['WORLD_SIZE'])
rank = dev.rank
rng = np.random.default_rng(5)
import time
t_start = time.time()

def quiet(fn, *a, **k):
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

def where(a, b):
    """for an assertion message: how many entries differ, and over which index ranges"""
    bad = np.argwhere(np.asarray(a) != np.asarray(b))
    return 'identical' if not len(bad) else '%d of %d entries differ, index ranges %s' % (
        len(bad), np.asarray(a).size, [(int(lo), int(hi)) for lo, hi in zip(bad.min(axis=0), bad.max(axis=0))])

F64, F32 = 'float64', 'float32'
CASES = [('synthetic3d', dict(N=20), 4, F64), ('synthetic3d', dict(N=20), 3, F64), ('synthetic3d', dict(N=20), -4, F64),
         ('storage_ar1', dict(), 4, F64),                  # 61 columns: uneven parts -> broadcasts
         ('nas_demo', dict(), 4, F64), ('inventory', dict(), 4, F64),     # inventory: LDS-staged tiles, node ranges
         ('synthetic3d_coupled', dict(N=20), 4, F64),                     # column kernel, table per control
         ('synthetic3d_coupled', dict(N=18, cross=0.2), 3, F64),          # staged tiles in 3-D, ragged
         # round 4: every fast family shards -- the filter on the shifted lattice (a perturbation that reaches
         # the stock), the reduced-array sweep for two stocks (a rank reduces its own rows and their reach),
         # 4-byte reals (pair table, wide first pass)
         ('synthetic3d', dict(N=20, stock_noise=0.07), 2, F64),
         ('two_reservoirs', dict(n_a=24, n_b=12, n_y=8, n_w=5, steps=(0.25, 0.25)), 2, F64),
         ('two_reservoirs', dict(n_a=7, n_b=12, n_y=8, n_w=5, steps=(0.5, 0.25)), 1, F64),   # fewer rows than 8 ranks
         ('synthetic3d', dict(N=24), 2, F32),
         # round 5: a column count that is no multiple of ranks x phases (the reference's Searev grid: 61 x 61 = 3721
         # columns) -- uneven parts in every phase, padded and gathered with ONE all-gather -- over a chain of 20 sweeps
         ('searev', dict(), 2, F64)]
if os.environ.get('SDP_TEST_CASES'):
    CASES = [CASES[int(k)] for k in os.environ['SDP_TEST_CASES'].split(',')]
EXCHANGES = os.environ.get('SDP_TEST_EXCHANGES', 'rccl,peer,sparse,direct,sendrecv').split(',')
# (SDP_TEST_REST: the exchanges of every case but the first -- the first one takes them all -- and only the sweeps,
# not the whole API, for those cases: the 8-rank run is about partitions, mappings and the rendezvous)
REST = [e for e in os.environ.get('SDP_TEST_REST', '').split(',') if e]
REST_FROM = int(os.environ.get('SDP_TEST_REST_FROM', '1'))  # (the first REST_FROM cases take every exchange)
# (SDP_TEST_FULL: of those first cases only these exchanges run the whole API, the others the sweeps.  With three and
# more processes on ONE GPU every rendezvous of the stand-in costs a queue switch of the hardware scheduler -- seconds
# per exchange with many phases, 0.1 s with two processes -- so the larger worlds keep what is about THEM: partitions,
# mappings, need lists, the one-rendezvous exchange; the API surface is the 2-rank runs' job)
FULL = [e for e in os.environ.get('SDP_TEST_FULL', '').split(',') if e]
SPARSE_OK = ('synthetic3d', 'storage_ar1', 'nas_demo', 'searev')       # full-table column kernels: need lists
PLAN = [(c, e, (bool(REST) and k >= REST_FROM) or (bool(FULL) and e not in FULL))
        for k, c in enumerate(CASES) for e in (REST if (REST and k >= REST_FROM) else EXCHANGES)]
for (name, kw, phases, dtype), exchange, light in PLAN:
    _, one = getattr(models, name)(**kw)
    _, two = getattr(models, name)(**kw)
    one.dtype = two.dtype = np.dtype(dtype)
    two.comm = dev
    # 'peer': rows written into the peers' buffers (HIP IPC); 'sparse': one slab per rank, and a
    # peer is sent only the rows it reads (full-table column kernels; the others keep the full exchange);
    # 'direct': the backup kernel itself stores J into the ranks that read it (sparse where it applies)
    # 'sendrecv': the sparse exchange through the collective library alone (grouped ncclSend / ncclRecv of the bounding
    # range of what each rank reads; the all-gather where the model has no need lists)
    two.comm_exchange = 'peer' if exchange == 'sparse' else exchange
    two.comm_sparse = exchange in ('sparse', 'direct')
    if exchange == 'sendrecv':
        exchange = 'sendrecv' if name in SPARSE_OK else 'rccl'
    elif exchange == 'sparse':
        exchange = 'peer-sparse' if name in SPARSE_OK else 'peer'
    elif exchange == 'direct':
        exchange = 'direct-sparse' if name in SPARSE_OK else 'direct'
    two.comm_phases, two.comm_taper = abs(phases), phases < 0          # negative: tapered phases
    V0 = rng.standard_normal(one._state_grid_shape).astype(dtype).astype(float)
    J1, p1 = one.value_iteration(V0, report_time=False); i1 = one.last_policy_index
    J2, p2 = two.value_iteration(V0, report_time=False); i2 = two.last_policy_index
    prob = [v for k, v in two._cache.items() if k[0] == 'problem'][0]
    assert prob.parts is not None and prob.parts.shape[1] == dev.nranks + 1
    assert two.backend_info['exchange'] == exchange, two.backend_info
    if exchange.endswith('-sparse') or exchange == 'sendrecv':                     # strictly less than everybody else's rows
        assert 0.0 < prob.need_fraction < (dev.nranks - 1.0) / dev.nranks + 1e-12, prob.need_fraction
        assert prob.parts.shape[0] == dev.nranks * min(abs(phases), prob.parts.shape[0])
    if name == 'two_reservoirs':
        assert two.backend_info['kernel'] == 'lead' == one.backend_info['kernel'], two.backend_info
    if kw.get('stock_noise'):
        assert two.backend_info['filter_form'] == 'shifted lattice', two.backend_info
    assert np.array_equal(J1, J2), (name, exchange, where(J1, J2), prob.parts.tolist())
    assert np.array_equal(p1, p2) and np.array_equal(i1, i2), (name, exchange, where(p1, p2), where(i1, i2))      # get_policy gathers
    ref = one._state_ref_ind
    Jd = J1 - J1[ref]
    (Ja, ra), _ = one.value_iteration((Jd, 0.), rel_dp=True, report_time=False)
    (Jb, rb), _ = two.value_iteration((Jd, 0.), rel_dp=True, report_time=False)
    assert np.array_equal(Ja, Jb) and ra == rb, name
    if name == 'searev':
        assert any(len(set(np.diff(ph))) > 1 for ph in prob.parts), prob.parts       # uneven parts
        Ka, _ = quiet(one.value_iterations, V0, 20)
        Kb, _ = quiet(two.value_iterations, V0, 20)
        assert np.array_equal(Ka, Kb), name
    if light:
        print('rank', rank, name, phases, dtype, exchange, 'ok (sweeps only)', round(time.time() - t_start, 1), flush=True)
        continue
    Ea, fa = quiet(one.eval_policy, p1, 5, True, V0, J_ref_full=True)
    Eb, fb = quiet(two.eval_policy, p1, 5, True, V0, J_ref_full=True)     # fused shift, all ranks
    assert np.array_equal(fa, fb), (name, fa, fb)
    assert np.array_equal(Ea, Eb), name
    Ka, _ = quiet(one.value_iterations, V0, 3)
    Kb, _ = quiet(two.value_iterations, V0, 3)
    assert np.array_equal(Ka, Kb), name
    print('rank', rank, name, phases, dtype, exchange, 'ok', round(time.time() - t_start, 1), flush=True)
dev.barrier()
for k in [k for k in list(two._cache) if k[0] == 'problem']:
    two._cache.pop(k).close()
dev.close()
print('rank', rank, 'all ok', flush=True)
'''


# (8 ranks: the partition / mapping / rendezvous logic with more ranks than rows or columns in some cases -- one
# case per kernel family is enough there, every exchange; the blocking stand-in with 2 ranks only: the
# asynchronous one is the stricter test.  Round 3 ran all cases in all five set-ups: 280 s of the suite.)
EIGHT = '0,10,12'                    # (a column kernel with every exchange; two stocks with fewer rows than ranks; uneven parts over 20 sweeps)
REST8 = 'rccl,direct,sendrecv'       # every case but the first: the all-gather, the direct exchange, grouped sends / receives


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,asynchronous,cases', [(2, False, ''), (2, True, ''), (3, True, ''), (8, True, EIGHT)])
def test_library_exchange_path_with_several_ranks_on_one_gpu(gpu, tmp_path, world, asynchronous, cases):
    """asynchronous: the stand-in only enqueues (staging copies + host-function rendezvous in
    stream order), so the library's events and stream joins carry the ordering, as with RCCL"""
    mock = _build_mock(tmp_path, asynchronous)
    script = tmp_path / 'lib_worker.py'
    script.write_text(LIB_WORKER.format(root=ROOT))
    extra = dict(SDP_RCCL_LIBRARY=mock, SDP_TEST_CASES=cases)
    if world >= 8:
        extra['SDP_TEST_REST'] = REST8
        extra['SDP_TEST_FULL'] = 'rccl'                           # (the whole API through the all-gather; sweeps through the others)
    elif world == 3:
        extra['SDP_TEST_EXCHANGES'] = 'rccl,sparse,direct,sendrecv'       # (peer copies: the 2-rank runs)
        extra['SDP_TEST_CASES'] = '1,3,5,7,9,11'                  # (uneven parts of every family; all cases: 2 ranks)
        extra['SDP_TEST_REST'] = 'rccl,direct,sendrecv'           # (need lists with uneven parts: the first two cases)
        extra['SDP_TEST_REST_FROM'] = '2'
    outs = _run_ranks(_with_hooks(tmp_path, script), world, extra)
    for rank, out in enumerate(outs):
        assert 'rank {} all ok'.format(rank) in out, out


@pytest.mark.timeout(900)
@pytest.mark.parametrize('asynchronous', [False, True])
def test_bench_multi_rank_path_on_one_gpu(gpu, tmp_path, asynchronous):
    """bench.py exactly as the driver launches it for N > 1 (env of
    torch.distributed.run), two ranks on the one GPU through the stand-in:
    phase tuning, timed region, max over ranks, one JSON line on rank 0, and the
    sharded result checked against a single-GPU chain of sweeps"""
    import json
    mock = _build_mock(tmp_path, asynchronous)
    outs = _run_ranks(_with_hooks(tmp_path, os.path.join(ROOT, 'bench.py')), 2, dict(SDP_RCCL_LIBRARY=mock),
                      argv=['--gpus', '2', '--grid', '48', '--steps', '3', '--warmup', '1',
                            '--no-cpu-baseline', '--exchanges', 'rccl,sendrecv,direct,sparse,peer'])
    assert outs[1].strip() == ''
    lines = outs[0].strip().splitlines()
    assert len(lines) == 1, outs[0]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['sharded_matches_single_gpu'] is True, d
    # the RCCL plans are timed first (and reported whatever happens later), then the optional exchanges
    assert set(d['config']['comm_phase_tuning_ms_per_sweep']) == (
        {'1', '2', '4', '8', '16', '4t', '8t'} | {p + '/peer' for p in ('1', '2', '4', '8')}
        | {p + '/sparse' for p in ('1', '2', '4')} | {p + '/direct' for p in ('1', '2')}
        | {p + '/sendrecv' for p in ('1', '2', '4')}), d['config']
    assert d['config']['comm_exchange'] in ('rccl', 'sendrecv', 'peer', 'peer-sparse', 'direct-sparse'), d['config']
    # the work-equivalent chain (every control the long way), sharded like the headline
    assert d['every_control_the_long_way']['n_gpus'] == 2 and d['every_control_the_long_way']['sweeps_per_s'] > 0
    note = d['config']['comm_exchange_note']
    assert note is None or 'not faster in tuning' in note, note
    assert d['value'] > 0 and d['steps'] == 3 and d['warmup'] == 1
    import glob
    for leftover in glob.glob('/dev/shm/sdp_mock_*'):      # bench.py leaves its communicator to the OS
        os.unlink(leftover)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world,kind', [(4, 'raise'), (2, 'reject'), (2, 'hang')])
def test_bench_keeps_the_rccl_result_when_an_optional_exchange_fails(gpu, tmp_path, world, kind):
    """bench.py times and keeps the RCCL exchange first; an optional exchange that fails on one
    rank (SDP_BENCH_FAULT: an exception, a wrong J, a rank that never answers) must not cost the
    result: rank 0 prints ONE line, from an exchange that passed its checks, with the reason in
    config.comm_exchange_note.  An exception or a wrong J end with status 0; a rank that HANGS is a hung GPU
    process: the line is printed, then every rank leaves with status 3 (a hang must not look like success).  (4 ranks on the asynchronous stand-in for the
    exception case: the 8-rank form of this test stalled once in round 4 -- all ranks silent for 560 s, not reproduced
    in six further runs, cause not found; eight processes on one GPU are covered by the library test above; the watchdog case waits out its time limit, so it runs with 2 ranks.)"""
    import json
    mock = _build_mock(tmp_path, asynchronous=True)
    env = dict(SDP_RCCL_LIBRARY=mock, SDP_BENCH_FAULT='peer:{}:1'.format(kind), SDP_BENCH_OPTIONAL_TIMEOUT='45')
    argv = ['--gpus', str(world), '--grid', '48', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
            '--exchanges', 'rccl,direct,sparse,peer']           # (the exchanges over mapped buffers are opt-in)
    if kind == 'hang':
        argv[-1] = 'rccl,peer'
        env['SDP_BENCH_OPTIONAL_TIMEOUT'] = '8'           # (what the test waits for)
    outs = _run_ranks(_with_hooks(tmp_path, os.path.join(ROOT, 'bench.py')), world, env, argv=argv,
                      ok_status=(3,) if kind == 'hang' else (0,))
    lines = [l for l in outs[0].strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1 and all(o.strip() == '' for o in outs[1:]), outs
    d = json.loads(lines[0])
    assert d['n_gpus'] == world and d['value'] > 0 and d['sharded_matches_single_gpu'] is True, d
    note = d['config']['comm_exchange_note']
    times = d['config']['comm_phase_tuning_ms_per_sweep']
    assert {'1', '2', '4', '8', '16', '4t', '8t'} <= set(times)              # the RCCL plans were all timed
    assert d['config']['comm_exchange'] != 'peer'
    if kind == 'raise':
        assert 'peer exchange not used' in note, note
        assert not any(k.endswith('/peer') for k in times)
    elif kind == 'reject':
        assert 'peer exchange not used' in note, note
    else:
        assert 'peer exchange abandoned' in note and d['config']['comm_exchange'] == 'rccl', note
        assert d['config']['optional_exchange_hang'] == 'peer'
    import glob
    for leftover in glob.glob('/dev/shm/sdp_mock_*') + glob.glob('/dev/shm/sdp_rccl_uid_*'):
        try:
            os.unlink(leftover)
        except OSError:
            pass


@pytest.mark.timeout(900)
def test_bench_under_the_real_launcher(gpu, tmp_path):
    """The driver's exact command line for N > 1 -- `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`
    -- on the one GPU of the test box (both ranks on device 0, collectives through the
    stand-in): the launcher owns MASTER_PORT, sets TORCHELASTIC_* and is the parent of both
    workers, which is what dist.FileRendezvous keys its file on; the workers never import torch."""
    import json
    if importlib.util.find_spec('torch') is None:
        pytest.skip('torch (the launcher) not installed')
    mock = _build_mock(tmp_path)
    port = _free_port()
    env = dict(os.environ, SDP_RCCL_LIBRARY=mock, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           _with_hooks(tmp_path, os.path.join(ROOT, 'bench.py')), '--gpus', '2', '--grid', '48', '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=850,
                         cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['sharded_matches_single_gpu'] is True and d['value'] > 0, d
    assert d['config']['torch_imported'] is False
    import glob
    for leftover in glob.glob('/dev/shm/sdp_mock_*'):
        os.unlink(leftover)
    assert not glob.glob('/dev/shm/sdp_rccl_uid_{}_*'.format(port))      # rank 0 removed its file
