#!/usr/bin/env python3
"""Hunt for the wrong-J fault of the phased host-array path (sdp_problem_backup_host).

An in-process loop of the failing test is nearly blind: from the second iteration on, device buffers and pooled
page-locked blocks are recycled and still hold the RIGHT values of the previous iteration, so a row that reaches the
host without having been written compares equal.  Here every buffer on the way is filled with recognisable bytes
before each call (test build of the library, -DSDP_TEST_HOOKS: sdp_problem_debug_poison, sdp_debug_pollute), and a
wrong entry is classified by what it holds:

    host sentinel   the host array was never written there (copy missing / not finished)
    0xF1.. (J)      the copy read the device J before the kernel had written it
    0xE0.. (stage)  the 2-D copy read the conversion buffer before the transposition had written it
    0xD7.. (fresh)  a freshly allocated conversion buffer was read unwritten
    other           something else

    python tools/host_phase_stress.py [--iters N] [--mode default|pageable|reuse] [--model ar1|synth]
"""
import argparse
import contextlib
import ctypes as C
import io
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np                                           # noqa: E402
from stodynprog_amd import _native as nat                    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=200)
ap.add_argument('--mode', default='default', choices=['default', 'pageable', 'reuse', 'plain'])
ap.add_argument('--model', default='ar1', choices=['ar1', 'synth', 'synth32'])
ap.add_argument('--pollute', type=int, default=1)
ap.add_argument('--busy', type=int, default=0, help='a second solver sweeping between the calls')
ap.add_argument('--hoard', type=int, default=0, help='keep that many earlier solvers (their streams, buffers) alive')
args = ap.parse_args()

hooks = nat.build_library(test_hooks_to=os.path.join(tempfile.mkdtemp(), 'libsdp_hip_testhooks.so'))
nat.LIB_PATH = hooks
lib = nat.lib()
assert lib.sdp_test_hooks() == 1
lib.sdp_problem_debug_poison.restype = C.c_int
lib.sdp_problem_debug_poison.argtypes = [C.c_void_p]
lib.sdp_debug_pollute.restype = C.c_int
lib.sdp_debug_pollute.argtypes = [C.c_size_t, C.c_int]

from stodynprog_amd import models                            # noqa: E402

if args.mode == 'pageable':
    nat.PINNED_LIVE_BYTES = 0

SENT = np.frombuffer(np.array([0x7ff8dead0000beef], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
_orig_pinned = nat.pinned_empty


def sentinel_empty(shape, dtype):
    a = _orig_pinned(shape, dtype)
    a.view(np.uint8).reshape(-1)[:] = 0xA5
    return a


nat.pinned_empty = sentinel_empty


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make():
    if args.model == 'ar1':
        return models.storage_ar1(n_E=1200, n_P=1000, steps=(1.0, 0.1))[1]
    s = models.synthetic3d(N=144)[1]
    if args.model == 'synth32':
        s.dtype = np.dtype(np.float32)
    return s


def classify(words):
    """words: uint8 view [n, itemsize] of the wrong entries"""
    out = {}
    for name, byte in (('host-sentinel', 0xA5), ('device-J', 0xF1), ('device-pol', 0xF2), ('device-idx', 0xF3),
                       ('stage0', 0xE0), ('stage1', 0xE1), ('stage2', 0xE2), ('fresh-alloc', 0xD7), ('zero', 0x00)):
        n = int((words == byte).all(axis=1).sum())
        if n:
            out[name] = n
    out['other'] = len(words) - sum(out.values())
    return out


ref = make()
V = np.random.default_rng(21).standard_normal(ref._state_grid_shape).astype(ref.dtype)
Jr, pr = quiet(ref.value_iterations, V, 1, False)
idr = ref.last_policy_index
shape = ref._state_grid_shape
n0 = shape[0]
P = int(np.prod(shape[1:]))
print('model', args.model, 'shape', shape, 'kernel', ref.backend_info.get('kernel'), 'mode', args.mode, flush=True)

busy = make() if args.busy else None
hoard = []
bad_runs = 0
t0 = time.time()
solver = None
UINT = np.uint64 if np.dtype(ref.dtype).itemsize == 8 else np.uint32


def problems(sv):
    return [v for k, v in sv._cache.items() if isinstance(k, tuple) and k and k[0] == 'problem']


def check(J, pol, what):
    global bad_runs
    bad = J.view(UINT) != Jr.view(UINT)
    badp = pol.view(UINT) != pr.view(UINT)
    if not (bad.any() or badp.any()):
        return
    bad_runs += 1
    print('iteration', it, what, ': J wrong at', int(bad.sum()), 'entries; pol wrong at', int(badp.sum()), flush=True)
    if bad.any():
        rows, cols = np.nonzero(bad.reshape(n0, P))
        print('   J rows', rows.min(), '..', rows.max(), '(', len(np.unique(rows)), 'distinct ) columns', cols.min(),
              '..', cols.max(), '(', len(np.unique(cols)), 'distinct )')
        print('   J columns by phase quarter:', np.bincount((cols * 4) // P, minlength=4).tolist())
        w = J.reshape(-1).view(np.uint8).reshape(-1, J.itemsize)[bad.reshape(-1)]
        print('   what the wrong J entries hold:', classify(w))
    if badp.any():
        rows, cols = np.nonzero(badp.reshape(n0, P, -1).any(axis=2))
        print('   pol columns by phase quarter:', np.bincount((cols * 4) // P, minlength=4).tolist())
        w = pol.reshape(-1).view(np.uint8).reshape(-1, pol.itemsize)[badp.reshape(-1)]
        print('   what the wrong pol entries hold:', classify(w))


for it in range(args.iters):
    if solver is None or args.mode != 'reuse':
        if args.hoard and solver is not None:
            hoard.append(solver)
            del hoard[:-args.hoard]
        solver = None                                     # (the old problem's buffers go back to the allocator)
        if args.pollute:
            for mb in (4, 8, 10, 16, 20, 32, 64):
                lib.sdp_debug_pollute(mb << 20, 0xD7)
        solver = make()
        if args.mode == 'plain':
            solver.host_overlap = False
        # the first call creates the problem: its J / pol / idx are poisoned right after creation, its
        # conversion buffers come fresh from the allocator (polluted above)
        solver._debug_after_create = lambda p: lib.sdp_problem_debug_poison(p.h)
    if busy is not None:
        quiet(busy.value_iterations, V, 1, False)
    J, pol = quiet(solver.value_iteration, V, False)
    check(J, pol, '(first call)')
    del J, pol
    # second call on the same, now warm, problem with everything poisoned (the conversion buffers exist)
    for prob in problems(solver):
        lib.sdp_problem_debug_poison(prob.h)
    J, pol = quiet(solver.value_iteration, V, False)
    check(J, pol, '(warm call)')
    del J, pol
print('{} bad calls in {} iterations (2 calls each), mode {}, model {}, {:.0f} s'.format(
    bad_runs, args.iters, args.mode, args.model, time.time() - t0))
