"""ctypes binding of libsdp_hip.so (C ABI: include/sdp_hip.h) and the build /
cache logic for the library and for generated-model code objects.

There is deliberately NO CPU fallback here: if the HIP library is missing,
cannot be loaded, or no GPU is visible, every compute entry point raises.
"""
import ctypes as C
import os
import re
import subprocess
import threading

import numpy as np

from . import codegen

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
REPO = os.path.dirname(_HERE)
LIB_PATH = os.path.join(CSRC, 'libsdp_hip.so')
KCACHE = os.path.join(_HERE, '_kcache')

SDP_F64, SDP_F32 = 0, 1
LAYOUT_NODES, LAYOUT_COLUMNS = 0, 1
VARIANT_DIRECT, VARIANT_STAGED = 0, 1
_ERR = {-1: ValueError, -2: Exception, -3: RuntimeError, -4: MemoryError,
        -5: RuntimeError, -6: RuntimeError}

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
codegen.HIPCC_PATH = HIPCC
LIB_FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-fast-math',
             '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value']


class NativeError(RuntimeError):
    pass


class sdp_problem_desc(C.Structure):
    _fields_ = [
        ('dtype', C.c_int32), ('d', C.c_int32), ('nu', C.c_int32), ('W', C.c_int32),
        ('orders', C.c_int64 * 4),
        ('axes', C.c_void_p * 4),
        ('wgrid', C.c_void_p), ('proba', C.c_void_p),
        ('box_per_node', C.c_int32), ('lanes_per_node', C.c_int32),
        ('layout', C.c_int32), ('variant', C.c_int32),
        ('box_lo', C.c_void_p), ('box_hi', C.c_void_p), ('box_n', C.c_void_p),
        ('node_begin', C.c_int64), ('node_end', C.c_int64),
        ('module_path', C.c_char_p),
        ('tile', C.c_int32 * 4),
        ('col_seg_nodes', C.c_int32), ('reserved', C.c_int32),
    ]


_lib = None
_lock = threading.Lock()


def _sources():
    return [os.path.join(CSRC, f) for f in
            ('sdp_hip.hip', 'sdp_device.h', 'sdp_kernel_args.h')] + \
           [os.path.join(REPO, 'include', 'sdp_hip.h')]


def build_library(force=False, verbose=False, test_hooks_to=None):
    """Compile csrc/sdp_hip.hip for gfx950 into csrc/libsdp_hip.so (in-tree).
    `test_hooks_to`: build the TEST variant (-DSDP_TEST_HOOKS: honours SDP_RCCL_LIBRARY, the
    collective stand-in of tests/) to that path instead -- never the product library."""
    srcs = _sources()
    out = test_hooks_to or LIB_PATH
    if (not force and os.path.exists(out)
            and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs)):
        return out
    cmd = [HIPCC] + LIB_FLAGS + (['-DSDP_TEST_HOOKS'] if test_hooks_to else []) + [
        '-o', out, os.path.join(CSRC, 'sdp_hip.hip')]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out


def _declare(lib):
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    sig = {
        'sdp_last_error': (C.c_char_p, []),
        'sdp_device_count': (C.c_int, [P(C.c_int)]),
        'sdp_set_device': (C.c_int, [C.c_int]),
        'sdp_device_info': (C.c_int, [C.c_int, C.c_char_p, P(C.c_int), P(i64), C.c_char_p]),
        'sdp_synchronize': (C.c_int, []),
        'sdp_mlinterp_f64': (C.c_int, [C.c_int, vp, vp, vp, vp, i64, vp, i64, vp]),
        'sdp_mlinterp_f32': (C.c_int, [C.c_int, vp, vp, vp, vp, i64, vp, i64, vp]),
        'sdp_interp_create': (C.c_int, [C.c_int, C.c_int, vp, vp, vp, vp, i64, P(vp)]),
        'sdp_interp_eval': (C.c_int, [vp, vp, i64, vp]),
        'sdp_interp_destroy': (C.c_int, [vp]),
        'sdp_problem_create': (C.c_int, [P(sdp_problem_desc), P(vp)]),
        'sdp_problem_destroy': (C.c_int, [vp]),
        'sdp_problem_set_value': (C.c_int, [vp, vp]),
        'sdp_problem_set_policy': (C.c_int, [vp, vp]),
        'sdp_problem_set_params': (C.c_int, [vp, vp, i32]),
        'sdp_problem_vi_sweep': (C.c_int, [vp, dbl, C.c_int, i64, P(dbl)]),
        'sdp_problem_eval_policy': (C.c_int, [vp, i32, C.c_int, i64, vp]),
        'sdp_problem_swap': (C.c_int, [vp]),
        'sdp_problem_get_value': (C.c_int, [vp, vp]),
        'sdp_problem_get_policy': (C.c_int, [vp, vp, vp]),
        'sdp_problem_last_kernel_ms': (C.c_int, [vp, P(dbl)]),
        'sdp_problem_bench_sweeps': (C.c_int, [vp, i32, C.c_int, i64, P(dbl), P(dbl)]),
        'sdp_problem_debug_stamps': (C.c_int, [vp, C.c_int, vp, i64]),
        'sdp_problem_backup_host': (C.c_int, [vp, vp, dbl, C.c_int, i64, vp, vp, vp, P(dbl)]),
        'sdp_problem_set_host_overlap': (C.c_int, [vp, C.c_int]),
        'sdp_problem_simulate': (C.c_int, [vp, vp, i64, i64, vp, vp, dbl, vp, vp, vp]),
        'sdp_host_alloc': (C.c_int, [C.c_size_t, P(vp)]),
        'sdp_host_free': (C.c_int, [vp]),
        'sdp_comm_library': (C.c_char_p, []),
        'sdp_test_hooks': (C.c_int, []),
        'sdp_comm_unique_id': (C.c_int, [C.c_char_p]),
        'sdp_comm_create': (C.c_int, [C.c_int, C.c_int, C.c_char_p, P(vp)]),
        'sdp_comm_destroy': (C.c_int, [vp]),
        'sdp_problem_attach_comm': (C.c_int, [vp, vp, i32, vp]),
        'sdp_problem_set_peer_needs': (C.c_int, [vp, vp, vp]),
        'sdp_problem_complete_value': (C.c_int, [vp]),
        'sdp_problem_enable_peer_exchange': (C.c_int, [vp]),
        'sdp_problem_disable_peer_exchange': (C.c_int, [vp]),
        'sdp_problem_set_direct_exchange': (C.c_int, [vp, C.c_int]),
        'sdp_problem_set_sendrecv_exchange': (C.c_int, [vp, C.c_int]),
        'sdp_problem_set_lead_halo': (C.c_int, [vp, i64]),
        'sdp_comm_allreduce_max': (C.c_int, [vp, P(dbl)]),
        'sdp_comm_barrier': (C.c_int, [vp]),
        'sdp_tab_create': (C.c_int, [C.c_int, vp, vp, vp, vp, P(vp)]),
        'sdp_tab_destroy': (C.c_int, [vp]),
        'sdp_tab_backup': (C.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return sorted(sig)


EXPORTS = None


def lib():
    """The loaded library.  Raises NativeError when it is absent (no fallback)."""
    global _lib, EXPORTS
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeError(
                    'HIP extension {} is missing: build it with '
                    '`python -c "import __graft_entry__ as g; g.build()"` '
                    '(needs hipcc); there is no CPU fallback'.format(LIB_PATH))
            try:
                loaded = C.CDLL(LIB_PATH)
            except OSError as e:
                raise NativeError('cannot load {}: {}'.format(LIB_PATH, e))
            EXPORTS = _declare(loaded)
            _lib = loaded
    return _lib


def rocm_runtimes():
    """Paths of the distinct libamdhip64 objects mapped into this process.

    The PyTorch-ROCm wheel ships its own copies of libamdhip64 / libhsa-runtime64
    / librccl.  A process that maps libsdp_hip.so (linked against /opt/rocm) and
    imports torch AFTERWARDS holds two ROCm runtimes, and RCCL then binds to the
    one that never initialised a device (measured: ncclCommInitRank fails with
    hsa_system_get_info 4107 / 'no ROCm-capable device').  This package never
    imports torch (the multi-process rendezvous is dist.FileRendezvous), so its
    own processes hold one runtime; `RcclCommunicator` checks this list and
    refuses to start in a process where the user mixed the two."""
    seen = []
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                path = line.split()[-1]
                if 'libamdhip64' in path and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def check(rc):
    if rc != 0:
        msg = lib().sdp_last_error().decode(errors='replace')
        raise _ERR.get(rc, RuntimeError)(msg)


def device_count():
    n = C.c_int(0)
    rc = lib().sdp_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_gpu():
    if device_count() < 1:
        raise NativeError('no HIP device visible: stodynprog_amd computes on an AMD GPU only '
                          '(there is no CPU fallback)')


def device_info(device=0):
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(64)
    cus = C.c_int(0)
    mem = C.c_int64(0)
    check(lib().sdp_device_info(device, name, C.byref(cus), C.byref(mem), arch))
    return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=mem.value,
                arch=arch.value.decode())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------
# generated-model code objects
# ---------------------------------------------------------------------------
def _hipcc(cmd, src, verbose=False):
    if verbose:
        print(' '.join(cmd))
    try:
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    except FileNotFoundError:
        raise NativeError('hipcc not found at {}: cannot compile the model kernel'.format(HIPCC))
    except subprocess.CalledProcessError as e:
        raise NativeError('hipcc failed on the generated model {}:\n{}'.format(
            src, e.stdout.decode(errors='replace')))


WAVES_CAPS = (4, 2, 1)        # waves per SIMD asked of the register allocator when a kernel's spill code is unsafe


def compile_model(source, verbose=False):
    """Compile a generated translation unit to a gfx950 code object, cached
    in-tree under stodynprog_amd/_kcache/<key>.hsaco.  Returns its path.

    A code object whose kernels spill vector registers is compiled to assembly as well and scanned for spill
    code that runs before the execution mask is restored (codegen.spill_hazards: the compiler does that, and the
    kernel then computes with registers of lanes that were never stored); such a kernel is rebuilt with fewer
    waves per SIMD asked of the register allocator (-DSDP_WAVES_CAP) until the scan is clean -- what was done
    is kept next to the code object in <key>.build.txt -- or refused."""
    key = codegen.source_key(source)
    os.makedirs(KCACHE, exist_ok=True)
    out = os.path.join(KCACHE, key + '.hsaco')
    if os.path.exists(out):
        return out
    src = os.path.join(KCACHE, key + '.hip')
    with open(src + '.tmp.{}'.format(os.getpid()), 'w') as f:     # ranks may compile concurrently
        f.write(source)
    os.replace(f.name, src)
    tmp = out + '.tmp.{}'.format(os.getpid())
    asm = tmp + '.s'
    flags = list(codegen.HIPCC_FLAGS)
    to_asm = ['--cuda-device-only', '-S'] + [x for x in flags if x != '--genco']
    _hipcc([HIPCC] + flags + ['-o', tmp, src], src, verbose)
    try:
        with open(tmp, 'rb') as f:
            may_spill = codegen.code_object_may_spill(f.read())
        if may_spill:
            log = []
            asked = re.search(r'^#define SDP_COL_MIN_WAVES (\d+)', source, re.M)       # (a cap at or above what the unit asks for changes nothing)
            for cap in (None,) + tuple(c for c in WAVES_CAPS if asked is None or c < int(asked.group(1))):
                extra = [] if cap is None else ['-DSDP_WAVES_CAP={}'.format(cap)]
                _hipcc([HIPCC] + to_asm + extra + ['-o', asm, src], src, verbose)
                with open(asm) as f:
                    hazards = codegen.spill_hazards(f.read())
                log.append('waves cap {}: {}'.format(cap, '; '.join(
                    '{} {}: {} before the mask restore'.format(k, b, ', '.join(ins)) for k, b, _, ins in hazards) or 'clean'))
                if not hazards:
                    break
            else:
                raise NativeError('the compiler places spill code before the execution mask is restored in {} at every '
                                  'register budget tried:\n{}'.format(src, '\n'.join(log)))
            if cap is not None:
                _hipcc([HIPCC] + flags + extra + ['-o', tmp, src], src, verbose)
            if len(log) > 1:
                with open(os.path.join(KCACHE, key + '.build.txt'), 'w') as f:
                    f.write('\n'.join(log) + '\n')
        os.replace(tmp, out)
    finally:
        for leftover in (tmp, asm):
            if os.path.exists(leftover):
                os.remove(leftover)
    return out


# ---------------------------------------------------------------------------
# page-locked host arrays (sdp_host_alloc) for what crosses the API every call
# ---------------------------------------------------------------------------
class _PinnedBlock(object):
    """One page-locked allocation; goes back to the pool when the numpy arrays
    viewing it are gone."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        try:
            _pinned_release(self.ptr, self.nbytes)
        except Exception:
            pass


_pinned_free = {}            # nbytes -> [ptr, ...]
_pinned_live = {}            # ptr -> nbytes of blocks currently viewed by arrays
PINNED_KEEP = 4              # free blocks kept per size ...
PINNED_KEEP_BYTES = 4 << 30  # ... and in total (page-locked memory is a scarce resource)
# page-locked bytes in arrays the caller still holds: a user loop that keeps every J_k / pol_k
# (as the reference allows) must not pin host RAM without bound -- beyond this, and whenever
# hipHostMalloc refuses, results come in ordinary pageable arrays (slower copies, same values)
PINNED_LIVE_BYTES = 8 << 30


def _pinned_release(ptr, nbytes):
    _pinned_live.pop(ptr, None)
    free = _pinned_free.setdefault(nbytes, [])
    pooled = sum(n * len(v) for n, v in _pinned_free.items())
    if len(free) < PINNED_KEEP and pooled + nbytes <= PINNED_KEEP_BYTES:
        free.append(ptr)
    elif _lib is not None:
        _lib.sdp_host_free(C.c_void_p(ptr))


def pinned_empty(shape, dtype):
    """numpy array of `shape` / `dtype` in page-locked memory (uninitialised).
    An ordinary, writable ndarray for the caller; the memory returns to a small
    pool when the array and its views are garbage-collected."""
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    nbytes_alloc = max(nbytes, 8)
    free = _pinned_free.get(nbytes_alloc)
    if free:
        ptr = free.pop()
    else:
        if sum(_pinned_live.values()) + nbytes_alloc > PINNED_LIVE_BYTES:
            return np.empty(shape, dtype=dt)
        p = C.c_void_p()
        if lib().sdp_host_alloc(nbytes_alloc, C.byref(p)) != 0 or not p.value:
            return np.empty(shape, dtype=dt)          # (sdp_problem_backup_host takes any host memory)
        ptr = p.value
    _pinned_live[ptr] = nbytes_alloc
    buf = (C.c_char * nbytes_alloc).from_address(ptr)
    buf._sdp_block = _PinnedBlock(ptr, nbytes_alloc)          # lifetime: array -> buf -> block
    return np.frombuffer(buf, dtype=dt, count=nbytes // dt.itemsize).reshape(shape)


def is_pinned(a):
    """does the array's memory lie inside a live page-locked block of ours?"""
    if not isinstance(a, np.ndarray) or not a.flags.c_contiguous:
        return False
    lo = a.ctypes.data
    hi = lo + a.nbytes
    for ptr, n in _pinned_live.items():
        if ptr <= lo and hi <= ptr + n:
            return True
    return False


def np_real(dtype):
    dt = np.dtype(dtype)
    if dt == np.float64:
        return SDP_F64
    if dt == np.float32:
        return SDP_F32
    raise ValueError('dtype must be float64 or float32, not {}'.format(dt))
