"""The drop-in boundary refuses what it cannot run correctly (include/sdp_hip.h,
sdp_problem_create): a model code object that was generated for another problem -- other grid,
other real type, other control lattice -- is an SDP_EMODULE error naming the field, never a launch
that returns stale results.  (The reference raises on a bad shape too, multilinear_cython.pyx:46-47.)"""
import numpy as np
import pytest

from stodynprog_amd import models, _native as nat

pytestmark = pytest.mark.gpu


def _module_of(solver):
    return nat.compile_model(solver._kernel_plan()['source'])


@pytest.mark.parametrize('case', ['axis0', 'dtype', 'perturbations', 'controls'])
def test_a_code_object_built_for_another_problem_is_refused(gpu, monkeypatch, case):
    _, good = models.synthetic3d(N=20)
    if case == 'axis0':
        _, other = models.synthetic3d(N=24)
        expect = 'points along axis 0'
    elif case == 'dtype':
        _, other = models.synthetic3d(N=20)
        other.dtype = np.dtype('float32')
        expect = 'byte reals'
    elif case == 'perturbations':
        _, other = models.synthetic3d(N=20, n_w=16)
        expect = 'perturbation points'
    else:
        _, other = models.synthetic3d(N=20)
        other.control_steps = (0.064,)                   # 33 controls instead of 64
        expect = 'controls'
    wrong = _module_of(other)
    assert wrong != _module_of(good)
    monkeypatch.setattr(nat, 'compile_model', lambda source, verbose=False: wrong)
    with pytest.raises(RuntimeError) as e:
        good.value_iteration(np.zeros(good._state_grid_shape), report_time=False)
    assert 'was not built for this problem' in str(e.value) and expect in str(e.value), str(e.value)


def test_a_code_object_without_metadata_is_refused(gpu, monkeypatch, tmp_path):
    import subprocess
    src = tmp_path / 'bare.hip'
    src.write_text('#include <hip/hip_runtime.h>\nextern "C" __global__ void sdp_sweep_col(int) {}\n'
                   'extern "C" __global__ void sdp_evalpol_col(int) {}\n')
    out = tmp_path / 'bare.hsaco'
    subprocess.check_call([nat.HIPCC, '--genco', '--offload-arch=gfx950', '-O1', '-o', str(out), str(src)])
    _, s = models.synthetic3d(N=20)
    monkeypatch.setattr(nat, 'compile_model', lambda source, verbose=False: str(out))
    with pytest.raises(RuntimeError) as e:
        s.value_iteration(np.zeros(s._state_grid_shape), report_time=False)
    assert 'sdp_meta' in str(e.value)


def test_a_model_nobody_has_seen_is_compiled_on_this_box(gpu):
    """The code-object cache travels with the tree (stodynprog_amd/_kcache, keys = source + kernel headers + flags +
    `hipcc --version`), so a GPU run may never start the compiler (the driver's run did not).  This
    model carries a constant drawn now: its unit cannot be cached, hipcc runs HERE, and the result is numpy's."""
    import os
    from oracle import vi_numpy
    from stodynprog_amd import SysDescription, DPSolver, codegen
    nonce = float(np.frombuffer(os.urandom(8), dtype=np.uint64)[0] % 10 ** 9) / 1e9 + 0.25
    s = SysDescription((2, 1, 1))
    s.dyn = lambda x, y, u, w: (x + 0.3 * u, 0.8 * y + w)
    s.cost = lambda x, y, u, w: (x - nonce) ** 2 + 0.1 * u * u
    s.control_box = lambda x, y: ((-1., 1.),)
    s.perturb_laws = [models.NormalLaw(0, 0.1)]
    solver = DPSolver(s)
    solver.discretize_state(-1, 1, 33, -1, 1, 9)
    solver.discretize_perturb(-0.3, 0.3, 5)
    solver.control_steps = (0.1,)
    source = solver._kernel_plan()['source']
    assert float(nonce).hex() in source                                  # (constants are emitted as hexadecimal literals)
    out = os.path.join(nat.KCACHE, codegen.source_key(source) + '.hsaco')
    assert not os.path.exists(out)
    assert codegen.compiler_identity().startswith('HIP version')
    V = np.random.default_rng(9).standard_normal((33, 9))
    J, u = solver.value_iteration(V, report_time=False)
    assert os.path.exists(out) and os.path.getsize(out) > 10000          # compiled just now, on this machine
    Jo, uo, io, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(solver), V)
    assert np.array_equal(J, Jo) and np.array_equal(u, uo)
    for ext in ('.hsaco', '.hip'):                                        # (not worth keeping: never asked for again)
        os.unlink(out[:-6] + ext)
