"""The benchmark sweep at FULL size against the oracle on every node
(tests/full_parity.py): 256^3 x 64 x 32 fp64, J bit for bit, policy index
exact.  The oracle side is ~32 s of OpenMP C on the 256-thread GPU box; on a
host with few cores it would take many minutes, so the test needs >= 64
threads (or SDP_FULL_PARITY=1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(1500)
def test_full_256cubed_sweep_matches_the_oracle_on_every_node(gpu):
    if (os.cpu_count() or 1) < 64 and not os.environ.get('SDP_FULL_PARITY'):
        pytest.skip('needs >= 64 host threads for the full-size oracle sweep (or SDP_FULL_PARITY=1)')
    out = subprocess.run([sys.executable, os.path.join(HERE, 'full_parity.py')],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1400)
    text = out.stdout.decode()
    assert out.returncode == 0, text
    assert 'nodes: 16777216   J bit-identical: True' in text and 'index mismatches: 0' in text, text


@pytest.mark.timeout(900)
def test_chained_sweeps_at_full_size_match_the_oracle_on_sampled_nodes(gpu):
    """What bench.py times are sweeps 5..25 of a chain, not the sweep from the closed-form V0:
    sweep 6 of the chain at 256^3, from the device's own J_5, against the C oracle on 20 000
    sampled nodes -- J bit for bit, indices exact."""
    import io
    import contextlib
    import numpy as np
    from stodynprog_amd import models
    from oracle import c_oracle
    _, s = models.synthetic3d(N=256)
    V0 = models.synthetic3d_V0(s.state_grid)
    with contextlib.redirect_stdout(io.StringIO()):
        J5, _ = s.value_iterations(V0, 5)
        J6, _ = s.value_iteration(J5)
    idx6 = s.last_policy_index
    assert s.backend_info['certified_filter']
    nodes = np.random.default_rng(6).choice(V0.size, size=20000, replace=False)
    Jo, io_, _ = c_oracle.vi_synth3d(s.state_grid, J5, models.SYNTH_PAR, -1., 1., 64,
                                     s.perturb_grid[0], s.perturb_proba[0], node_ids=nodes,
                                     n_threads=min(c_oracle.max_threads(), 64))
    assert np.array_equal(J6.ravel()[nodes], Jo)
    assert np.array_equal(idx6.ravel()[nodes], io_)


@pytest.mark.timeout(900)
def test_every_node_through_the_multi_survivor_pass_at_full_size(gpu, debug_defines):
    """At the benchmark size no node keeps a second control in 8-byte reals, so the second pass's
    walk over several survivors never runs there: force it (radius x 1e12: every node keeps
    several) once at 256^3 and compare all 16.7 M nodes with the ordinary run."""
    import numpy as np
    from stodynprog_amd import models
    _, a = models.synthetic3d(N=256)
    V0 = models.synthetic3d_V0(a.state_grid)
    Ja, _ = a.value_iteration(V0, report_time=False)
    ia = a.last_policy_index
    debug_defines.set(SDP_COL_FILTER_SCALE='1e12')
    _, b = models.synthetic3d(N=256)
    Jb, _ = b.value_iteration(V0, report_time=False)
    assert 'SDP_COL_FILTER_SCALE' in b._kernel_plan()['source']
    assert np.array_equal(Ja, Jb) and np.array_equal(ia, b.last_policy_index)


@pytest.mark.timeout(900)
def test_noise_in_the_stock_at_full_size(gpu):
    """The benchmark problem with the perturbation also in the stock (x0' = (x0 + b u) - 0.07 w: the shape of the
    reference's inventory example) at 256^3: sweep 4 of a chain with the certified filter on the shifted
    lattice against the kernel that evaluates every control the long way on all 16.7 M nodes, and against
    the numpy oracle on 3000 sampled nodes -- J bit for bit, indices exact."""
    import io
    import contextlib
    import numpy as np
    from stodynprog_amd import models
    from oracle import vi_numpy
    out = {}
    for flt in (True, False):
        _, s = models.synthetic3d(N=256, stock_noise=0.07)
        s.certified_filter = flt
        V0 = models.synthetic3d_V0(s.state_grid)
        with contextlib.redirect_stdout(io.StringIO()):
            J3, _ = s.value_iterations(V0, 3)
            J4, _ = s.value_iteration(J3)
        out[flt] = (J3, J4, s.last_policy_index, s)
        assert s.backend_info['filter_form'] == ('shifted lattice' if flt else None)
        for k_ in [k_ for k_ in s._cache if k_[0] == 'problem']:
            s._cache.pop(k_).close()
    assert np.array_equal(out[True][0], out[False][0]) and np.array_equal(out[True][1], out[False][1])
    assert np.array_equal(out[True][2], out[False][2])
    nodes = np.random.default_rng(8).choice(out[True][0].size, size=3000, replace=False)
    Jo, _, io_, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(out[True][3]), out[True][0], nodes=np.sort(nodes))
    assert np.array_equal(out[True][1].ravel()[np.sort(nodes)], Jo)
    assert np.array_equal(out[True][2].ravel()[np.sort(nodes)], io_)


@pytest.mark.timeout(900)
def test_two_reservoirs_at_bench_size(gpu):
    """Two controlled stocks, 128 x 128 x 64 nodes x 16 x 16 controls x 16 perturbation points (bench.py --config
    reservoirs): sweep 3 of a chain on the reduced-array sweep against the direct kernel on all 1 M nodes and
    against the numpy oracle on 2000 sampled nodes."""
    import io
    import contextlib
    import numpy as np
    from stodynprog_amd import models
    from oracle import vi_numpy
    out = {}
    for kernel in ('auto', 'generic'):
        _, s = models.two_reservoirs(n_a=128, n_b=128, n_y=64, n_w=16, steps=(1. / 15, 1. / 15))
        s.kernel = kernel
        a, b, y = [np.asarray(g) for g in s.state_grid]
        V0 = ((a[:, None, None] - 1.0) ** 2 + 0.5 * (b[None, :, None] - 0.7) ** 2
              + 0.3 * np.cos(3 * y)[None, None, :] * (1 + 0.1 * a[:, None, None]))
        with contextlib.redirect_stdout(io.StringIO()):
            J2, _ = s.value_iterations(V0, 2)
            J3, _ = s.value_iteration(J2)
        out[kernel] = (J2, J3, s.last_policy_index, s)
        assert s.backend_info['kernel'] == ('lead' if kernel == 'auto' else 'generic')
        for k_ in [k_ for k_ in s._cache if k_[0] == 'problem']:
            s._cache.pop(k_).close()
    assert np.array_equal(out['auto'][1], out['generic'][1]) and np.array_equal(out['auto'][2], out['generic'][2])
    nodes = np.sort(np.random.default_rng(9).choice(out['auto'][0].size, size=2000, replace=False))
    Jo, _, io_, _ = vi_numpy.value_iteration(vi_numpy.Spec.from_solver(out['auto'][3]), out['auto'][0], nodes=nodes)
    assert np.array_equal(out['auto'][1].ravel()[nodes], Jo)
    assert np.array_equal(out['auto'][2].ravel()[nodes], io_)
