#!/usr/bin/env python3
"""Kernel times of the BASELINE.json parity configs (not bench lines): sweep and
policy-evaluation durations measured with HIP events inside the library."""
import io, contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

def sweep_ms(solver, V, reps=5):
    prob = solver._problem()
    prob.set_value(V)
    prob.bench_sweeps(2)
    loop, kern = prob.bench_sweeps(reps)
    return kern / reps, solver.backend_info

cases = [
    ('C2 storage-AR1 200x200, <=51 controls, 9 w', models.storage_ar1(n_E=200, n_P=200, steps=(8. / 49, 0.1))[1]),
    ('storage-AR1 reference size 41x61, <=8001 controls', models.storage_ar1()[1]),
    ('C3 Searev 128^3, <=33 controls, 9 w', models.searev(n_E=128, n_S=128, n_A=128, step=2.2 / 31)[1]),
    ('Searev reference size 31x61x61, <=2201 controls', models.searev()[1]),
]
for name, s in cases:
    V = np.zeros(s._state_grid_shape)
    ms, info = sweep_ms(s, V)
    lo, hi, n = s._box_table()
    cells = float(np.prod(n.astype(np.int64), axis=0).sum()) * len(s.perturb_grid[0]) if n.shape[1] > 1 else \
        float(np.prod(n[:, 0])) * V.size * len(s.perturb_grid[0])
    print('{:55s} sweep {:8.3f} ms  ({:.3g} lattice cells, {:.3g} cells/s, kernel {})'.format(
        name, ms, cells, cells / ms * 1e3, info['kernel']))
_, sea = models.searev()
pol = models.searev_linear_policy(sea)
for n_it in (1000,):
    t0 = time.perf_counter(); quiet(sea.eval_policy, pol, n_it, True); dt = time.perf_counter() - t0
    print('Searev eval_policy {} iterations: {:.3f} s wall ({:.1f} us/iteration), device {:.3f} s'.format(
        n_it, dt, dt / n_it * 1e6, sea._problem().last_kernel_ms() / 1e3))
_, ar1 = models.storage_ar1()
pol = models.storage_ar1_empirical_policy(ar1)
t0 = time.perf_counter(); quiet(ar1.eval_policy, pol, 50, True); dt = time.perf_counter() - t0
print('storage-AR1 eval_policy 50 iterations: {:.4f} s wall, device {:.2f} ms'.format(dt, ar1._problem().last_kernel_ms()))
