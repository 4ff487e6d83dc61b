#!/usr/bin/env python3
"""Host time of ONE `value_iteration` call (numpy arrays in and out: the reference's convention, stodynprog.py:466-534)
next to the kernel time of its sweep, for the reference's own problem sizes -- where the call, not the kernel, is what a
user's loop waits for.  `--profile`: cProfile of 200 calls of the AR1 case."""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

cases = [('inventory (tutorial, 10 nodes)', models.inventory()[1]),
         ('inventory_fine (600 nodes x 257 controls)', models.inventory_fine()[1]),
         ('storage-AR1 reference size 41x61', models.storage_ar1()[1]),
         ('Searev reference size 31x61x61', models.searev()[1]),
         ('synthetic3d 48^3', models.synthetic3d(N=48)[1])]
for name, s in cases:
    V = np.zeros(s._state_grid_shape)
    J, pol = s.value_iteration(V, report_time=False)
    J, pol = s.value_iteration(J, report_time=False)
    n = 100
    t = time.perf_counter()
    for _ in range(n):
        J, pol = s.value_iteration(J, report_time=False)
    dt = (time.perf_counter() - t) / n
    prob = s._problem()
    prob.set_value(V); prob.bench_sweeps(2)
    _, kern = prob.bench_sweeps(10)
    print('{:45s} call {:7.3f} ms   kernel {:7.3f} ms   host share {:4.0f} %'.format(name, dt * 1e3, kern / 10, 100 * (1 - kern / 10 / (dt * 1e3))))
if '--profile' in sys.argv:
    import cProfile, pstats
    s = cases[2][1]
    J = np.zeros(s._state_grid_shape)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        J, pol = s.value_iteration(J, report_time=False)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(28)
    print(out.getvalue()[:6000])
