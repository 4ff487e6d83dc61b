#!/usr/bin/env python3
"""Host time per step of a finite-horizon recursion with time-indexed data (models.pv_storage: the reference's
examples/01 Deterministic storage control): cProfile of `bellman_recursion` over the whole horizon."""
import os, sys, time, io, contextlib, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

for kw in (dict(), dict(T=14 * 48, N_E=100)):
    _, s = models.pv_storage(**kw)
    T = kw.get('T', None)
    with contextlib.redirect_stdout(io.StringIO()):
        s.bellman_recursion(3, np.zeros(s._state_grid_shape))          # (compile, warm up)
    n = T or 48
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    with contextlib.redirect_stdout(io.StringIO()):
        J, pol = s.bellman_recursion(n, np.zeros(s._state_grid_shape))
    pr.disable()
    dt = time.perf_counter() - t
    print('pv_storage {}: {} steps on {} nodes in {:.3f} s = {:.3f} ms per step'.format(kw, n, s._state_grid_shape, dt, dt / n * 1e3))
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(16)
    print('\n'.join(out.getvalue().splitlines()[6:26]))
