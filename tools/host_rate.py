#!/usr/bin/env python3
"""PCIe-inclusive rate of DPSolver.value_iteration called with host arrays
(the reference's calling convention): numpy in, numpy out, every sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models
_, s = models.synthetic3d()
V = models.synthetic3d_V0(s.state_grid)
J, u = s.value_iteration(V, report_time=False)
t0 = time.perf_counter()
for _ in range(3):
    J, u = s.value_iteration(J, report_time=False)
dt = (time.perf_counter() - t0) / 3
print('value_iteration with host arrays, 256^3 fp64: {:.3f} s per call = {:.2f} sweeps/s '
      '(kernel {:.1f} ms)'.format(dt, 1 / dt, s._problem().last_kernel_ms()))
