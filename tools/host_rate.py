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
J, u = s.value_iteration(J, report_time=False)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    J, u = s.value_iteration(J, report_time=False)
dt = (time.perf_counter() - t0) / n
print('value_iteration with host arrays, 256^3 fp64: {:.4f} s per call = {:.1f} calls/s '
      '(kernel {:.2f} ms)'.format(dt, 1 / dt, s._problem().last_kernel_ms()))
t0 = time.perf_counter()
idx = s.last_policy_index
print('policy indices fetched on demand: {:.4f} s ({})'.format(time.perf_counter() - t0, idx.dtype))
# same bits as the device-resident loop
_, r = models.synthetic3d()
Jr, ur = r.value_iterations(V, n + 2, report_time=False)
print('bit-identical to the device-resident loop:', bool(np.array_equal(J, Jr) and np.array_equal(u, ur)
                                                        and np.array_equal(idx, r.last_policy_index)))
# pageable input every call (a fresh array the library has never seen)
t0 = time.perf_counter()
for _ in range(3):
    J2, u2 = s.value_iteration(np.array(J), report_time=False)
print('with a pageable input array each call: {:.4f} s per call'.format((time.perf_counter() - t0) / 3))
