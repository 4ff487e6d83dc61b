#!/usr/bin/env python3
"""Stand-alone multilinear interpolation throughput on the cases of the
reference's own benchmark notebook (stodynprog/linear_interp_benchmark.ipynb:
2-D 50x51 grid / 1.001 M points, 3-D 50x51x52 grid / 5.005 M points) plus a
cache-missing case (256^3 grid, 5 M uniformly random points)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd.dolointerpolation import MultilinearInterpolator
from stodynprog_amd.interp import _DeviceValues

rng = np.random.default_rng(0)
for name, orders, n in (('2-D 50x51, 1.001 M pts', (50, 51), 1001000),
                        ('3-D 50x51x52, 5.005 M pts', (50, 51, 52), 5005000),
                        ('3-D 256^3, 5 M random pts', (256, 256, 256), 5000000)):
    d = len(orders)
    vals = rng.standard_normal((1, int(np.prod(orders))))
    s = np.ascontiguousarray(rng.uniform(0, 1, (d, n)))
    dev = _DeviceValues(np.zeros(d), np.ones(d), np.array(orders), vals)
    dev.eval(s)
    t0 = time.perf_counter()
    for _ in range(3):
        out = dev.eval(s)
    dt = (time.perf_counter() - t0) / 3
    print('{:28s} {:8.2f} ms per call incl. PCIe copies = {:7.1f} Mpts/s (fp64)'.format(name, dt * 1e3, n / dt / 1e6))
