#!/usr/bin/env python3
"""The benchmark problem with noise in the stock (x0' = (x0 + b u) - 0.07 w: the filter on the shifted lattice), whole
table against resident chunks: J / index bit for bit after a chain of sweeps, kernel time.
usage: python tools/noisy_ab.py [N]      (through gpurun)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ref = None
for c, extra in ((0, {}), (16, {}), (16, {'SDP_COL_MIN_WAVES': '2'})):
    _, s = models.synthetic3d(N=N, stock_noise=0.07)
    s.debug_defines = dict({'SDP_COL_WRES': str(c)}, **extra)
    prob = s._problem()
    V0 = models.synthetic3d_V0(s.state_grid)
    prob.set_value(V0)
    prob.bench_sweeps(3)
    prob.set_value(V0)
    _, k = prob.bench_sweeps(10)
    J = prob.get_value()
    _, idx = prob.get_policy()
    same = 'reference' if ref is None else 'J identical {}, index identical {}'.format(
        np.array_equal(J, ref[0]), np.array_equal(idx, ref[1]))
    ref = ref or (J, idx)
    print('resident points {:3d} {}: kernel {:7.3f} ms per sweep   {}'.format(c, extra, k / 10, same), flush=True)
    prob.close()
