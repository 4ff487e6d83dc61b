#!/usr/bin/env python3
"""Resident-chunk form of the filtered column kernel (csrc/sdp_colres_kernel.h) against the plain one on the
benchmark problem: J / policy index bit for bit over a chain of sweeps, and the kernel time.
usage: python tools/wres_ab.py [N] [chunk sizes ...]      (through gpurun)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunks = [int(c) for c in sys.argv[2:]] or [16, 20, 24]
ref = None
for c in [0] + chunks:
    _, s = models.synthetic3d(N=N)
    s.debug_defines = {'SDP_COL_WRES': str(c)}
    prob = s._problem()
    V0 = models.synthetic3d_V0(s.state_grid)
    prob.set_value(V0)
    prob.bench_sweeps(3)
    prob.set_value(V0)
    _, k = prob.bench_sweeps(10)
    J = prob.get_value()
    _, idx = prob.get_policy()
    if ref is None:
        ref = (J, idx)
        same = 'reference'
    else:
        same = 'J identical {}, index identical {}'.format(np.array_equal(J, ref[0]), np.array_equal(idx, ref[1]))
    print('resident points {:3d}: kernel {:7.3f} ms per sweep   {}'.format(c, k / 10, same), flush=True)
    prob.close()
