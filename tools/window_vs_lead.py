#!/usr/bin/env python3
"""One stock on an axis too long for the LDS table (1024 x 128 x 128 x 64 controls x 32 perturbation points):
the row-window column kernel (kernel='column') against the reduced-array sweep (kernel='lead', what 'auto' picks
since round 3) -- kernel time per sweep, J and policy index bit for bit.  usage: through gpurun"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models
for kernel in ('column', 'lead', 'auto'):
    _, s = models.synthetic3d(N=128)
    s.discretize_state(0, 1, 1024, 0, 1, 128, 0, 1, 128)
    s.kernel = kernel
    V0 = models.synthetic3d_V0(s.state_grid)
    prob = s._problem()
    prob.set_value(V0)
    prob.bench_sweeps(2); prob.swap()
    _, k = prob.bench_sweeps(5)
    J = prob.get_value(); _, idx = prob.get_policy()
    print(kernel, s.backend_info['kernel'], s.backend_info.get('row_window'), '%.3f ms' % (k / 5), flush=True)
    if kernel == 'column': ref = (J, idx)
    else: print('identical', np.array_equal(J, ref[0]), np.array_equal(idx, ref[1]))
    for k_ in [k_ for k_ in s._cache if k_[0] == 'problem']: s._cache.pop(k_).close()
