#!/usr/bin/env python3
"""A/B of the certified filter on a cost that depends on the perturbation (sdp_col_cost_expect,
csrc/sdp_column_kernel.h): the benchmark problem with `+ (0.3 w) u` in its cost, 256^3 x 64 x 32
fp64, filter off / on -- kernel time per sweep, J and policy indices compared bit for bit.
usage: python tools/costw_ab.py      (through gpurun)"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from stodynprog_amd import models
out = {}
for flt in (False, True):
    sysd, s = models.synthetic3d(N=256)
    p = models.SYNTH
    k1, k0, eps, kx = p['k1'], p['k0'], p['eps'], p['kx']
    def cost(x0, x1, x2, u, w):
        e = (k1 * x1 - k0) - u
        return e * e + eps * (u * u) + kx * x0 + (0.3 * w) * u
    sysd.cost = cost
    s._cache.clear()
    s.certified_filter = flt
    V0 = models.synthetic3d_V0(s.state_grid)
    prob = s._problem()
    prob.set_value(V0)
    prob.bench_sweeps(3); prob.swap()
    _, k = prob.bench_sweeps(10)
    out[flt] = (prob.get_value(), prob.get_policy()[1], k / 10)
    print('filter', flt, s.backend_info['certified_filter'], 'kernel %.3f ms' % (k / 10), flush=True)
print('identical', np.array_equal(out[False][0], out[True][0]), np.array_equal(out[False][1], out[True][1]))
