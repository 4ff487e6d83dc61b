#!/usr/bin/env python3
"""What a 1-D problem would gain from the column family: the same problem written with a second, inert state variable
(two grid points, z' = z, the cost-to-go zero in the second column) gets the filtered column kernel where its table
fits the LDS -- J and indices of the first column equal the 1-D problem's -- and the reduced-array sweep beyond.
Round 5 measured 0.93 -> 0.105 ms (600 nodes x 257 controls x 16 w, inventory shape), 1.33 -> 0.064 ms (2048 x 1025 x 8,
storage shape), 5.3 -> 4.4 ms (65536 x 4097 x 8) -- against the STAGED kernel.  The direct kernel, which `kernel = 'auto'` picks
for such problems since the same round, is 3 - 4 x faster still (0.023 and 0.027 ms): the lifting was built into DPSolver, measured
against it and taken out again (docs/NOTEBOOK.md section 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw
def make(n_x, n_u, n_w, noise, embedded):
    sysd = SysDescription((2 if embedded else 1, 1, 1), name='1-D')
    if embedded:
        if noise:
            sysd.dyn = lambda x, z, u, w: (x + u - w, z)
            sysd.cost = lambda x, z, u, w: 0.5 * u + 0.1 * (x - 2.0) * (x - 2.0)
            sysd.control_box = lambda x, z: ((0., 8.),)
        else:
            sysd.dyn = lambda x, z, u, w: (x + u, z)
            sysd.cost = lambda x, z, u, w: (u - 0.3 * w) * (u - 0.3 * w) + 0.05 * x
            sysd.control_box = lambda x, z: ((-1., 1.),)
    else:
        if noise:
            sysd.dyn = lambda x, u, w: (x + u - w,)
            sysd.cost = lambda x, u, w: 0.5 * u + 0.1 * (x - 2.0) * (x - 2.0)
            sysd.control_box = lambda x: ((0., 8.),)
        else:
            sysd.dyn = lambda x, u, w: (x + u,)
            sysd.cost = lambda x, u, w: (u - 0.3 * w) * (u - 0.3 * w) + 0.05 * x
            sysd.control_box = lambda x: ((-1., 1.),)
    sysd.perturb_laws = [NormalLaw(2.0, 0.8) if noise else NormalLaw(0., 1.)]
    s = DPSolver(sysd)
    if embedded:
        s.discretize_state(-8., 24., n_x, 0., 1., 2)
    else:
        s.discretize_state(-8., 24., n_x)
    s.discretize_perturb(0., 4., n_w)
    s.control_steps = ((8. if noise else 2.) / (n_u - 1),)
    return s
rng = np.random.default_rng(3)
for args in ((100, 65, 9, True), (600, 257, 16, True), (2048, 1025, 8, False), (65536, 4097, 8, False)):
    V1 = rng.standard_normal(args[0])
    out = {}
    for emb in (False, True):
        s = make(*args, emb)
        V = np.stack([V1, np.zeros_like(V1)], axis=1) if emb else V1
        J, pol = s.value_iteration(V, report_time=False)
        prob = s._problem(); prob.set_value(V); prob.bench_sweeps(2); loop, kern = prob.bench_sweeps(5)
        out[emb] = (J[:, 0] if emb else J, s.last_policy_index[:, 0] if emb else s.last_policy_index, kern / 5, s.backend_info['kernel'], s.backend_info['filter_form'])
    a, b = out[False], out[True]
    print(args, '1-D: %.3f ms %s | embedded: %.3f ms %s %s | J equal %s, index equal %s' % (a[2], a[3], b[2], b[3], b[4], np.array_equal(a[0], b[0]), np.array_equal(np.ravel(a[1]), np.ravel(b[1]))), flush=True)
