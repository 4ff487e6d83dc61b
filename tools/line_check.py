#!/usr/bin/env python3
"""The filtered line kernel (csrc/sdp_line_kernel.h: one state variable, x' = a(x, u) +- b(w)) against the direct kernel:
same J, policy and index on every node over a chain of sweeps, and the kernel time per sweep of both.     (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw


def inventory(n_x, n_u, n_w, smooth=True):
    sysd = SysDescription((1, 1, 1), name='inventory')
    sysd.dyn = lambda x, u, w: (x + u - w,)
    if smooth:
        sysd.cost = lambda x, u, w: 0.5 * u + 0.1 * (x - 2.0) * (x - 2.0)
    else:
        sysd.cost = lambda x, u, w: np.where(x > 0, x * 0.5, -x * 3.) + u * 1.
    sysd.control_box = lambda x: ((0., 8.),)
    sysd.perturb_laws = [NormalLaw(2.0, 0.8)]
    s = DPSolver(sysd)
    s.discretize_state(-8., 24., n_x)
    s.discretize_perturb(0., 4., n_w)
    s.control_steps = (8. / (n_u - 1),)
    return s


def run(make, kernel, V0, sweeps, reps=5):
    s = make()
    s.kernel = kernel
    prob = s._problem()
    prob.set_value(V0)
    prob.bench_sweeps(1)
    for _ in range(sweeps - 1):
        prob.swap()
        prob.bench_sweeps(1)
    J = prob.get_value()
    pol, idx = prob.get_policy()
    prob.swap()
    _, kern = prob.bench_sweeps(reps)
    return J, pol, idx, kern / reps, s.backend_info


sizes = [(600, 257, 16), (4096, 1025, 32), (65536, 1025, 16), (65536, 4097, 16)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for (n_x, n_u, n_w) in sizes:
    for smooth in (True, False):
        make = lambda: inventory(n_x, n_u, n_w, smooth)
        rng = np.random.default_rng(n_x)
        x = np.linspace(-8, 24, n_x)
        for vname, V0 in (('zeros', np.zeros(n_x)), ('smooth', 0.3 * (x - 3) ** 2 + np.sin(x)), ('random', rng.standard_normal(n_x))):
            ref = run(make, 'generic', V0, 4)
            got = run(make, 'line', V0, 4)
            same = (np.array_equal(ref[0], got[0]), np.array_equal(ref[1], got[1]), np.array_equal(ref[2], got[2]))
            print('{:6d} x {:5d} x {:3d} {:7s} V0 {:7s}: J, policy, index identical {}   direct {:8.3f} ms   line {:8.3f} ms  ({} / {})'.format(
                n_x, n_u, n_w, 'smooth' if smooth else 'kinked', vname, same, ref[3], got[3], ref[4]['kernel'], got[4]['kernel']), flush=True)
            if not all(same):
                bad = np.flatnonzero(ref[0] != got[0])
                print('      J differs at', bad[:10], 'of', bad.size, '; index differs at', np.flatnonzero(ref[2] != got[2])[:10])
