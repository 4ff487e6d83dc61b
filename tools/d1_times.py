#!/usr/bin/env python3
"""One state variable (the reference's tutorial and `det/` examples are 1-D; VERDICT r04: "d = 1 problems never enter
the column family ... a cliff for a fine 1-D grid with thousands of controls per node"): what such problems cost on the
kernel they get -- since round 5 the direct kernel (`lanes` threads per node), before that the LDS-staged tile kernel (a
thread per node: 30 - 60 x slower at these sizes) --, fine grids and thousands of controls included.  Kernel time per sweep by
HIP events, lattice cells (node x control x perturbation point) per second beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw


def inventory(n_x, n_u, n_w):
    """x' = x + u - w (reference doc/example_inventory.py:31-33), smooth holding / shortage cost"""
    sysd = SysDescription((1, 1, 1), name='inventory')
    sysd.dyn = lambda x, u, w: (x + u - w,)
    sysd.cost = lambda x, u, w: 0.5 * u + 0.1 * (x - 2.0) * (x - 2.0)
    sysd.control_box = lambda x: ((0., 8.),)
    sysd.perturb_laws = [NormalLaw(2.0, 0.8)]
    s = DPSolver(sysd)
    s.discretize_state(-8., 24., n_x)
    s.discretize_perturb(0., 4., n_w)
    s.control_steps = (8. / (n_u - 1),)
    return s


def storage(n_x, n_u):
    """x' = x + u, nothing random in the dynamics (reference examples/01 Deterministic storage control)"""
    sysd = SysDescription((1, 1, 1), name='storage')
    sysd.dyn = lambda x, u, w: (x + u,)
    sysd.cost = lambda x, u, w: (u - 0.3 * w) * (u - 0.3 * w) + 0.05 * x
    sysd.control_box = lambda x: ((-1., 1.),)
    sysd.perturb_laws = [NormalLaw(0., 1.)]
    s = DPSolver(sysd)
    s.discretize_state(0., 10., n_x)
    s.discretize_perturb(-2., 2., 8)
    s.control_steps = (2. / (n_u - 1),)
    return s


def sweep_ms(solver, V, reps=5):
    prob = solver._problem()
    prob.set_value(V)
    prob.bench_sweeps(2)
    loop, kern = prob.bench_sweeps(reps)
    return kern / reps, solver.backend_info


for name, make, cells in (
        ('inventory 100 nodes x 65 controls x 9 w (tutorial size)', lambda: inventory(100, 65, 9), 100 * 65 * 9),
        ('inventory 600 nodes x 257 controls x 16 w', lambda: inventory(600, 257, 16), 600 * 257 * 16),
        ('storage 2048 nodes x 1025 controls x 8 w', lambda: storage(2048, 1025), 2048 * 1025 * 8),
        ('inventory 4096 nodes x 1025 controls x 32 w', lambda: inventory(4096, 1025, 32), 4096 * 1025 * 32),
        ('inventory 65536 nodes x 4097 controls x 16 w', lambda: inventory(65536, 4097, 16), 65536 * 4097 * 16),
        ('storage 65536 nodes x 4097 controls x 8 w', lambda: storage(65536, 4097), 65536 * 4097 * 8)):
    for kernel in ('auto', 'staged'):
        s = make()
        s.kernel = kernel                  # ('auto': the direct kernel for one state variable since round 5; 'staged': what ran before)
        V = np.zeros(s._state_grid_shape)
        ms, info = sweep_ms(s, V)
        print('{:50s} {:7s} sweep {:9.3f} ms  ({:.3g} lattice cells, {:.3g} cells/s, kernel {})'.format(
            name, kernel, ms, float(cells), cells / ms * 1e3, info['kernel']))
