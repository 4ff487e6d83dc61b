#!/usr/bin/env python3
"""A cascade of two reservoirs fed by an AR(1) inflow: TWO controlled state variables next to an
exogenous one, two controls.  The reference's API takes such a system like any other
(`SysDescription((3, 2, 1))`, stodynprog.py:57-81; the 2-D control lattice of stodynprog.py:655-660);
on the GPU it runs the reduced-array sweep (stodynprog_amd/csrc/sdp_lead_kernel.h).
Relative value iteration until the policy settles, then a closed-loop simulation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import SysDescription, DPSolver
from stodynprog_amd.models import NormalLaw


def main(n_a=48, n_b=48, n_y=24, n_iter=60, n_steps=400, verbose=True):
    casc = SysDescription((3, 2, 1), name='Two reservoirs')
    casc.state = ['upper', 'lower', 'inflow']
    casc.control = ['release_upper', 'release_lower']

    def dyn(a, b, y, u, v, w):
        'upper and lower levels, inflow (an AR(1) process)'
        return (a + (0.7 + 0.5 * y) - u, b + u - v, 0.3 + 0.7 * (y - 0.3) + w)
    casc.dyn = dyn
    casc.perturb_laws = [NormalLaw(0, 0.1)]

    def releases(a, b, y):
        return ((0., 1.), (0., 1.))
    casc.control_box = releases

    def cost(a, b, y, u, v, w):
        'the turbine should deliver 0.8; spilling and running dry are penalised'
        spill = np.where(a > 1.7, a - 1.7, 0.0 * a) + np.where(b > 1.7, b - 1.7, 0.0 * b)
        dry = np.where(a < 0.3, 0.3 - a, 0.0 * a) + np.where(b < 0.3, 0.3 - b, 0.0 * b)
        return (v - 0.8) * (v - 0.8) + 0.05 * (u - v) * (u - v) + 4.0 * spill + 8.0 * dry
    casc.cost = cost

    dpsolv = DPSolver(casc)
    dpsolv.discretize_state(0., 2., n_a, 0., 2., n_b, -0.2, 0.8, n_y)
    dpsolv.discretize_perturb(-0.3, 0.3, 9)
    dpsolv.control_steps = (0.125, 0.125)
    if verbose:
        casc.print_summary()
        dpsolv.print_summary()
    J = np.zeros(dpsolv._state_grid_shape)
    J_ref = 0.
    for k in range(n_iter):
        (J, J_ref), pol = dpsolv.value_iteration((J, J_ref), rel_dp=True, report_time=False)
    if verbose:
        print('kernel family:', dpsolv.backend_info['kernel'], '/', dpsolv.backend_info['filter_form'])
        print('average cost per stage after {} iterations: {:.5f}'.format(n_iter, J_ref))
    # closed loop: look the two releases up in the policy (multilinear interpolation), advance the system
    law_u = dpsolv.interp_on_state(pol[..., 0])
    law_v = dpsolv.interp_on_state(pol[..., 1])
    rng = np.random.default_rng(1)
    x = np.array([1.0, 1.0, 0.3])
    levels = np.zeros((n_steps, 3))
    out = np.zeros(n_steps)
    for k in range(n_steps):
        levels[k] = x
        xq = np.clip(x, [0., 0., -0.2], [2., 2., 0.8])       # look the policy up inside the grid
        u, v = float(law_u(*xq)), float(law_v(*xq))
        out[k] = v
        w = float(np.clip(rng.normal(0, 0.1), -0.3, 0.3))
        x = np.array(dyn(x[0], x[1], x[2], u, v, w))
    if verbose:
        print('turbine output: mean {:.3f}, std {:.3f}; levels stay in [{:.2f}, {:.2f}]'.format(
            out.mean(), out.std(), levels[:, :2].min(), levels[:, :2].max()))
    return dict(J=J, J_ref=J_ref, pol=pol, levels=levels, output=out, solver=dpsolv)


if __name__ == '__main__':
    main()
