#!/usr/bin/env python3
"""Finite-horizon storage control with perfect knowledge of a PV production
profile -- the workflow of the reference's examples/01 Deterministic storage
control (pv_storage_control.py, det_storage_control.py): a non-stationary
system whose cost looks the production up by time index, solved backwards
with `bellman_recursion`, then simulated forwards with the policy of each step.

    python examples/pv_storage.py [T_horiz] [N_E]
"""
from __future__ import division, print_function
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from stodynprog_amd import models


def main(T=14 * 48, N_E=100):
    sto_sys, dpsolv = models.pv_storage(T=T, N_E=N_E)
    sto_sys.print_summary()
    dpsolv.print_summary()
    t0 = time.time()
    J, pol = dpsolv.bellman_recursion(T, np.zeros(N_E))
    dt_solve = time.time() - t0
    info = dpsolv.backend_info
    print('solved {} steps x {} nodes x <= {} controls in {:.2f} s  (mode {}, kernel {}, '
          '{} lifted constants per step)'.format(T, N_E, info.get('max_controls'), dt_solve,
                                                 info['mode'], info.get('kernel'),
                                                 info.get('lifted_constants')))
    # forward simulation from a half-full storage (reference pv_storage_control.py:118-140)
    pol_sto = pol[..., 0]
    E = np.zeros(T + 1)
    E[0] = 1.0
    P_sto = np.zeros(T)
    for k in range(T):
        law = dpsolv.interp_on_state(pol_sto[k])
        P_sto[k] = law(E[k])
        E[k + 1], = sto_sys.dyn(k, E[k], P_sto[k])
    P_grid = dpsolv.P_prod_data - P_sto
    cost = sum(float(sto_sys.cost(k, E[k], P_sto[k])) for k in range(T))
    print('simulated cost {:.6f}  (cost-to-go at E=1.0, k=0: {:.6f})'.format(
        cost, float(dpsolv.interp_on_state(J[0])(1.0))))
    print('grid power: max {:.3f}, min {:.3f}; stored energy within [{:.3f}, {:.3f}]'.format(
        P_grid.max(), P_grid.min(), E.min(), E.max()))
    return J, pol, E, P_sto


if __name__ == '__main__':
    args = [int(a) for a in sys.argv[1:]]
    main(*args)
