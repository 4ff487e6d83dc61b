#!/usr/bin/env python3
"""Inventory control: the tutorial problem of the reference's documentation
(doc/example_inventory.rst), written against stodynprog_amd.  A user script for
the reference differs from this one only by its import line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import SysDescription, DPSolver
from stodynprog_amd.models import DiscreteLaw


def main(n_sweeps=6, verbose=True):
    shop = SysDescription((1, 1, 1), name='Shop Inventory')

    def dyn_inv(x, u, w):
        'stock at the next period'
        return (x + u - w,)
    shop.dyn = dyn_inv
    shop.perturb_laws = [DiscreteLaw([0, 1, 2, 3], [0.2, 0.4, 0.3, 0.1])]

    def admissible_orders(x):
        return ((0, 10),)
    shop.control_box = admissible_orders

    h, p, c = 0.5, 3, 1

    def op_cost(x, u, w):
        'holding or shortage cost of the stock, plus the ordering cost'
        return np.where(x > 0, x * h, -x * p) + u * c
    shop.cost = op_cost

    dpsolv = DPSolver(shop)
    dpsolv.discretize_state(-3, 6, 10)
    dpsolv.discretize_perturb(0, 3, 4)
    dpsolv.control_steps = (1,)
    if verbose:
        shop.print_summary()
        dpsolv.print_summary()
    J = np.zeros(10)
    policies = []
    for k in range(n_sweeps):
        J, u = dpsolv.value_iteration(J, report_time=verbose)
        policies.append(u[..., 0].copy())
        if verbose:
            print(u[..., 0])
    if verbose:
        print('stock + order (x+u):', dpsolv.state_grid[0] + policies[-1])
    return J, policies


if __name__ == '__main__':
    main()
