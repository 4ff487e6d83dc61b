#!/usr/bin/env python3
"""Storage control for the SEAREV wave-energy converter: policy iteration,
then a closed-loop simulation under the optimal policy, and a pickle round trip
of the policy interpolator -- the workflow of the reference's
examples/20 Searev storage control/storage_control.py.

The simulation runs twice: as the reference writes it (a host loop with one
interp_on_state call per step, storage_control.py:242-251) and with
DPSolver.simulate, which runs the same loop -- for a whole batch of
disturbance sequences -- on the GPU without a host round trip per step."""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models


def main(n_val=200, n_pol=2, n_sim=2000, grid=(31, 61, 61), verbose=True):
    wec, dpsolv = models.searev(n_E=grid[0], n_S=grid[1], n_A=grid[2])
    if verbose:
        dpsolv.print_summary()
    pol_lin = models.searev_linear_policy(dpsolv)
    (J, r), pol = dpsolv.policy_iteration(pol_lin, n_val, n_pol, rel_dp=True)
    if verbose:
        print('reference cost after {:d} policy improvements: {:3f}'.format(n_pol, r))

    # the optimal storage power law as an interpolating function of the state
    P_sto_law = dpsolv.interp_on_state(pol[..., 0])
    P_sto_law = pickle.loads(pickle.dumps(P_sto_law))        # the example saves it to disk

    # closed-loop trajectory under the optimal law
    dt, E_rated = models.SEAREV['dt'], models.SEAREV['E_rated']
    rng = np.random.default_rng(0)
    w = rng.normal(0., models.SEAREV['innov_std'], n_sim)
    E = np.zeros(n_sim + 1); S = np.zeros(n_sim + 1); A = np.zeros(n_sim + 1)
    E[0] = E_rated / 3
    P_sto = np.zeros(n_sim); P_prod = np.zeros(n_sim)
    for k in range(n_sim):
        P_prod[k] = models.searev_power(S[k])
        P_sto[k] = P_sto_law(E[k], S[k], A[k])
        E[k + 1], S[k + 1], A[k + 1] = wec.dyn(E[k], S[k], A[k], P_sto[k], w[k])
    P_grid = P_prod - P_sto
    if verbose:
        print('simulated {:d} steps: std(P_prod) = {:.4f} MW, std(P_grid) = {:.4f} MW, '
              'E in [{:.2f}, {:.2f}] MJ'.format(n_sim, P_prod.std(), P_grid.std(), E.min(), E.max()))

    # the same trajectory, and 255 more disturbance sequences, on the device
    import time
    W = np.column_stack([w, rng.normal(0., models.SEAREV['innov_std'], (n_sim, 255))])
    x0 = np.tile([E_rated / 3, 0., 0.], (256, 1))
    t0 = time.perf_counter()
    x, u, g = dpsolv.simulate(pol, x0, W)
    dt_dev = time.perf_counter() - t0
    same = np.array_equal(x[:, 0, 0], E) and np.array_equal(u[:, 0, 0], P_sto)
    if verbose:
        print('DPSolver.simulate: 256 trajectories x {:d} steps in {:.3f} s; the first one is {} the '
              'host loop; mean cost {:.5f} (relative-DP reference cost {:.5f})'.format(
                  n_sim, dt_dev, 'bit-identical to' if same else 'DIFFERENT from', g.mean(), r))
    return dict(J=J, J_ref=r, pol=pol, E=E, P_prod=P_prod, P_grid=P_grid, batch_cost=g.mean(),
                device_matches_host=same)


if __name__ == '__main__':
    main()
