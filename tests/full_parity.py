#!/usr/bin/env python3
"""Full-size parity run (test infrastructure, like tests/): ONE complete
256^3 x 64 x 32 fp64 sweep of the benchmark problem on the GPU against the C
oracle (oracle/sdp_oracle.c, OpenMP over the host cores) on ALL 16.7 M nodes --
J bit for bit, policy index exact.  Takes ~40 s of CPU on the 256-thread box.
usage: python tests/full_parity.py [N]   (also run by tests/test_gpu_full_size.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models
from oracle import c_oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_, solver = models.synthetic3d(N=N)
V0 = models.synthetic3d_V0(solver.state_grid)
t = time.time()
J, pol = solver.value_iteration(V0, report_time=False)
idx = solver.last_policy_index
print('GPU sweep incl. host copies: {:.3f} s  ({})'.format(time.time() - t, solver.backend_info['kernel']))
threads = c_oracle.max_threads()
t = time.time()
Jo, io, mo = c_oracle.vi_synth3d(solver.state_grid, V0, models.SYNTH_PAR, -1., 1., 64,
                                 solver.perturb_grid[0], solver.perturb_proba[0],
                                 n_nodes=V0.size, n_threads=threads)
print('C oracle, {} threads: {:.1f} s'.format(threads, time.time() - t))
Jo = Jo.reshape(V0.shape); io = io.reshape(V0.shape)
print('nodes: {}   J bit-identical: {}   max |dJ|: {:.3e}   index mismatches: {}'.format(
    V0.size, bool(np.array_equal(J, Jo)), float(np.abs(J - Jo).max()), int((idx != io).sum())))
u = np.linspace(-1., 1., 64)
print('policy values consistent with indices: {}'.format(bool(np.array_equal(pol[..., 0], u[idx]))))
sys.exit(0 if np.array_equal(J, Jo) and np.array_equal(idx, io) else 1)
