#!/usr/bin/env python3
"""Split the column kernel's sweep time into the part that scales with the
control count (phase B) and the fixed part (phases W and A, launch): time the
synthetic 256^3 x U x 32 sweep for several U and fit a line.
usage: [SDP_ARITH=fused] python tools/fixed_cost.py [float32|float64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

dtype = np.dtype(sys.argv[1] if len(sys.argv) > 1 else 'float64')
rows = []
for step in (4.0, 0.5, 0.125, 2 / 62.5):
    _, s = models.synthetic3d(N=256)
    s.dtype = dtype
    s.control_steps = (step,)
    s.arithmetic = os.environ.get('SDP_ARITH', 'exact')
    U = len(s.control_grids((0.5, 0.5, 0.5))[0][0])
    prob = s._problem(None)
    prob.set_value(models.synthetic3d_V0(s.state_grid, dtype))
    prob.bench_sweeps(3)
    loop_ms, kern_ms = prob.bench_sweeps(10)
    rows.append((U, kern_ms / 10))
    print('U = {:3d}: {:8.3f} ms/sweep'.format(U, kern_ms / 10), flush=True)
    prob.close()
U = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
slope, icpt = np.polyfit(U, T, 1)
print('{}: fixed {:.3f} ms + {:.4f} ms per control  (64 controls: {:.3f} ms in phase B)'.format(
    dtype.name, icpt, slope, 64 * slope))
