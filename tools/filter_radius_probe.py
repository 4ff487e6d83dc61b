#!/usr/bin/env python3
"""How much margin does the filter's error radius have?  On an objective that is flat in the
control (tests/test_gpu_filter.py:_flat -- the reference's argmin hangs on rounding noise) the
radius is scaled DOWN (SDP_COL_FILTER_SCALE) until policy indices start to differ from the
kernel that evaluates every control the long way.  Prints mismatching nodes per scale.
usage: python tools/filter_radius_probe.py      (through gpurun)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import DPSolver
from tests.test_gpu_filter import _flat, _sweep


def _scale(v):
    DPSolver.debug_defines = {'SDP_COL_FILTER_SCALE': v} if v else None


for dtype in (np.float64, np.float32):
    for box_on_state in (False, True):
        make = lambda: _flat(0.0, box_on_state=box_on_state)[:2]
        V = _flat(0.0)[2]
        _scale(None)
        off = _sweep(make, False, V, dtype)
        row = []
        for scale in ('1', '0.5', '0.25', '0.1', '3e-2', '1e-2', '3e-3', '1e-3', '1e-4', '1e-6'):
            _scale(scale)
            on = _sweep(make, True, V, dtype)
            row.append('{}: {}'.format(scale, int((on[2] != off[2]).sum())))
        _scale(None)
        print('{} {:24s} nodes {:5d}   index mismatches by radius scale   {}'.format(
            np.dtype(dtype).name, 'box depends on x0' if box_on_state else 'constant box (table)',
            V.size, '   '.join(row)), flush=True)

# the shifted lattice (a perturbation that reaches the stock): the same objective with noise in the stock --
# the cost-to-go is linear in the stock, so the interpolation bound vanishes and the rounding part decides
from tests import test_gpu_shift as ts
for box_on_state in (False, True):
    make = lambda: ts._flat_shop(0.0, box_on_state)[:2]
    V = ts._flat_shop(0.0)[2]
    _scale(None)
    off = ts._sweep(make, False, V)
    row = []
    for scale in ('1', '0.5', '0.1', '1e-2', '1e-3', '1e-4', '1e-5', '1e-6', '1e-8'):
        _scale(scale)
        on = ts._sweep(make, True, V)
        row.append('{}: {}'.format(scale, int((on[2] != off[2]).sum())))
    _scale(None)
    print('float64 shifted lattice, {:24s} nodes {:5d}   index mismatches by radius scale   {}'.format(
        'box depends on x0' if box_on_state else 'constant box (table)', V.size, '   '.join(row)), flush=True)
