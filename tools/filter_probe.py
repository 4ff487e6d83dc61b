#!/usr/bin/env python3
"""How selective the certified filter of the column kernel is (diagnostic SDP_STAMP=3 build):
nodes that keep more than one control after the first pass, and controls evaluated with the
reference's operations, per sweep.  usage: python tools/filter_probe.py [N] [sweeps] [dtype]  (through gpurun)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, DPSolver, _native as nat
# diagnostic build (clock stamps), and whatever other switches the command line names as K=V
DPSolver.debug_defines = dict([('SDP_STAMP', '3'), ('SDP_COL_WRES', '0')] + [a.split('=', 1) for a in sys.argv[1:] if a.startswith('SDP_') and '=' in a])
sys.argv = [a for a in sys.argv if not (a.startswith('SDP_') and '=' in a)]

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dtype = np.dtype(sys.argv[3]) if len(sys.argv) > 3 else np.dtype('float64')
_, s = models.synthetic3d(N=N, stock_noise=float(os.environ.get('SDP_STOCK_NOISE', 0)))
s.dtype = dtype
prob = s._problem()
assert s.backend_info['certified_filter']
prob.set_value(models.synthetic3d_V0(s.state_grid, dtype))
for k in range(sweeps):
    nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 0, None, 0))
    nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
    prob.bench_sweeps(1)
    st = np.zeros(4, dtype=np.uint64)
    nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), st.size))
    slow, exact, nodes = (int(v) for v in st[:3])
    print('sweep {}: {} nodes, {} keep more than one control ({:.4f} %), {} exact evaluations '
          '(incl. one per lane of a node: {:.3f} per node)'.format(
              k, nodes, slow, 100.0 * slow / max(nodes, 1), exact, exact / max(nodes, 1)))
    prob.swap()
