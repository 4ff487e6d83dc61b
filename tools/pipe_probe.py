#!/usr/bin/env python3
"""Where the two halves of the pipelined filtered column kernel (csrc/sdp_column_pipe.h) spend
a step (diagnostic SDP_STAMP=2 build): shader clocks thread 0 (a consumer) and the first
producer thread are busy inside their half; the rest of a step is the wait at its barrier.
usage: python tools/pipe_probe.py [N] [dtype]      (through gpurun)"""
import ctypes as C
import os
import sys

os.environ['SDP_STAMP'] = '2'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, _native as nat

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype('float64')
_, s = models.synthetic3d(N=N)
s.dtype = dtype
prob = s._problem()
prob.set_value(models.synthetic3d_V0(s.state_grid, dtype))
prob.bench_sweeps(30)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
_, k = prob.bench_sweeps(5)
st = np.zeros(65536 * 4, dtype=np.uint64)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), st.size))
st = st.reshape(-1, 4).astype(float)
n_wg = int(np.nonzero(st[:, 3] > 0)[0].max() + 1) // 2
cons, prod = st[:n_wg], st[n_wg:2 * n_wg]
busy = prod[:, 1] > 0
cons, prod = cons[busy], prod[busy]
print('kernel {:.3f} ms (5 launches accumulate in the stamps); {} workgroups with work; steps per workgroup: median {:.0f}'.format(
    k / 5, len(cons), np.median(prod[:, 1])))
life = cons[:, 3]
print('lifetime of a workgroup: median {:.3e} clk = {:.0f} clk per step'.format(np.median(life), np.median(life / prod[:, 1])))
print('consumer thread 0: busy {:.1f} % of the lifetime (first pass {:.1f} %, second pass + stores {:.1f} %)'.format(
    100 * np.median(cons[:, 0] / life), 100 * np.median(cons[:, 1] / life), 100 * np.median(cons[:, 2] / life)))
print('first producer thread: busy {:.1f} % of the lifetime'.format(100 * np.median(prod[:, 0] / prod[:, 3])))
