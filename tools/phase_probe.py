#!/usr/bin/env python3
"""Where a workgroup of the column kernel spends its time (diagnostic SDP_STAMP=2 build):
shader clocks of thread 0 between the phase barriers -- W (trailing cells), A (table
build), B (node x control x perturbation loop, including the stores) -- summed over the
columns the workgroup takes, and the workgroup's lifetime.
usage: python tools/phase_probe.py      (through gpurun)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, DPSolver, _native as nat
# diagnostic build (clock stamps), and whatever other switches the command line names as K=V
DPSolver.debug_defines = dict([('SDP_STAMP', '2'), ('SDP_COL_WRES', '0')] + [a.split('=', 1) for a in sys.argv[1:] if a.startswith('SDP_') and '=' in a])
sys.argv = [a for a in sys.argv if not (a.startswith('SDP_') and '=' in a)]

_, s = models.synthetic3d(N=256, stock_noise=float(os.environ.get('SDP_STOCK_NOISE', 0)))
prob = s._problem()
prob.set_value(models.synthetic3d_V0(s.state_grid))
prob.bench_sweeps(30)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
_, k = prob.bench_sweeps(5)
st = np.zeros(65536 * 4, dtype=np.uint64)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), st.size))
st = st.reshape(-1, 4).astype(float)
n_wg = int((st[:, 3] > 0).sum())
inside = st[2 * n_wg:3 * n_wg, 0]
extra = st[n_wg:2 * n_wg]                 # filter builds: reduce, first pass, second pass of phase B
st = st[:n_wg]
busy = st[:, 2] > 0                       # (units are claimed in order: late workgroups find nothing left)
st, extra, inside = st[busy], extra[busy[:len(extra)]] if len(extra) else extra, inside[busy]
tot = st[:, 3]
print('kernel {:.3f} ms; {} workgroups; lifetime of a workgroup: median {:.3e} clk'.format(k / 5, len(st), np.median(tot)))
for name, col in (('W (trailing cells + barrier)', 0), ('A (table build + barrier)', 1), ('B (cells, argmin, stores + barrier)', 2)):
    print('{:40s} {:6.2f} % of the workgroup lifetime (median over workgroups; p10 {:.2f}, p90 {:.2f})'.format(
        name, 100 * np.median(st[:, col] / tot), 100 * np.percentile(st[:, col] / tot, 10),
        100 * np.percentile(st[:, col] / tot, 90)))
print('unaccounted (column bookkeeping, kernel prologue) {:.2f} %'.format(100 * np.median(1 - st[:, :3].sum(axis=1) / tot)))
if s.backend_info.get('certified_filter'):
    for name, col in (('B: reduction of the table over w + barrier', 0), ('B: first pass (bounds of every control)', 1),
                      ('B: second pass (survivors, merge, stores)', 2)):
        print('{:40s} {:6.2f} %'.format(name, 100 * np.median(extra[:, col] / tot)))
    if inside.any():
        print('{:40s} {:6.2f} %'.format('   of it: thread 0 inside the reduction', 100 * np.median(inside / tot)))
