#!/usr/bin/env python3
"""Where a WAVE of the resident-chunk column kernel spends its time (diagnostic SDP_STAMP=2 build of
csrc/sdp_colres_kernel.h): shader clocks between the top of a unit and the barrier that completes the reduced table
(three table builds, the reductions, their barriers), in the first pass (every lane's controls: branch and bound or
the full pass), and in the rest (second pass, rebuild of the tail, stores), summed over the units of the wave.
usage: python tools/colres_probe.py [K=V switches of the generated kernels ...]      (through gpurun)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, DPSolver, _native as nat
DPSolver.debug_defines = dict([('SDP_STAMP', '2')] + [a.split('=', 1) for a in sys.argv[1:] if '=' in a])
_, s = models.synthetic3d(N=256)
prob = s._problem()
prob.set_value(models.synthetic3d_V0(s.state_grid))
prob.bench_sweeps(30)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
_, k = prob.bench_sweeps(5)
st = np.zeros(65536 * 4, dtype=np.uint64)
nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), st.size))
st = st.reshape(-1, 4).astype(float)
st = st[st[:, 3] > 0]
tot = st[:, 3]
print('{}: kernel {:.3f} ms; {} waves; lifetime of a wave: median {:.3e} clk'.format(
    ' '.join(sys.argv[1:]) or '(default)', k / 5, len(st), np.median(tot)))
rest = tot - st[:, 0] - st[:, 1] - st[:, 2]
for name, v in (('waiting at the barrier at the top of a unit', st[:, 2]), ('table builds, reductions, their barriers', st[:, 0]),
                ('first pass', st[:, 1]), ('second pass, rebuild of the tail, next unit\'s tables, stores', rest)):
    print('  {:62s} {:6.2f} % of the lifetime   (median {:.3e} clk per wave)'.format(name, 100 * np.median(v / tot), np.median(v)))
for w in range(4):
    sel = st[w::4]
    print('  wave {} of its workgroup: top {:.3e}  builds {:.3e}  first pass {:.3e}  rest {:.3e}'.format(
        w, np.median(sel[:, 2]), np.median(sel[:, 0]), np.median(sel[:, 1]), np.median(sel[:, 3] - sel[:, 0] - sel[:, 1] - sel[:, 2])))
