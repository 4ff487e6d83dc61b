"""Blocks of controls a node asks its branch and bound for (diagnostic build SDP_DIAG_BNB_COUNT: J := the count + 100 x the
guess's block).  usage: python tools/bnb_count.py [stock_noise [sweeps]]      (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, DPSolver
DPSolver.debug_defines = {'SDP_EXTRA_DEFINES': 'SDP_DIAG_BNB_COUNT=1'}
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_, s = models.synthetic3d(N=256, stock_noise=noise) if noise else models.synthetic3d(N=256)
V0 = models.synthetic3d_V0(s.state_grid)
if sweeps > 1:                                            # (the chain's earlier sweeps with the product kernel)
    DPSolver.debug_defines = None
    _, s0 = models.synthetic3d(N=256, stock_noise=noise) if noise else models.synthetic3d(N=256)
    V0, _ = s0.value_iterations(V0, sweeps - 1, report_time=False)
    DPSolver.debug_defines = {'SDP_EXTRA_DEFINES': 'SDP_DIAG_BNB_COUNT=1'}
J, pol = s.value_iterations(V0, 1, report_time=False)
idx = s.last_policy_index
cnt = np.round(J).astype(int)
need, bg = cnt % 100, cnt // 100
print('need: mean %.3f, hist' % need.mean(), np.bincount(need.ravel(), minlength=9)[:9])
print('guess block hist', np.bincount(bg.ravel(), minlength=8)[:8], ' best block hist', np.bincount((idx // 8).ravel(), minlength=8)[:8])
print('guess block == best block: %.3f' % (bg == idx // 8).mean())
w = need.reshape(256, -1)            # rows = axis 0 (lanes), columns
per_wave = w.reshape(4, 64, -1).max(axis=1)
print('max over the 64 lanes of a wave: mean %.3f' % per_wave.mean(), np.bincount(per_wave.ravel(), minlength=9)[:9])
for k in range(4):
    print('wave', k, ': mean of the largest count of its lanes %.3f' % per_wave[k].mean(), ' mean count per lane %.3f' % w.reshape(4, 64, -1)[k].mean())
