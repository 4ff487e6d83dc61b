#!/usr/bin/env python3
"""Is the slower start of a chain of sweeps the CONTENT (a rough cost-to-go: more blocks survive the branch and bound) or the
CLOCK (a GPU that was idle)?  The same chain from the same V0 twice in a row, kernel time per chunk of 5 sweeps: what repeats
in the second run is content, what does not is the clock.      usage: python tools/ramp_probe.py      (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

_, s = models.synthetic3d(N=256)
V0 = models.synthetic3d_V0(s.state_grid)
prob = s._problem()
for run in range(3):
    prob.set_value(V0)
    out = []
    for chunk in range(10):
        if chunk:
            prob.swap()
        _, k = prob.bench_sweeps(5)
        out.append(k / 5)
    print('run', run, 'kernel ms per sweep, chunks of 5:', ' '.join('%.3f' % v for v in out), flush=True)
