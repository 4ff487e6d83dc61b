#!/usr/bin/env python3
"""What the division of the axis-0 position costs the storage-AR1 kernels: the same model with a span of 10 (a true division per
control, pyx:75) and of 8 (a power of two: the product with the reciprocal is bit-identical).  Round 6: 0.318 against 0.267 ms at the
reference's size -- with BOTH passes free of it; a first pass on the reciprocal alone (a position term in the radius) would gain
less, and was not built.      usage: python tools/div_probe.py      (through gpurun)"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models
for E in (10., 8.):
    for kw in (dict(), dict(n_E=200, n_P=200, steps=(8. / 49, 0.1))):
        _, s = models.storage_ar1(E_rated=E, **kw)
        V0 = np.random.default_rng(0).standard_normal(s._state_grid_shape)
        p = s._problem(); p.set_value(V0); p.bench_sweeps(5)
        ts = []
        for r in range(5):
            p.swap(); _, k = p.bench_sweeps(20); ts.append(k / 20)
        print('E_rated', E, kw and 'config 2' or 'reference size', 'kernel ms median %.4f' % np.median(ts), flush=True)
