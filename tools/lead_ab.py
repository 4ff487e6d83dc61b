#!/usr/bin/env python3
"""Two controlled stocks next to an exogenous inflow (models.two_reservoirs): the reduced-array sweep
(csrc/sdp_lead_kernel.h) against the other kernel families on the same problem -- kernel time per sweep
and J / policy index bit for bit.  usage: python tools/lead_ab.py [n_a n_b n_y n_w n_u]  (through gpurun)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models

n_a, n_b, n_y, n_w, n_u = (int(v) for v in (sys.argv[1:6] + ['128', '128', '64', '16', '16'][len(sys.argv) - 1:]))
out = {}
for kernel, env in (('lead', {}), ('staged', {}), ('column', {}), ('generic', {})):
    if kernel == 'generic' and os.environ.get('LEAD_AB_SKIP_GENERIC'):
        continue
    if os.environ.get('LEAD_AB_ONLY') and kernel not in os.environ['LEAD_AB_ONLY'].split(','):
        continue
    _, s = models.two_reservoirs(n_a=n_a, n_b=n_b, n_y=n_y, n_w=n_w, steps=(1.0 / (n_u - 1), 1.0 / (n_u - 1)))
    s.kernel = kernel
    a, b, y = [np.asarray(g) for g in s.state_grid]
    V0 = ((a[:, None, None] - 1.0) ** 2 + 0.5 * (b[None, :, None] - 0.7) ** 2
          + 0.3 * np.cos(3 * y)[None, None, :] * (1 + 0.1 * a[:, None, None]))
    try:
        prob = s._problem()
    except Exception as e:
        print('{:8s} not available: {}'.format(kernel, str(e)[:100]))
        continue
    prob.set_value(V0)
    prob.bench_sweeps(2)
    prob.swap()
    n = 5
    _, kern = prob.bench_sweeps(n)
    J = prob.get_value()
    _, idx = prob.get_policy()
    out[kernel] = (J, idx)
    info = s.backend_info
    print('{:8s} {:9.3f} ms per sweep   kernel family {} {}'.format(
        kernel, kern / n, info['kernel'], 'table per control' if info.get('table_per_control') else (info.get('filter_form') or '')), flush=True)
    for k_ in [k_ for k_ in s._cache if k_[0] == 'problem']:
        s._cache.pop(k_).close()
ref = out.get('generic') or out.get('staged')
for k, (J, idx) in out.items():
    print('{:8s} J identical to the {} kernel: {}   index identical: {}'.format(
        k, 'direct' if 'generic' in out else 'staged', np.array_equal(J, ref[0]), np.array_equal(idx, ref[1])))
print('{} x {} x {} nodes, {} x {} controls, {} perturbation points: {:.3g} lattice cells per sweep'.format(
    n_a, n_b, n_y, n_u, n_u, n_w, float(n_a) * n_b * n_y * n_u * n_u * n_w))
