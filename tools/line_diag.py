#!/usr/bin/env python3
"""How the filtered line kernel decides its nodes (diagnostic SDP_LINE_DIAG build of csrc/sdp_line_kernel.h).  (through gpurun)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import _native as nat
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'line_check.py')).read().split("sizes = ")[0]
ns = {'__file__': __file__}
exec(compile(src, 'lc', 'exec'), ns)
sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(65536, 4097, 16)]
for (n_x, n_u, n_w) in sizes:
    for smooth in (True, False):
        s = ns['inventory'](n_x, n_u, n_w, smooth)
        s.kernel = 'line'
        s.debug_defines = {'SDP_EXTRA_DEFINES': 'SDP_LINE_DIAG=1'}
        x = np.linspace(-8, 24, n_x)
        prob = s._problem()
        prob.set_value(0.3 * (x - 3) ** 2 + np.sin(x))
        for k in range(4):
            if k: prob.swap()
            prob.bench_sweeps(1)
        nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
        prob.swap()
        _, kern = prob.bench_sweeps(1)
        st = np.zeros(16, dtype=np.uint64)
        nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), st.size))
        names = ['single1', 'pair1', 'undecided1', 'single2', 'pair2', 'left2', 'bad', 'evals2', 'long way']
        print(n_x, n_u, n_w, 'smooth' if smooth else 'kinked', 'kernel %.3f ms' % kern, dict(zip(names, st[:9].tolist())), flush=True)
