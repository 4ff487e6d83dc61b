#!/usr/bin/env python3
"""In-kernel clock of the sweep kernel on the benchmark problem (diagnostic
build, never the production code object).

MI355X_MICROARCH.md, DVFS give-back item 6: clock = d s_memtime / d s_memrealtime
x 100 MHz, stamped once around the kernel body after >= 2 s of back-to-back
launches on real data; median over workgroups.  The stamps go to a buffer of
their own (sdp_problem_debug_stamps); no output value depends on them.

usage: python tools/clock_probe.py [out.json]      (run through gpurun)
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models, DPSolver, _native as nat
# diagnostic build (clock stamps), and whatever other switches the command line names as K=V
DPSolver.debug_defines = dict([('SDP_STAMP', '1')] + [a.split('=', 1) for a in sys.argv[1:] if a.startswith('SDP_') and '=' in a])
sys.argv = [a for a in sys.argv if not (a.startswith('SDP_') and '=' in a)]


def probe(dtype, N=256, warm_s=2.5):
    _, s = models.synthetic3d(N=N)
    s.dtype = np.dtype(dtype)
    prob = s._problem()
    assert s.backend_info['kernel'] == 'column'
    prob.set_value(models.synthetic3d_V0(s.state_grid, dtype))
    _, k = prob.bench_sweeps(5)
    reps = max(20, int(warm_s * 1e3 / (k / 5)))
    t0 = time.perf_counter()
    prob.bench_sweeps(reps)                         # >= 2 s back to back, unstamped buffer absent
    warm = time.perf_counter() - t0
    nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, None, 0))
    _, k = prob.bench_sweeps(10)
    words = 65536 * 4
    st = np.zeros(words, dtype=np.uint64)
    nat.check(nat.lib().sdp_problem_debug_stamps(prob.h, 1, st.ctypes.data_as(C.c_void_p), words))
    st = st.reshape(-1, 4)
    st = st[(st[:, 1] != 0) & (st[:, 3] > st[:, 1])]
    ghz = (st[:, 2] - st[:, 0]).astype(float) / (st[:, 3] - st[:, 1]).astype(float) * 0.1
    span_ms = (st[:, 3].max() - st[:, 1].min()) / 1e5          # 100 MHz ticks -> ms
    res = dict(dtype=np.dtype(dtype).name, grid=N, workgroups=int(len(ghz)),
               ghz_median=float(np.median(ghz)), ghz_p10=float(np.percentile(ghz, 10)),
               ghz_p90=float(np.percentile(ghz, 90)), warm_seconds=warm,
               stamped_kernel_ms=k / 10, kernel_span_from_stamps_ms=float(span_ms))
    prob.close()
    return res


if __name__ == '__main__':
    out = {'method': 's_memtime / s_memrealtime x 100 MHz, thread 0 of every workgroup, kernel entry -> exit, '
                     'median over workgroups, after >= 2 s of back-to-back sweeps (diagnostic SDP_STAMP build)',
           'device': nat.device_info(0)}
    out['f64'] = probe('float64')
    out['f32'] = probe('float32')
    out['sweep_kernel_ghz'] = out['f64']['ghz_median']
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as f:
            f.write(text + '\n')
