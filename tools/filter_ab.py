"""A/B of the certified expectation-first filter of the column kernel (SdpColFilter in
csrc/sdp_colfilter_kernel.h): same problem with DPSolver.certified_filter on / off -- J, policy
index bit for bit, and kernel time per sweep.  Usage: python tools/filter_ab.py [N] [dtype]
(SDP_STOCK_NOISE=c in the environment: the perturbation also reaches the stock, x0' = (x0 + b u) - c w)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dtype = np.dtype(sys.argv[2]) if len(sys.argv) > 2 else np.dtype('float64')
    out = {}
    for flt in (False, True):
        sysd, s = models.synthetic3d(N=N, stock_noise=float(os.environ.get('SDP_STOCK_NOISE', 0)))
        s.dtype = dtype
        s.certified_filter = flt
        V0 = models.synthetic3d_V0(s.state_grid, dtype)
        prob = s._problem()
        prob.set_value(V0)
        prob.bench_sweeps(3)
        prob.swap()
        loop, kern = prob.bench_sweeps(10)
        J = prob.get_value()
        pol, idx = prob.get_policy()
        out[flt] = (J, idx, kern / 10)
        print('certified_filter={}: kernel {:.3f} ms per sweep   ({})'.format(
            flt, kern / 10, {k: s.backend_info[k] for k in ('kernel', 'certified_filter', 'filter_form')}), flush=True)
    (Ja, ia, ta), (Jb, ib, tb) = out[False], out[True]
    print('J identical: {}   index identical: {}   speed-up {:.2f}x'.format(
        np.array_equal(Ja, Jb), np.array_equal(ia, ib), ta / tb))


if __name__ == '__main__':
    main()
