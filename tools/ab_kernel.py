#!/usr/bin/env python3
"""Same-box A/B of kernel variants of one bench configuration: every variant is a set of K=V switches of the generated
kernels (codegen.DEBUG_NAMES); the variants are timed in interleaved rounds in ONE process (devices differ by a few per
cent, and a chip warms up), and every variant's J and policy index after the chain of sweeps are compared with the
first variant's over all nodes.
usage: python tools/ab_kernel.py [--config synth256|synth512f32|noisy256|searev|ar1|ar1_ref|coupled256|reservoirs] [--rounds R] [--sweeps K] -- [K=V ..] -- [K=V ..] ...
       (an empty group is the default kernel)                                          (through gpurun)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stodynprog_amd import models


def build(config, defs):
    if config == 'synth256':
        _, s = models.synthetic3d(N=256)
        V0 = models.synthetic3d_V0(s.state_grid)
    elif config == 'noisy256':
        _, s = models.synthetic3d(N=256, stock_noise=0.07)
        V0 = models.synthetic3d_V0(s.state_grid)
    elif config == 'synth512f32':
        _, s = models.synthetic3d(N=512)
        s.dtype = np.dtype('float32')
        V0 = models.synthetic3d_V0(s.state_grid, np.float32)
    elif config == 'coupled256':                           # trailing axes that see the control: a table per control
        _, s = models.synthetic3d_coupled(N=256)
        V0 = models.synthetic3d_V0(s.state_grid)
    elif config == 'reservoirs':                           # two controlled stocks: the reduced-array sweep
        _, s = models.two_reservoirs(n_a=128, n_b=128, n_y=64, n_w=16, steps=(1. / 15, 1. / 15))
        V0 = np.random.default_rng(0).standard_normal(s._state_grid_shape)
    elif config == 'searev':                               # BASELINE configs[2] as bench.py --config searev
        _, s = models.searev(n_E=128, n_S=128, n_A=128, step=2.2 / 31)
        V0 = np.random.default_rng(0).standard_normal(s._state_grid_shape)
    elif config == 'ar1':                                  # BASELINE configs[1]
        _, s = models.storage_ar1(n_E=200, n_P=200, steps=(8. / 49, 0.1))
        V0 = np.random.default_rng(0).standard_normal(s._state_grid_shape)
    elif config == 'ar1_ref':                              # the reference's published case
        _, s = models.storage_ar1()
        V0 = np.random.default_rng(0).standard_normal(s._state_grid_shape)
    else:
        raise SystemExit('unknown config ' + config)
    s.debug_defines = dict(kv for kv in defs if kv[0] != '(again)') or None
    prob = s._problem()
    prob.set_value(V0)
    return s, prob


def main():
    argv = sys.argv[1:]
    head = argv[:argv.index('--')] if '--' in argv else argv
    rest = argv[len(head):]
    config, rounds, sweeps, warm = 'synth256', 5, 10, 3
    it = iter(head)
    for a in it:
        if a == '--config': config = next(it)
        elif a == '--rounds': rounds = int(next(it))
        elif a == '--sweeps': sweeps = int(next(it))
        elif a == '--warm': warm = int(next(it))
    groups, cur = [], None
    for a in rest:
        if a == '--':
            if cur is not None: groups.append(cur)
            cur = []
        else:
            cur.append(tuple(a.split('=', 1)))
    if cur is not None: groups.append(cur)
    if not groups: groups = [[]]
    # The first variant is built FIRST and has measured 2-4 % slower than an identical one built later (round 6, box 20:
    # where its buffers lie, presumably) -- so it is built once more at the end: its two rows bracket what position alone does.
    if len(groups) > 1:
        groups.append(list(groups[0]) + [('(again)', '')])
    probs = []
    for g in groups:
        s, p = build(config, g)
        p.bench_sweeps(max(warm, 1))
        probs.append((g, s, p))
    times = [[] for _ in probs]
    for r in range(rounds if warm > 0 else 0):
        for k, (g, s, p) in enumerate(probs):
            p.swap()                                   # (go on from the last J: a call starts on the buffers it finds)
            _, kern = p.bench_sweeps(sweeps)
            times[k].append(kern / sweeps)
    ref = None
    for k, (g, s, p) in enumerate(probs):
        # every problem has run the same number of sweeps from the same V0 (bench_sweeps ping-pongs V and J)
        J = p.get_value()
        _, idx = p.get_policy()
        if ref is None:
            ref = (J, idx)
            same = 'reference'
        else:
            same = 'J identical: {}  index identical: {}'.format(np.array_equal(J, ref[0]), np.array_equal(idx, ref[1]))
            if not np.array_equal(J, ref[0]):
                bad = J != ref[0]
                where = np.argwhere(bad)
                same += '  ({} entries differ, largest {:.3e}; first at {}, last at {})'.format(
                    int(bad.sum()), float(np.abs(J - ref[0])[bad].max()), where[0].tolist(), where[-1].tolist())
        t = np.array(times[k])
        print('{:60s} kernel min {:.4f}  median {:.4f}  max {:.4f} ms   {}'.format(
            ' '.join('='.join(kv).rstrip('=') for kv in g) .replace('(again)', '(the first variant again)') or '(default)', t.min(), np.median(t), t.max(), same), flush=True)


if __name__ == '__main__':
    main()
