#!/usr/bin/env python3
"""Vector-issue cost of the benchmark kernel BY INSTRUCTION CLASS: the code object is
disassembled (no GPU needed), its hot loops are recognised by what they contain, their static instruction mix is
multiplied by trip counts that follow from the problem's dimensions, and what the loops do not account for (the
per-unit straight-line code: PMC total minus the loops) is priced at the slowest class.  Classes and clocks per
wave64 instruction per SIMD with two or more waves resident (profiles/r03_ubench_valu_rate.txt):
    f64   fp64 arithmetic, min / max / compare / convert                         4.3
    vop3  32-bit work in the VOP3 encoding (v_med3, v_lshl_add, packed fp32 ..)  4.3
    v32   plain 32-bit VOP1 / VOP2 (v_mov_b32, v_add_u32, v_and_b32, ..)         2.4
    lane  v_readlane / v_writelane (SGPR spill traffic)                         4.3
(`v_cndmask_b32` reading vcc measured 22.8 clk in a stream of NOTHING BUT such instructions -- no producer of vcc
in sight -- but 4.0 next to the compare that feeds it (row "cmp_lt_f64 vcc + cndmask": 8.2 - 8.7 for the pair), which
is how the first pass uses it: priced 4.3 like every VOP2 that is not in the fast list.)

usage: python tools/issue_model.py [--pmc profiles/pmc_<key>.json] [--out profiles/issue_classes_<key>.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import kernel_blocks as kb      # noqa: E402

CLOCKS = {'f64': 4.3, 'vop3': 4.3, 'v32': 2.4, 'lane': 4.3}


def vclass(op):
    if op in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'):
        return 'lane'
    c = kb.issue_class(op)
    return c if c in ('f64', 'vop3', 'v32') else None


def mix_of(body):
    m = {}
    for x in body:
        c = vclass(x['op'])
        if c:
            m[c] = m.get(c, 0) + 1
    return m


def count(body, prefix):
    return sum(x['op'].startswith(prefix) for x in body)


def main():
    import numpy as np
    from stodynprog_amd import models, codegen, _native as nat
    args = sys.argv[1:]
    pmc_path = args[args.index('--pmc') + 1] if '--pmc' in args else os.path.join(ROOT, 'profiles', 'pmc_synth256_f64_column_filter.json')
    out_path = args[args.index('--out') + 1] if '--out' in args else os.path.join(ROOT, 'profiles', 'issue_classes_synth256_f64_column_filter.json')
    _, s = models.synthetic3d(N=256)
    plan = s._kernel_plan()
    key = codegen.source_key(plan['source'])
    ins = kb.disassemble(nat.compile_model(plan['source']), 'sdp_sweep_col')
    N0, W, U = 256, 32, 64
    src = plan['source']
    bnb = '#define SDP_COL_BNB 1' in src
    bnb_blocks = float(os.environ.get('SDP_BNB_BLOCKS_PER_WAVE', '1.49'))
    threads = int(src.split('#define SDP_COL_THREADS ')[1].split()[0])
    wres = int(src.split('#define SDP_COL_WRES ')[1].split()[0]) if '#define SDP_COL_WRES ' in src else W
    waves = threads // 64
    S = N0 ** 3
    units = S // N0
    loops = sorted({(x['target'], x['off']) for x in ins if x['target'] is not None and x['target'] <= x['off']})
    bodies = [[x for x in ins if lo <= x['off'] <= hi] for lo, hi in loops]
    found = {}

    def consider(kind, body, items, trips_total):
        """keep, per kind, the smallest body (the clone the benchmark's grid takes: axis [0, 1], ordinary lattice)"""
        if items <= 0:
            return
        per_item = len(body) / float(items)
        if kind not in found or per_item < found[kind]['per_item']:
            found[kind] = dict(per_item=per_item, body=body, items=items, trips_per_wave_unit=trips_total / float(items))

    tail = W - wres
    hold = '#define SDP_COL_TAIL_HOLD 1' in src                # round 6: the tail is built once (unrolled: no loop) and held in registers
    entries_per_unit = (wres if hold else W + tail) * N0     # table entries built per unit BY THE LOOP (without the hold: the tail twice)
    for body in bodies:
        if len(body) > 400:
            continue                                         # an outer loop
        n_ld = count(body, 'global_load_dwordx4') + count(body, 'global_load_dwordx2')
        n_cvt = count(body, 'v_cvt_i32_f64')
        n_r64 = count(body, 'ds_read_b64')
        n_r128 = count(body, 'ds_read_b128')
        n_r2 = count(body, 'ds_read2_b64')
        if n_cvt >= 4 and n_r128 + n_r2 >= n_cvt and not count(body, 'v_div_scale_f64') and not n_ld:
            # controls per wave-node: all of them, or -- branch and bound, SDP_COL_BNB -- those of the blocks a wave does
            # evaluate (tools/bnb_count.py: 1.49 blocks of 8 per wave on this problem)
            consider('first pass', body, n_cvt, bnb_blocks * 8 if bnb else U)
        elif n_ld >= 4 and count(body, 'ds_write') >= 1:
            # 2^(d-1) = 4 vertex loads of 16 bytes (two rows) per pair of entries
            consider('table build', body, n_ld, entries_per_unit * 4 / 2.0 / 64 / waves)
        elif n_r64 >= 8 and not n_ld and not n_cvt and count(body, 'v_max_f64') >= n_r64 // 2:
            consider('reduction over w', body, n_r64, W)                             # a row per thread, W entries
        elif n_r64 >= 8 and not n_ld and not n_cvt and count(body, 'v_mul_f64') >= n_r64:
            consider('second pass', body, n_r64 // 2, W)                             # two reads per perturbation point
    per_wave_unit = {}
    table = {}
    for kind, f in found.items():
        m = mix_of(f['body'])
        table[kind] = dict(static_mix_per_trip=m, items_per_trip=f['items'], trips_per_wave_unit=f['trips_per_wave_unit'],
                           instructions_per_trip=len(f['body']))
        for c, n in m.items():
            per_wave_unit[c] = per_wave_unit.get(c, 0.0) + n * f['trips_per_wave_unit']
    launches = units * waves
    loops_total = {c: n * launches for c, n in per_wave_unit.items()}
    out = dict(kernel='sdp_sweep_col', kernel_source_key=key, clocks_per_wave_instruction=CLOCKS, branch_and_bound=bnb,
               first_pass_blocks_per_wave=(bnb_blocks if bnb else None),
               grid=dict(N0=N0, W=W, controls=U, threads=threads, resident_points=wres, units=units), loops=table,
               loops_wave_instructions_per_launch=loops_total)
    try:
        with open(pmc_path) as f:
            pmc = json.load(f)
        total = float(pmc['counters_mean_per_dispatch']['SQ_INSTS_VALU'])
        out['pmc_source'] = os.path.relpath(pmc_path, ROOT)
        out['pmc_kernel_source_key'] = pmc.get('kernel_source_key')
        out['SQ_INSTS_VALU_per_launch'] = total
        rest = total - sum(loops_total.values())
        out['outside_the_loops_wave_instructions'] = rest
        out['outside_priced_at'] = CLOCKS['f64']
        cycles = sum(CLOCKS[c] * n for c, n in loops_total.items()) + CLOCKS['f64'] * max(rest, 0.0)
        out['issue_cycles_per_launch'] = cycles
        out['issue_cycles_uniform_4clk'] = 4.0 * total
    except (OSError, KeyError, ValueError) as e:
        out['pmc_error'] = str(e)
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'loops'}, indent=1))
    for kind, t in table.items():
        print('{:18s} {:4d} instructions per trip of {:3d} items, {:7.2f} trips per wave and unit, mix {}'.format(
            kind, t['instructions_per_trip'], t['items_per_trip'], t['trips_per_wave_unit'], t['static_mix_per_trip']))


if __name__ == '__main__':
    main()
