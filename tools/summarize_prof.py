#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) of tools/profile_bench.sh
into a short text summary and a machine-readable pmc_<key>.json -- both are
committed under profiles/; bench.py reads the JSON for its roofline block.

usage: summarize_prof.py <prof dir> [pmc key] [tag]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else None
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(root.rstrip('/'))
KERNEL = 'sdp_sweep'          # sdp_sweep, sdp_sweep_col, sdp_sweep_lds


def rows(pattern):
    for path in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield path, r


print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for path, r in rows('trace/**/*kernel_stats.csv'):
    print('{:60s} calls={:>5s} total_ns={:>14s} avg_ns={:>14s} pct={}'.format(
        r.get('Name', '')[:60], r.get('Calls', ''), r.get('TotalDurationNs', ''),
        r.get('AverageNs', ''), r.get('Percentage', '')))

print('\n== per-dispatch durations of {}* (kernel trace) =='.format(KERNEL))
durs = []
last = None
kname = None
for path, r in rows('trace/**/*kernel_trace.csv'):
    if KERNEL in r.get('Kernel_Name', ''):
        durs.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        last = r
        kname = r['Kernel_Name']
if durs:
    print('n={} ms: {}'.format(len(durs), ' '.join('%.2f' % d for d in durs)))
    print('VGPR={} AGPR={} SGPR={} LDS={} scratch={} grid={} wg={}'.format(
        last.get('VGPR_Count', '?'), last.get('Accum_VGPR_Count', '?'), last.get('SGPR_Count', '?'),
        last.get('LDS_Block_Size', '?'), last.get('Scratch_Size', '?'),
        last.get('Grid_Size', last.get('Grid_Size_X', '?')),
        last.get('Workgroup_Size', last.get('Workgroup_Size_X', '?'))))

print('\n== PMC counters per {}* dispatch =='.format(KERNEL))
acc = defaultdict(list)
for path, r in rows('pmc_*/**/*counter_collection.csv'):
    if KERNEL in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
mean = {}
for name, vals in sorted(acc.items()):
    mean[name] = sum(vals) / len(vals)
    print('{:24s} n={:3d} mean={:.6g} min={:.6g} max={:.6g}'.format(
        name, len(vals), mean[name], min(vals), max(vals)))
if 'FETCH_SIZE' in mean:
    f = mean['FETCH_SIZE']
    print('FETCH_SIZE is in KiB; gfx950 reports 1/2 of wide streaming reads '
          '(MI355X_MICROARCH.md HBM section): raw {:.4g} MB, x2 {:.4g} MB per launch'.format(
              f * 1024 / 1e6, 2 * f * 1024 / 1e6))
if 'WRITE_SIZE' in mean:
    print('WRITE_SIZE {:.4g} MB per launch'.format(mean['WRITE_SIZE'] * 1024 / 1e6))
timed = durs[2:] if len(durs) > 2 else durs        # bench.py: 2 warm-up launches first
avg_ms = sum(timed) / len(timed) if timed else None
f64 = [mean.get('SQ_INSTS_VALU_{}_F64'.format(k)) for k in ('ADD', 'MUL', 'FMA')]
if all(v is not None for v in f64) and avg_ms:
    n = sum(f64)
    print('fp64 VALU wave-instructions per launch: {:.6g} (ADD {:.4g} + MUL {:.4g} + FMA {:.4g}); '
          '/ {:.3f} ms = {:.4g} /s = {:.3f} of the 6.144e11 /s spec issue rate'.format(
              n, f64[0], f64[1], f64[2], avg_ms, n / (avg_ms * 1e-3), n / (avg_ms * 1e-3) / 6.144e11))
if 'GRBM_GUI_ACTIVE' in mean and avg_ms:
    print('GRBM_GUI_ACTIVE / 8 XCD / kernel time (trace pass duration): {:.3f} GHz (reads high on short '
          'dispatches; the in-kernel figure is profiles/clock.json)'.format(
              mean['GRBM_GUI_ACTIVE'] / 8 / (avg_ms * 1e-3) / 1e9))

if key:
    out = {'tag': tag, 'kernel': kname, 'avg_kernel_ms_trace_pass': avg_ms, 'dispatches': len(durs),
           'counters_mean_per_dispatch': mean,
           'command': 'rocprofv3 --pmc <counters> -- python3 bench.py '
                      '(tools/profile_bench.sh {} {})'.format(tag, key)}
    if all(v is not None for v in f64) and sum(f64) > 0 and 'f64' in key:
        out['valu_wave_instr'] = sum(f64)
        out['valu_wave_instr_counters'] = 'SQ_INSTS_VALU_ADD_F64 + _MUL_F64 + _FMA_F64'
    elif 'SQ_INSTS_VALU' in mean:
        out['valu_wave_instr'] = mean['SQ_INSTS_VALU']
        out['valu_wave_instr_counters'] = 'SQ_INSTS_VALU (all vector ALU instructions)'
    # the version of the kernel the counts belong to (bench.py prints it; it marks the counts stale
    # when the kernel sources change afterwards)
    for bj in sorted(glob.glob(os.path.join(root, 'bench_*.json'))):
        try:
            with open(bj) as f:
                key_ = json.loads(f.read().strip().splitlines()[-1])['roofline'].get('kernel_source_key')
        except (OSError, ValueError, KeyError, IndexError):
            key_ = None
        if key_:
            out['kernel_source_key'] = key_
            break
    if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
        out['hbm_bytes'] = (2 * mean['FETCH_SIZE'] + mean['WRITE_SIZE']) * 1024
        out['hbm_bytes_formula'] = ('(2 x FETCH_SIZE + WRITE_SIZE) KiB: separate --pmc passes; FETCH_SIZE '
                                    'doubled per the gfx950 correction of MI355X_MICROARCH.md (it tallies '
                                    '128-B requests at 64 B)')
    with open(os.path.join(root, 'pmc_{}.json'.format(key)), 'w') as f:
        json.dump(out, f, indent=1)
        f.write('\n')
