#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text
summary that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


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

print('\n== per-dispatch durations of sdp_sweep (kernel trace) ==')
durs = []
for path, r in rows('trace/**/*kernel_trace.csv'):
    if 'sdp_sweep' in r.get('Kernel_Name', ''):
        durs.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        last = r
if durs:
    print('n={} ms: {}'.format(len(durs), ' '.join('%.2f' % d for d in durs)))
    print('VGPR={} SGPR={} LDS={} scratch={} grid={} wg={}'.format(
        last.get('VGPR_Count', last.get('Arch_VGPR_Count', '?')), last.get('SGPR_Count', '?'),
        last.get('LDS_Block_Size', '?'), last.get('Scratch_Size', '?'),
        last.get('Grid_Size', '?'), last.get('Workgroup_Size', '?')))

print('\n== PMC counters per sdp_sweep dispatch ==')
acc = defaultdict(list)
for path, r in rows('pmc_*/**/*counter_collection.csv'):
    if 'sdp_sweep' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for name, vals in sorted(acc.items()):
    print('{:24s} n={:3d} mean={:.6g} min={:.6g} max={:.6g}'.format(
        name, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
if 'FETCH_SIZE' in acc:
    f = sum(acc['FETCH_SIZE']) / len(acc['FETCH_SIZE'])
    print('FETCH_SIZE is in KiB; gfx950 reports 1/2 of wide streaming reads '
          '(MI355X_MICROARCH.md HBM section): raw {:.4g} MB, x2 {:.4g} MB per launch'.format(
              f * 1024 / 1e6, 2 * f * 1024 / 1e6))
if 'WRITE_SIZE' in acc:
    w = sum(acc['WRITE_SIZE']) / len(acc['WRITE_SIZE'])
    print('WRITE_SIZE {:.4g} MB per launch'.format(w * 1024 / 1e6))
