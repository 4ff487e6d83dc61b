#!/usr/bin/env python3
"""Copy what tools/profile_all.sh left under gpurun_out/ into profiles/ (the committed evidence):
per config the summary, rocprofv3's kernel stats and the PMC json bench.py reads; the in-kernel
clock merged into profiles/clock.json; the probe outputs.  usage: python tools/collect_profiles.py [round tag]"""
import json
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
runs = ((tag + '_final', 'synth256_f64_column_filter'), (tag + '_nofilter', 'synth256_f64_column'),
        (tag + '_synth512f32', 'synth512f32_f32_column_filter'), (tag + '_searev', 'searev_f64_column_filter'),
        (tag + '_ar1', 'ar1_f64_column_filter'), (tag + '_coupled', 'coupled256_f64_column'),
        (tag + '_noisy', 'noisy256_f64_column_filter'), (tag + '_reservoirs', 'reservoirs_f64_lead_filter'))
for t, key in runs:
    d = os.path.join(root, 'gpurun_out', 'prof_' + t)
    if not os.path.isdir(d):
        print('missing', d)
        continue
    shutil.copy(os.path.join(d, 'summary.txt'), os.path.join(root, 'profiles', t + '_summary.txt'))
    shutil.copy(os.path.join(d, 'kernel_stats.csv'), os.path.join(root, 'profiles', t + '_kernel_stats.csv'))
    shutil.copy(os.path.join(d, 'pmc_' + key + '.json'), os.path.join(root, 'profiles', 'pmc_' + key + '.json'))
    j = json.load(open(os.path.join(d, 'pmc_' + key + '.json')))
    print(t, j.get('kernel_source_key'), round(j['avg_kernel_ms_trace_pass'], 4), 'ms')
clk = os.path.join(root, 'gpurun_out', tag + '_clock.json')
if os.path.exists(clk):
    new = json.load(open(clk))
    old = json.load(open(os.path.join(root, 'profiles', 'clock.json')))
    old['f64'], old['f32'], old['sweep_kernel_ghz'], old['round'] = new['f64'], new['f32'], new['sweep_kernel_ghz'], tag
    json.dump(old, open(os.path.join(root, 'profiles', 'clock.json'), 'w'), indent=1)
for f in (tag + '_filter_radius_probe.txt', tag + '_filter_probe.txt', tag + '_parity_report.txt',
          tag + '_ubench_valu_rate.txt', tag + '_fixed_cost_sharded.txt', tag + '_filter_probe_noisy.txt',
          tag + '_filter_ab_noisy.txt', tag + '_lead_ab.txt'):
    src = os.path.join(root, 'gpurun_out', f)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(root, 'profiles', f))
