#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/d; mkdir -p $O
rm -f $O/parity_report.txt
SDP_PARITY_REPORT=$PWD/$O/parity_report.txt timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
sort -u $O/parity_report.txt > $O/parity_report_sorted.txt; cat $O/parity_report_sorted.txt
timeout 600 python tools/host_rate.py > $O/host_rate.txt 2>&1; cat $O/host_rate.txt
for cfg in ar1 searev synth512f32; do
  timeout 900 python bench.py --config $cfg --steps 10 --warmup 2 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; python -c "
import json; d=json.load(open('$O/bench_$cfg.json')); print('$cfg', d.get('value'), d.get('ms_per_step'), d.get('roofline',{}).get('kernel_ms'), d.get('cpu_baseline',{}).get('value'), d.get('error'))"; tail -1 $O/bench_$cfg.err
done
