#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/l; mkdir -p $O
timeout 900 bash tools/profile_bench.sh r02_final synth256_f64_column > $O/prof.log 2>&1; tail -34 $O/prof.log
timeout 300 python tools/clock_probe.py $O/clock.json > $O/clock.log 2>&1; tail -3 $O/clock.log
for i in 1 2 3; do timeout 600 python bench.py --config searev --no-cpu-baseline --steps 20 --warmup 5 > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('searev', d['roofline']['kernel_ms'])"; done
timeout 600 python tools/host_rate.py > $O/host_rate.txt 2>&1; cat $O/host_rate.txt
timeout 600 python tools/fixed_cost.py > $O/fixed_cost.txt 2>&1; cat $O/fixed_cost.txt
