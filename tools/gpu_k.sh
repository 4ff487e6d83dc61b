#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/k; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('synth256', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
for cfg in coupled256 searev synth512f32; do timeout 600 python bench.py --config $cfg --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; python -c "
import json; d=json.load(open('$O/bench_$cfg.json')); print('$cfg', d['config']['kernel_family'], d['value'], d['ms_per_step'])"; done
timeout 600 python bench.py --config coupled256 --kernel staged --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_cs.json 2> $O/bench_cs.err; python -c "
import json; d=json.load(open('$O/bench_cs.json')); print('coupled staged', d['ms_per_step'])"
timeout 600 python bench.py --config synth256 --kernel staged --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err; python -c "
import json; d=json.load(open('$O/bench_ss.json')); print('synth256 staged', d['ms_per_step'])"
