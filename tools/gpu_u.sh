#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/u; mkdir -p $O
rm -f $O/parity_report.txt
SDP_PARITY_REPORT=$PWD/$O/parity_report.txt timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
sort -u $O/parity_report.txt > $O/parity_report_sorted.txt
timeout 900 bash tools/profile_bench.sh r02_final synth256_f64_column > $O/prof.log 2>&1; tail -8 $O/prof.log
timeout 300 python tools/clock_probe.py $O/clock.json > $O/clock.log 2>&1; tail -2 $O/clock.log
timeout 300 python tools/phase_probe.py > $O/phase.txt 2>&1; cat $O/phase.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
