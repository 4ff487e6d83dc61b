#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/y; mkdir -p $O
rm -f $O/parity_report.txt
SDP_PARITY_REPORT=$PWD/$O/parity_report.txt timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
sort -u $O/parity_report.txt > $O/parity_report_sorted.txt
timeout 900 bash tools/profile_bench.sh r02_final synth256_f64_column > $O/prof.log 2>&1; grep -E "ms:|fp64 VALU|VGPR" $O/prof.log
timeout 300 python tools/clock_probe.py $O/clock.json > $O/clock.log 2>&1; tail -2 $O/clock.log
timeout 300 python tools/phase_probe.py > $O/phase.txt 2>&1; cat $O/phase.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
timeout 600 python tools/host_rate.py > $O/host_rate.txt 2>&1; head -1 $O/host_rate.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
