#!/bin/bash
# round-2 GPU batch A: regression tests, clock probe, ubench, bench line, profile
export TMPDIR=/tmp
O=gpurun_out/a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/clock_probe.py $O/clock.json > $O/clock.log 2>&1; tail -3 $O/clock.log
(cd tools/ubench && timeout 120 ./fp64_rate) > $O/ubench.txt 2>&1; cat $O/ubench.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
timeout 900 bash tools/profile_bench.sh r02_base synth256_f64_column > $O/prof.log 2>&1; tail -30 $O/prof.log
rocprofv3 -L > $O/counters.txt 2>&1 || true
