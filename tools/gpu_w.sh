#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/w; mkdir -p $O
SDP_COL_A_ORDER=2 timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "--config searev" "--dtype float32" "--config synth512f32"; do
  for k in "SDP_COL_A_ORDER=0" "SDP_COL_A_ORDER=2" "SDP_COL_A_ORDER=0" "SDP_COL_A_ORDER=2"; do
    TUNE_BENCH_ARGS="$cfg" python tools/tune.py "$k" 2>&1 | sed "s/^/$cfg  /"
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
