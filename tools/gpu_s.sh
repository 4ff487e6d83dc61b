#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/s; mkdir -p $O
for cfg in "--dtype float32" "--config searev" "--config synth512f32"; do
  for k in "SDP_COL_BATCH=2" "SDP_COL_BATCH=4" "SDP_COL_BATCH=2" "SDP_COL_BATCH=4"; do
    TUNE_BENCH_ARGS="$cfg" python tools/tune.py "$k" 2>&1 | sed "s/^/$cfg  /"
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
