#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/z6; mkdir -p $O
TUNE_BENCH_ARGS="--config synth512f32" python tools/tune.py "SDP_COL_UNROLL_U=2" "X=0" "SDP_COL_UNROLL_U=2" "X=0" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_sweep.py -q -k "fp32 or float32 or f32" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
