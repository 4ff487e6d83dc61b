#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/v; mkdir -p $O
python tools/tune.py "X=0" "SDP_COL_A_ORDER=2" "X=0" "SDP_COL_A_ORDER=2" "SDP_COL_A_ORDER=2 SDP_COL_A_GROUP=8" "SDP_COL_A_ORDER=2 SDP_COL_A_GROUP=2" "SDP_COL_A_ORDER=2 SDP_COL_A_GROUP=16" > $O/ab.txt 2>&1; cat $O/ab.txt
SDP_COL_A_ORDER=2 timeout 600 python -m pytest tests/test_gpu_sweep.py -q -x -k "synthetic or nas or ar1_reference or searev or column or fused or fp32 or inventory" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
