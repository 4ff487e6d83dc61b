#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
python tools/tune.py "SDP_COL_A_LW=64" "SDP_COL_A_LW=16" "SDP_COL_A_LW=32" "SDP_COL_A_LW=64" "SDP_COL_A_LW=16" "SDP_COL_A_LW=32" "SDP_COL_A_ORDER=0" > $O/ab.txt 2>&1; cat $O/ab.txt
