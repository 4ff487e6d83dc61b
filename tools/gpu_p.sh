#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/p; mkdir -p $O
python tools/tune.py "SDP_COL_STAGGER=0" "SDP_COL_STAGGER=4" "SDP_COL_STAGGER=8" "SDP_COL_STAGGER=12" "SDP_COL_STAGGER=0" "SDP_COL_STAGGER=8" "SDP_COL_STAGGER=16" "SDP_COL_STAGGER=2" > $O/ab.txt 2>&1; cat $O/ab.txt
