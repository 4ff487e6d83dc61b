#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/m; mkdir -p $O
python tools/tune.py "SDP_NO_POW2=1" "SDP_NO_POW2=0" "SDP_NO_POW2=1" "SDP_NO_POW2=0" "SDP_NO_POW2=1" "SDP_NO_POW2=0" > $O/ab.txt 2>&1; cat $O/ab.txt
