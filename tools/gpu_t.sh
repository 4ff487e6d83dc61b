#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/t; mkdir -p $O
for b in 2 4 2 4; do echo "== SDP_COL_BATCH=$b"; SDP_COL_BATCH=$b timeout 600 python tools/config_times.py 2>&1 | grep -v "^$" | head -4; done > $O/cfg.txt 2>&1; cat $O/cfg.txt
