#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/q; mkdir -p $O
timeout 600 python tools/phase_probe.py > $O/phase.txt 2>&1; cat $O/phase.txt
