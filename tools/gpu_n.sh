#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/n; mkdir -p $O
for cu in 4 3 6 8; do
  SDP_STG_CU=$cu timeout 600 python bench.py --config coupled256 --kernel staged --steps 3 --warmup 1 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('staged cu=$cu', d['ms_per_step'])"; tail -1 $O/tmp.err
done
