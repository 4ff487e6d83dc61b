#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_percontrol.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for knobs in "SDP_COLU_BLOCK=16" "SDP_COLU_BLOCK=1" "SDP_COLU_BLOCK=8" "SDP_COLU_BLOCK=32" "SDP_COLU_BLOCK=16 SDP_COL_WCHUNK=32" "SDP_COLU_BLOCK=8 SDP_COL_WCHUNK=8"; do
  env $knobs timeout 600 python bench.py --config coupled256 --steps 3 --warmup 1 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('$knobs', d['config']['kernel_family'], d['ms_per_step'])"; tail -1 $O/tmp.err
done
