#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_percontrol.py -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_sweep.py -q -k "random_models" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
for knobs in "X=0" "SDP_COL_A_GROUP=8" "SDP_COL_A_GROUP=2" "SDP_COL_MIN_WAVES=3" "SDP_COL_MIN_WAVES=5"; do
  env $knobs timeout 600 python bench.py --config coupled256 --steps 3 --warmup 1 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('$knobs', d['config']['kernel_family'], d['ms_per_step'])"; tail -1 $O/tmp.err
done
