#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_percontrol.py tests/test_gpu_staged.py -x -q > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_sweep.py -q -k "random_models or fp32_512" > $O/pytest2.log 2>&1; tail -4 $O/pytest2.log
for k in auto staged; do
  timeout 600 python bench.py --config coupled256 --kernel $k --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_coupled_$k.json 2> $O/bench_coupled_$k.err; python -c "
import json; d=json.load(open('$O/bench_coupled_$k.json')); print('$k', d['config']['kernel_family'], d['ms_per_step'], d['roofline']['kernel_ms'])"; tail -2 $O/bench_coupled_$k.err
done
for knobs in "SDP_COL_A_ORDER=1" "SDP_COL_A_GROUP=8" "SDP_COL_A_ORDER=1 SDP_COL_A_GROUP=8" "SDP_COL_MIN_WAVES=2" ; do
  env $knobs timeout 600 python bench.py --config coupled256 --steps 3 --warmup 1 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('$knobs', d['config']['kernel_family'], d['ms_per_step'])"
done
