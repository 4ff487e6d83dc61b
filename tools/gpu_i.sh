#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/i; mkdir -p $O
for knobs in "SDP_COL_WCHUNK=8" "SDP_COL_WCHUNK=32" "SDP_COL_WCHUNK=4" "SDP_COL_WCHUNK=8 SDP_COL_MIN_WAVES=6"; do
  env $knobs timeout 600 python bench.py --config coupled256 --steps 3 --warmup 1 --no-cpu-baseline > $O/tmp.json 2> $O/tmp.err; python -c "
import json; d=json.load(open('$O/tmp.json')); print('$knobs', d['config']['kernel_family'], d['ms_per_step'])"; tail -1 $O/tmp.err
done
PROF_STEPS=3 timeout 900 bash tools/profile_bench.sh r02_coupled_percontrol coupled256_f64_column --config coupled256 > $O/prof.log 2>&1; tail -32 $O/prof.log
