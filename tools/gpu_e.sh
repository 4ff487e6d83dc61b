#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sweep.py -q -k "policy_iteration or searev_policy" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_simulate.py tests/test_gpu_integration_doc.py tests/test_gpu_window.py -q > $O/pytest2.log 2>&1; tail -4 $O/pytest2.log
# column kernel knobs A/B (default first)
timeout 1200 python tools/tune.py "SDP_COL_A_ORDER=0" "SDP_COL_A_ORDER=1" "SDP_COL_A_ORDER=1 SDP_COL_A_GROUP=8" "SDP_COL_A_ORDER=1 SDP_COL_A_GROUP=2" "SDP_COL_BATCH=4" "SDP_COL_UNROLL_U=4" "SDP_COL_UNROLL_U=4 SDP_COL_BATCH=1" "SDP_COL_UNROLL_U=3" "SDP_COL_MIN_WAVES=3" "SDP_COL_UNROLL_W=8" "SDP_COL_UNROLL_W=2" > $O/tune.txt 2>&1; cat $O/tune.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_of_measured_issue_rate'), d['cpu_baseline'])"
for cfg in ar1 searev; do PROF_STEPS=10 timeout 600 bash tools/profile_bench.sh r02_$cfg ${cfg}_f64_column --config $cfg > $O/prof_$cfg.log 2>&1; tail -6 $O/prof_$cfg.log; done
PROF_F32=1 PROF_STEPS=5 timeout 900 bash tools/profile_bench.sh r02_synth512f32 synth512f32_f32_column --config synth512f32 > $O/prof_f32.log 2>&1; tail -8 $O/prof_f32.log
