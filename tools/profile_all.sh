set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r03_final synth256_f64_column_filter > /dev/null 2>&1
bash tools/profile_bench.sh r03_nofilter synth256_f64_column --no-filter > /dev/null 2>&1
PROF_F32=1 bash tools/profile_bench.sh r03_synth512f32 synth512f32_f32_column_filter --config synth512f32 > /dev/null 2>&1
bash tools/profile_bench.sh r03_searev searev_f64_column_filter --config searev > /dev/null 2>&1
bash tools/profile_bench.sh r03_ar1 ar1_f64_column_filter --config ar1 > /dev/null 2>&1
PROF_STEPS=4 bash tools/profile_bench.sh r03_coupled coupled256_f64_column --config coupled256 > /dev/null 2>&1
bash tools/profile_bench.sh r03_noisy noisy256_f64_column_filter --config noisy256 > /dev/null 2>&1
bash tools/profile_bench.sh r03_reservoirs reservoirs_f64_lead_filter --config reservoirs > /dev/null 2>&1
for t in r03_final r03_nofilter r03_synth512f32 r03_searev r03_ar1 r03_coupled r03_noisy r03_reservoirs; do echo "== $t"; grep -A3 "per-dispatch durations" gpurun_out/prof_$t/summary.txt | head -4; grep "SQ_INSTS_VALU \|FETCH_SIZE is\|TCC_HIT" gpurun_out/prof_$t/summary.txt; done
python tools/clock_probe.py gpurun_out/r03_clock.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('gpurun_out/r03_clock.json')); print('clock', d['f64']['ghz_median'], d['f32']['ghz_median'], d['f64']['stamped_kernel_ms'])"
python tools/filter_radius_probe.py > gpurun_out/r03_filter_radius_probe.txt 2>&1; cat gpurun_out/r03_filter_radius_probe.txt
python tools/filter_probe.py > gpurun_out/r03_filter_probe.txt 2>&1; tail -5 gpurun_out/r03_filter_probe.txt
SDP_STOCK_NOISE=0.07 python tools/filter_probe.py 256 3 > gpurun_out/r03_filter_probe_noisy.txt 2>&1; tail -3 gpurun_out/r03_filter_probe_noisy.txt
SDP_STOCK_NOISE=0.07 python tools/filter_ab.py 256 > gpurun_out/r03_filter_ab_noisy.txt 2>&1; tail -3 gpurun_out/r03_filter_ab_noisy.txt
python tools/lead_ab.py > gpurun_out/r03_lead_ab.txt 2>&1; cat gpurun_out/r03_lead_ab.txt
