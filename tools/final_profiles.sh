#!/bin/bash
# End-of-round evidence in one gpurun call: the rocprofv3 trace + PMC passes of the default bench line and of config 5,
# the class-priced issue table from the fresh PMC totals, and the bench line that then prices its roofline with it.
# Everything lands in gpurun_out/final_<tag>/ (copy into profiles/ to commit).
#     bash tools/final_profiles.sh <tag>            (through gpurun, from the repo root)
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/final_$TAG
mkdir -p "$OUT"
# (the driver's own command line: --steps 20 --warmup 5)
PROF_STEPS=20 PROF_WARMUP=5 bash tools/profile_bench.sh ${TAG}_final synth256_f64_column_filter > "$OUT/profile_default.log" 2>&1
cp gpurun_out/prof_${TAG}_final/summary.txt "$OUT/${TAG}_final_summary.txt"
cp gpurun_out/prof_${TAG}_final/kernel_stats.csv "$OUT/${TAG}_final_kernel_stats.csv"
cp gpurun_out/prof_${TAG}_final/pmc_synth256_f64_column_filter.json "$OUT/"
cp gpurun_out/prof_${TAG}_final/pmc_synth256_f64_column_filter.json profiles/
python3 tools/issue_model.py > "$OUT/issue_model.log" 2>&1
cp profiles/issue_classes_synth256_f64_column_filter.json "$OUT/"
PROF_F32=1 bash tools/profile_bench.sh ${TAG}_synth512f32 synth512f32_f32_column_filter --config synth512f32 > "$OUT/profile_synth512f32.log" 2>&1
cp gpurun_out/prof_${TAG}_synth512f32/summary.txt "$OUT/${TAG}_synth512f32_summary.txt"
cp gpurun_out/prof_${TAG}_synth512f32/kernel_stats.csv "$OUT/${TAG}_synth512f32_kernel_stats.csv"
cp gpurun_out/prof_${TAG}_synth512f32/pmc_synth512f32_f32_column_filter.json "$OUT/"
cp gpurun_out/prof_${TAG}_synth512f32/pmc_synth512f32_f32_column_filter.json profiles/
python3 bench.py > "$OUT/${TAG}_bench_final.json" 2> "$OUT/bench_final.err"
python3 bench.py --config synth512f32 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench_synth512f32.json" 2> "$OUT/bench_synth512f32.err"
tail -c 400 "$OUT/${TAG}_bench_final.json"
ls -la "$OUT"
