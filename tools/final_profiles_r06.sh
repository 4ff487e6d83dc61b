#!/bin/bash
# Round 6: the end-of-round evidence of tools/final_profiles.sh plus the profiles the verdict of round 5 asked for --
# the shifted lattice under its current key, the direct kernel on a fine 1-D grid and the line kernel that replaces it.
#     bash tools/final_profiles_r06.sh            (through gpurun, from the repo root)
set -u
bash tools/final_profiles.sh r06
OUT=$PWD/gpurun_out/final_r06
PROF_STEPS=6 bash tools/profile_bench.sh r06_noisy noisy256_f64_column_filter --config noisy256 > "$OUT/profile_noisy.log" 2>&1
PROF_STEPS=6 bash tools/profile_bench.sh r06_line1d inventory1d_fine_f64_line_filter --config inventory1d_fine > "$OUT/profile_line1d.log" 2>&1
PROF_STEPS=4 bash tools/profile_bench.sh r06_direct1d inventory1d_fine_f64_generic --config inventory1d_fine --kernel generic > "$OUT/profile_direct1d.log" 2>&1
# .. and the configurations whose counters dated from rounds 3-5: every pmc_<key>.json bench.py reads is this round's code
PROF_STEPS=20 bash tools/profile_bench.sh r06_ar1 ar1_f64_column_filter --config ar1 > "$OUT/profile_ar1.log" 2>&1
PROF_STEPS=10 bash tools/profile_bench.sh r06_searev searev_f64_column_filter --config searev > "$OUT/profile_searev.log" 2>&1
PROF_STEPS=6 bash tools/profile_bench.sh r06_reservoirs reservoirs_f64_lead_filter --config reservoirs > "$OUT/profile_reservoirs.log" 2>&1
PROF_STEPS=3 bash tools/profile_bench.sh r06_coupled256 coupled256_f64_column --config coupled256 > "$OUT/profile_coupled256.log" 2>&1
for t in r06_noisy r06_line1d r06_direct1d r06_ar1 r06_searev r06_reservoirs r06_coupled256; do
  cp gpurun_out/prof_$t/summary.txt "$OUT/${t}_summary.txt"
  cp gpurun_out/prof_$t/kernel_stats.csv "$OUT/${t}_kernel_stats.csv"
  cp gpurun_out/prof_$t/pmc_*.json "$OUT/" 2>/dev/null
done
ls -la "$OUT"
